"""Diagnostic: in-kernel cycle stamps of the persistent GEGLU launch (k_gemm_p, FORCE=5) on a stamps build given by EDADM_LIB_PATH,
inputs scaled like a real layer (pre-activations of unit size, a step size that leaves the exact path rare)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EDADM_GEMM_FORCE", "5")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops, lib
dev = torch.device("cuda", 0)
L = lib.load()
torch.manual_seed(0)
for (M, N, K, mode) in ((102400, 3072, 384, 3), (102400, 3072, 384, 2), (102400, 3072, 384, 0)):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    sc = torch.full((N,), 1.5e-4, device=dev) * (0.5 + torch.rand(N, device=dev))
    bs = torch.randn(N, device=dev) * 0.1
    oqp = torch.tensor([0.02, 128.0, 255.0, 0.0], device=dev)
    out = torch.empty(M, N, device=dev)
    f = (lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, out)) if mode == 0 else (lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp))
    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(3):
        f(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    us = e0.elapsed_time(e1) * 1e3
    n = max(buf[3], 1)
    print("%s M=%d N=%d K=%d mode=%d | %.0f us | MFMA wave0 per launch: wait=%.0f compute=%.0f epilogue=%.0f | loader w0: write(+wait loads)=%.0f wait-free=%.0f load-issue=%.0f total=%.0f ticks (100 MHz)"
          % (os.path.basename(os.environ.get("EDADM_LIB_PATH", "product")), M, N, K, mode, us, buf[0] / n, buf[1] / n, buf[2] / n, buf[4] / n, buf[5] / n, buf[6] / n, buf[7] / n))
