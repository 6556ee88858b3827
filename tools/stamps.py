"""Diagnostic: in-kernel cycle stamps of k_gemm_nt8 (libedadm_stamps.so, `make -C eda-dm_amd/csrc stamps`)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EDADM_LIB_PATH"] = os.path.join(ROOT, "eda-dm_amd", "csrc", "libedadm_stamps.so")
os.environ["EDADM_GEMM_FORCE"] = os.environ.get("FORCE", "3")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops, lib
dev = torch.device("cuda", 0)
L = lib.load()

def case(M, N, K, mode, residual):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if residual else None
    oqp = torch.tensor([0.05, 128.0, 8.0, 0.0], device=dev)
    out = torch.empty(M, N, device=dev)
    if mode == 0:
        f = lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, out, residual=res)
    else:
        f = lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=res)
    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(3):
        f(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    if os.environ["EDADM_GEMM_FORCE"] in ("5", "6"):
        n = buf[3]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
        us = e0.elapsed_time(e1) * 1e3
        print("M=%d N=%d K=%d mode=%d res=%d | %.0f us | MFMA wave0: barrier=%.0f compute=%.0f epilogue=%.0f | loader w0: write(+wait loads)=%.0f barrier=%.0f load-issue=%.0f total=%.0f ticks" %
              (M, N, K, mode, residual, us, buf[0] / n, buf[1] / n, buf[2] / n, buf[4] / n, buf[5] / n, buf[6] / n, buf[7] / n))
        return
    n = buf[4]
    print("M=%d N=%d K=%d mode=%d res=%d | per-wave cycles: consts=%.0f first_tile=%.0f main(all)=%.0f epilogue=%.0f total=%.0f" %
          (M, N, K, mode, residual, buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[5] / n))

for M, N, K in ((102400, 384, 384), (102400, 3072, 384), (409600, 192, 1728)):
    for mode, residual in ((0, 0), (0, 1), (2, 0), (3, 0)):
        case(M, N, K, mode, residual)
