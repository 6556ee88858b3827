"""Diagnostic: quantised-output dense layers of LDM-4 on the persistent 4-wave kernel (EDADM_GEMM_NTQ=1) against k_gemm_nt (=0), diagnostic build."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for rep in range(2):
        for f in os.environ.get("NTQ_SET", "1,0").split(","):
            env = dict(os.environ, EDADM_GEMM_NTQ=f, EDADM_GEMM_P=os.environ.get("EDADM_GEMM_P", "1"), EDADM_LIB_PATH=os.path.join(ROOT, "eda-dm_amd", "csrc", "libedadm_diag.so"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), f], env=env, capture_output=True, text=True)
            print(r.stdout, end="")
            if r.returncode:
                print("NTQ=%s failed: %s" % (f, r.stderr[-600:]))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


oqp = torch.tensor([0.05, 128.0, 255.0, 0.0], device=dev)
row = ["NTQ=%s" % sys.argv[1]]
ref = {}
for M, N, K, mode, res in ((102400, 3072, 384, 3, 0), (102400, 384, 384, 1, 0), (102400, 384, 384, 2, 0), (25600, 576, 576, 1, 0), (6400, 960, 960, 1, 0), (102400, 384, 1536, 2, 1),
                           (25600, 576, 2304, 2, 1), (6400, 960, 3840, 2, 1), (25600, 4608, 576, 3, 0), (6400, 7680, 960, 3, 0)):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    us = timeit(lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=r))
    o = ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=r)
    row.append("%dx%dx%d m%d r%d: %6.1f us (sum %d)" % (M, N, K, mode, res, us, int(o.view(torch.int8 if mode != 1 else torch.float16).float().sum().item())))
print(" | ".join(row))
