"""Diagnostic: the quantised-output GEMM shapes of LDM-4 under each kernel structure (EDADM_GEMM_FORCE: 0 heuristic,
2 four-wave tile, 3 eight-wave tile, 5 persistent wave-specialised) on the DIAGNOSTIC build (`make -C eda-dm_amd/csrc diag`: the
product library has these heuristics compiled in).  python tools/force_sweep.py  (re-runs itself per value)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for f in ("0", "2", "3", "5"):
        env = dict(os.environ, EDADM_GEMM_FORCE=f, EDADM_LIB_PATH=os.path.join(ROOT, "eda-dm_amd", "csrc", "libedadm_diag.so"))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), f], env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode:
            print("FORCE=%s failed: %s" % (f, r.stderr[-400:]))
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
row = ["FORCE=%s" % sys.argv[1]]
for M, N, K, mode, res in ((102400, 3072, 384, 3, 0), (25600, 4608, 576, 3, 0), (6400, 7680, 960, 3, 0), (102400, 384, 384, 1, 0),
                           (25600, 576, 576, 1, 0), (102400, 384, 1536, 2, 1), (25600, 576, 2304, 2, 1), (102400, 384, 384, 0, 1),
                           (409600, 192, 384, 0, 0)):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    try:
        if mode == 0:
            out = torch.empty(M, N, device=dev)
            us = timeit(lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, out, residual=r))
        else:
            us = timeit(lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=r))
        row.append("%dx%dx%d m%d r%d: %6.1f us" % (M, N, K, mode, res, us))
    except Exception as e:
        row.append("%dx%dx%d m%d: n/a" % (M, N, K, mode))
print(" | ".join(row))
