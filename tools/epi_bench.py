"""Diagnostic: what the epilogue costs on short-K layers (out modes, residual on/off) + pure store rate."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    return e0.elapsed_time(e1) / n

def case(M, N, K, mode, residual):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if residual else None
    oqp = torch.tensor([0.05, 128.0, 8.0, 0.0], device=dev)
    if mode == 0:
        out = torch.empty(M, N, device=dev)
        f = lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, out, residual=res)
        ob = M * N * 4
    else:
        f = lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=res)
        ob = M * N * (2 if mode == 1 else 1) // (2 if mode == 3 else 1)
    ms = timeit(f)
    by = M * K + ob + (M * N * 4 if residual else 0)
    print("M=%d N=%d K=%d mode=%d res=%d : %.3f ms  %.0f TF/s  %.2f TB/s (alg bytes %.0f MB)" %
          (M, N, K, mode, residual, ms, 2.0 * M * N * K / ms / 1e9, by / ms / 1e9, by / 1e6))

for M, N, K in ((102400, 384, 384), (409600, 192, 1728), (102400, 3072, 384)):
    for mode in (0, 1, 2, 3):
        for residual in (0, 1):
            if mode == 3 and residual: continue
            case(M, N, K, mode, residual)
x = torch.empty(409600 * 192, device=dev)
print("fill 315MB: %.3f ms" % timeit(lambda: x.fill_(1.0)))
y = torch.empty_like(x)
print("copy 315MB: %.3f ms" % timeit(lambda: y.copy_(x)))
print("add  315MB x3: %.3f ms" % timeit(lambda: torch.add(x, y, out=y)))
