"""Diagnostic: time edadm_qgemm_i8 on the dominant LDM-4 shapes (not part of the product)."""
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

def conv_case(B, H, Cin, Cout, residual=False, rowadd=False):
    x = torch.randint(-128, 128, (B, H, H, Cin), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (Cout, 9 * Cin), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(Cout, device=dev) * 1e-3, torch.randn(Cout, device=dev)
    M = B * H * H
    out = torch.empty(M, Cout, device=dev)
    res = torch.randn(M, Cout, device=dev) if residual else None
    ra = torch.randn(B, Cout, device=dev) if rowadd else None
    geom = ops.make_geom(B, H, H, Cin, H, H, 3, 3, 1, 1, False, -1)
    ms = timeit(lambda: ops.qgemm_i8(x, w, M, Cout, 9 * Cin, sc, bs, out, geom=geom, residual=res, rowadd=ra, rows_per_batch=H * H))
    fl = 2.0 * M * Cout * 9 * Cin
    print("conv  B=%d H=%d Cin=%d Cout=%d res=%d ra=%d : %.3f ms  %.1f TF/s" % (B, H, Cin, Cout, residual, rowadd, ms, fl / ms / 1e9))

def dense_case(M, N, K, residual=False):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev) if residual else None
    ms = timeit(lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, out, residual=res))
    print("dense M=%d N=%d K=%d res=%d : %.3f ms  %.1f TF/s  out %.0f MB" % (M, N, K, residual, ms, 2.0 * M * N * K / ms / 1e9, M * N * 4 / 1e6))

conv_case(100, 64, 192, 192)
conv_case(100, 64, 192, 192, residual=True)
conv_case(100, 64, 192, 192, rowadd=True)
conv_case(100, 64, 384, 192)
conv_case(100, 32, 384, 384)
conv_case(100, 16, 576, 576)
conv_case(100, 8, 960, 960)
dense_case(409600, 192, 1728)
dense_case(409600, 192, 3456)
dense_case(409600, 192, 192)
dense_case(102400, 384, 384)
dense_case(102400, 384, 384, residual=True)
dense_case(102400, 3072, 384)
dense_case(8192, 8192, 8192)
dense_case(4096, 4096, 4096)

def dense_q(M, N, K, mode):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    oqp = torch.tensor([0.05, 128.0, 255.0, 0.0], device=dev)
    ms = timeit(lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp))
    print("dense_q mode%d M=%d N=%d K=%d : %.3f ms  %.1f TF/s" % (mode, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))

dense_q(102400, 384, 384, 1)
dense_q(102400, 384, 384, 2)
dense_q(102400, 3072, 384, 3)
dense_case(102400, 3072, 384)
dense_case(25600, 576, 576)
dense_case(25600, 576, 576, residual=True)
dense_case(6400, 960, 960, residual=True)
dense_case(100, 960, 768)
