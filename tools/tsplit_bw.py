"""Diagnostic: edadm_transpose_split_f16 (the weight gradient's operands: transpose + slab cut + two-term f16 expansion in one pass)
at the sizes of a 64x64 ResBlock iteration (batch 32): bytes moved / time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops, contract

dev = torch.device("cuda", 0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


B, H, W = 32, 64, 64
M = B * H * W
for (O, C) in [(192, 384), (192, 192), (384, 384)]:
    gy = torch.randn(M, O, device=dev)
    x = torch.randn(B, H, W, C, device=dev)
    for S in (contract._split(M, O, 9 * C), 2 * contract._split(M, O, 9 * C), 4 * contract._split(M, O, 9 * C)):
        Ms = M // S
        pg, px = ops.absmax_parts(gy), ops.absmax_parts(x)
        t_g = timeit(lambda: ops.transpose_split_f16(gy, Ms, 2, amax=pg))
        t_a = timeit(lambda: ops.transpose_split_f16(x, Ms, 2, amax=px, conv=(3, 3, 1, 1, H, W)))
        t_1 = timeit(lambda: ops.transpose_split_f16(x.reshape(M, C), Ms, 2, amax=px))
        bg = M * O * (4 + 4)
        ba = M * C * 4 + M * 9 * C * 4
        b1 = M * C * 8
        gt, _ = ops.transpose_split_f16(gy, Ms, 2, amax=pg)
        at, _ = ops.transpose_split_f16(x, Ms, 2, amax=px, conv=(3, 3, 1, 1, H, W))
        t_w = timeit(lambda: ops.gemm_f16x3_nt(gt, gt.shape[1], 2 * Ms, at, at.shape[1], 2 * Ms, S, O, 9 * C, 2 * Ms))
        fl = 2.0 * M * O * 9 * C * 3
        print("O %4d C %4d S %2d | dY^T %7.1f us %5.2f TB/s | im2col(x)^T %7.1f us %5.2f TB/s (write %.2f GB) | plain x^T %7.1f us %5.2f TB/s | "
              "wgrad GEMM %7.1f us %6.1f TF f16" % (O, C, S, t_g, bg / t_g / 1e6, t_a, ba / t_a / 1e6, M * 9 * C * 4 / 1e9, t_1, b1 / t_1 / 1e6,
                                                    t_w, fl / t_w / 1e6))
