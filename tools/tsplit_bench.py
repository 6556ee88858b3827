"""Diagnostic: edadm_transpose_split_f16 (weight-gradient operands) bandwidth."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for (B, H, C, O) in ((32, 64, 192, 192), (32, 32, 384, 384), (32, 16, 576, 576)):
    x = torch.randn(B, H, H, C, device=dev)
    gy = torch.randn(B * H * H, O, device=dev)
    M = B * H * H
    S = 32
    Ms = M // S
    px, pg = ops.absmax_parts(x), ops.absmax_parts(gy)
    t_b = timeit(lambda: ops.transpose_split_f16(x, Ms, 2, amax=px, conv=(3, 3, 1, 1, H, H)))
    t_a = timeit(lambda: ops.transpose_split_f16(gy, Ms, 2, amax=pg))
    wb, wa = 9 * C * 2 * M * 2 / 1e9, O * 2 * M * 2 / 1e9
    print("C%d@%d: cols^T expansion %.3f ms (%.0f GB/s written) | gy^T expansion %.3f ms (%.0f GB/s written + %.0f read)" % (
        C, H, t_b, wb / t_b * 1e3, t_a, wa / t_a * 1e3, M * O * 4 / 1e9 / t_a * 1e3))

# plain matrices (the linear layers of a transformer block at batch 32): [tokens][features] -> [features][tokens] expansion
def ev(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (R, C) in ((32768, 384), (32768, 1536), (32768, 3072), (8192, 576), (2048, 960)):
    g = torch.randn(R, C, device=dev)
    S = 1
    while ((C + 127) // 128) * 3 * S < 256 and S < 64 and R % (S * 2 * 16) == 0 and R // (S * 2) >= 256:
        S *= 2
    pg = ops.absmax_parts(g)
    t = ev(lambda: ops.transpose_split_f16(g, R // S, 2, amax=pg))
    print("[%d][%d] -> transposed expansion, %d slabs: %.1f us = %.0f GB/s read + %.0f GB/s written" % (
        R, C, S, t * 1e3, R * C * 4 / t / 1e6, R * C * 4 / t / 1e6))
