"""Diagnostic: the weight-gradient GEMM (edadm_gemm_f16x3_nt over S split-K slabs + the ordered slab sum) against the slab count S,
at the (rows, outputs, reduction) shapes of the LDM-4 reconstruction units (batch 32).  contract._split picks S."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops, contract

dev = torch.device("cuda", 0)


def timeit(fn, n=8):
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


shapes = [(131072, 192, 9 * 192), (131072, 192, 9 * 384), (131072, 384, 9 * 384), (131072, 192, 384), (32768, 384, 9 * 384), (32768, 384, 9 * 768),
          (32768, 384, 384), (32768, 3072, 384), (32768, 384, 1536), (65536, 384, 384), (8192, 576, 9 * 576), (8192, 576, 9 * 1152), (8192, 576, 576),
          (8192, 4608, 576), (8192, 576, 2304), (2048, 960, 9 * 960), (2048, 960, 960), (2048, 7680, 960)]
for (M, O, K) in shapes:
    gy = torch.randn(M, O, device=dev)
    a = torch.randn(M, K, device=dev) if M * K * 4 < (3 << 30) else None
    if a is None:
        continue
    cur = contract._split(M, O, K)
    res = []
    for S in (1, 2, 4, 8, 16, 32, 64, 128):
        if M % (S * 16) or M // S < 128:
            continue
        Ms = M // S
        gt, _ = ops.transpose_split_f16(gy, Ms, 2)
        at, _ = ops.transpose_split_f16(a, Ms, 2)
        def run():
            slabs = ops.gemm_f16x3_nt(gt, gt.shape[1], 2 * Ms, at, at.shape[1], 2 * Ms, S, O, K, 2 * Ms)
            return ops.sum_slabs(slabs) if S > 1 else slabs[0]
        res.append((S, timeit(run)))
        del gt, at
    best = min(res, key=lambda r: r[1])
    tcur = dict(res).get(cur)
    print("M %6d O %5d K %5d  tiles %4d  now S=%2d %7.1f us | best S=%2d %7.1f us | %s" % (
        M, O, K, ((O + 127) // 128) * ((K + 127) // 128), cur, tcur if tcur else -1, best[0], best[1],
        " ".join("%d:%.0f" % r for r in res)))
