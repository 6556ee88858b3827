"""Diagnostic: achieved HBM bandwidth of the reconstruction loop's element-wise kernels at the 32 x 32 transformer block's sizes
(32 rows x 1024 tokens): fake-quant forward / backward with the 0.5 mask, Lp loss, f16 expansions, |x| maxima."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops

dev = torch.device("cuda", 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


d, z = torch.tensor([0.05], device=dev), torch.tensor([128.0], device=dev)
for (R, C) in [(32768, 384), (32768, 1536), (32768, 3072)]:
    x = torch.randn(R, C, device=dev)
    g = torch.randn(R, C, device=dev)
    nb = x.numel()
    y = torch.empty_like(x)
    t_c = timeit(lambda: y.copy_(x))
    t_f = timeit(lambda: ops.fake_quant_fwd(x, d, z, 255.0, prob=0.5, seed=7))
    t_f1 = timeit(lambda: ops.fake_quant_fwd(x, d, z, 255.0, prob=1.0, seed=7))
    t_b = timeit(lambda: ops.fake_quant_bwd(g, x, d, z, 255.0, prob=0.5, seed=7))
    t_b1 = timeit(lambda: ops.fake_quant_bwd(g, x, d, z, 255.0, prob=1.0, seed=7))
    t_m = timeit(lambda: ops.absmax_parts(x))
    am = ops.absmax_parts(x)
    w = torch.randn(384, C, device=dev)
    wb, inv_b, _ = ops.split_f16(w, 384, 1, C, 2, True)
    t_s = timeit(lambda: ops.split_f16(x, R, 1, C, 2, False, other=inv_b, N=384, amax=am))
    t_t = timeit(lambda: ops.transpose_split_f16(x, 2048, 2, amax=am))
    t_a = timeit(lambda: torch.add(x, g))
    print("R %6d C %5d | copy %5.2f TB/s | fq fwd p=.5 %6.1f us %5.2f TB/s  p=1 %6.1f us %5.2f | fq bwd p=.5 %6.1f us %5.2f TB/s  p=1 %6.1f us %5.2f | "
          "absmax %5.1f us %5.2f | split_f16 %6.1f us %5.2f | transpose_split %6.1f us %5.2f | torch add %6.1f us %5.2f"
          % (R, C, nb * 8 / t_c / 1e6, t_f, nb * 8 / t_f / 1e6, t_f1, nb * 8 / t_f1 / 1e6, t_b, nb * 12 / t_b / 1e6, t_b1, nb * 12 / t_b1 / 1e6,
             t_m, nb * 4 / t_m / 1e6, t_s, nb * 8 / t_s / 1e6, t_t, nb * 8 / t_t / 1e6, t_a, nb * 12 / t_a / 1e6))
