"""Diagnostic: achieved HBM bandwidth of the sampling path's normalisation passes at the headline shapes (100 rows per UNet call):
GroupNorm apply -> int8 operand (k_gn_apply16), its partial sums (k_gn_partial), LayerNorm -> three int8 operands (k_ln_quant_v4)."""
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


qp = ops.qp_tensor([(0.05, 128.0, 255.0)] * 3, dev)
print("GroupNorm (B=100)")
for (hw, C) in [(4096, 192), (4096, 384), (4096, 576), (1024, 384), (1024, 768), (1024, 960), (256, 576), (256, 1152), (256, 1536), (64, 960), (64, 1920)]:
    x = torch.randn(100, hw, 1, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    st = ops.groupnorm_stats(x, 32, 1e-5)
    t_s = timeit(lambda: ops.groupnorm_stats(x, 32, 1e-5))
    t_a = timeit(lambda: ops.groupnorm_apply(x, st, g, b, 32, True, qp=qp, nq=1))
    t_n = timeit(lambda: ops.groupnorm_apply(x, st, g, b, 32, False, qp=qp, nq=1))
    y = torch.empty_like(x)
    t_c = timeit(lambda: y.copy_(x))
    nb = x.numel()
    print("  HW %5d C %5d  stats %7.1f us %5.2f TB/s   apply->i8 %7.1f us %5.2f TB/s   without swish %7.1f us %5.2f TB/s   torch copy %5.2f TB/s"
          % (hw, C, t_s, nb * 4 / t_s / 1e6, t_a, nb * 5 / t_a / 1e6, t_n, nb * 5 / t_n / 1e6, nb * 8 / t_c / 1e6))
print("LayerNorm -> 3 int8 operands")
for (rows, C) in [(102400, 384), (25600, 576), (6400, 960)]:
    x = torch.randn(rows, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    for nq in (1, 3):
        t = timeit(lambda: ops.layernorm_quant(x, g, b, 1e-5, qp=qp, nq=nq))
        print("  rows %6d C %4d nq %d  %7.1f us %5.2f TB/s" % (rows, C, nq, t, x.numel() * (4 + nq) / t / 1e6))
