"""Diagnostic: edadm_gemm_f32_nt throughput on calibration-graph shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, K in ((131072, 192, 1728), (32768, 384, 3456), (4096, 4096, 4096), (8192, 8192, 2048), (192, 1728, 131072)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    ms = timeit(lambda: ops.gemm_f32_nt(a, b, M, N, K))
    ms_t = timeit(lambda: torch.matmul(a, b.t()))
    print("M=%d N=%d K=%d: own %.3f ms %.1f TF/s | rocBLAS %.3f ms %.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, ms_t, 2.0 * M * N * K / ms_t / 1e9))
