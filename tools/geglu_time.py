"""Diagnostic: the GEGLU feed-forward projection (int8 GEMM, out_mode 3: a * gelu(gate) -> int8 operand in the epilogue) against the
same contraction with a plain int8 (out_mode 2) and an fp32 output: what the epilogue's arithmetic costs at the headline shapes."""
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


oqp = ops.qp_tensor([(0.02, 128.0, 255.0)], dev)
for (M, N, K) in [(102400, 3072, 384), (25600, 4608, 576), (6400, 7680, 960)]:
    A = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    W = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    sc = torch.rand(N, device=dev) * 1e-3
    bi = torch.randn(N, device=dev) * 0.1
    out = torch.empty(M, N, device=dev)
    t0 = timeit(lambda: ops.qgemm_i8(A, W, M, N, K, sc, bi, out))
    t2 = timeit(lambda: ops.qgemm_i8_q(A, W, M, N, K, sc, bi, 2, oqp))
    t3 = timeit(lambda: ops.qgemm_i8_q(A, W, M, N, K, sc, bi, 3, oqp))
    fl = 2.0 * M * N * K
    print("M %6d N %5d K %4d   fp32 out %7.1f us   int8 out %7.1f us (%6.0f TOP/s)   GEGLU int8 out %7.1f us (%6.0f TOP/s)"
          % (M, N, K, t0, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6))
