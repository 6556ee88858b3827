"""Diagnostic: three-product f16 expansion vs the exact-fp32 MFMA path -- accuracy against fp64 and time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timeit(fn, n=12):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.time()
        fn()
        torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for (B, H, C, N) in ((32, 64, 192, 192), (32, 32, 384, 384), (32, 16, 576, 576), (32, 8, 960, 960), (50, 256, 128, 128)):
    x = (torch.randn(B, H, H, C, generator=g) * torch.where(torch.rand(B, H, H, C, generator=g) < 1e-3, 50.0, 1.0)).to(dev)
    w = (torch.randn(N, 3, 3, C, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    if B * H * H * C * 6 >= 1 << 31:
        x = x[:8].contiguous(); B = 8
    y32 = ops.conv2d_f32_nhwc(x, w, b)
    y16 = ops.conv2d_f16x3_nhwc(x, w, b)
    nb = min(B, 2)
    ref = torch.nn.functional.conv2d(x[:nb].double().permute(0, 3, 1, 2).cpu(), w.double().permute(0, 3, 1, 2).cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1)
    rms = ref.pow(2).mean().sqrt()
    e32 = (y32[:nb].double().cpu() - ref).abs(); e16 = (y16[:nb].double().cpu() - ref).abs()
    t32 = timeit(lambda: ops.conv2d_f32_nhwc(x, w, b)); t16 = timeit(lambda: ops.conv2d_f16x3_nhwc(x, w, b))
    fl = 2.0 * B * H * H * 9 * C * N / 1e9
    print("conv B%d %dx%d C%d->%d: fp32 %.3f ms (%.0f TF/s) err mean %.2e max %.2e | f16x3 %.3f ms (%.0f TF/s equiv) err mean %.2e max %.2e" % (
        B, H, H, C, N, t32, fl / t32, e32.mean() / rms, e32.max() / rms, t16, fl / t16, e16.mean() / rms, e16.max() / rms))
for (M, K, N) in ((32768, 384, 3072), (32768, 1536, 384), (131072, 192, 192), (2048, 960, 960), (32768, 384, 384), (32768, 384, 768), (8192, 576, 576), (131072, 192, 384), (131072, 384, 192)):
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(dev); b = torch.randn(N, generator=g).to(dev)
    y32 = ops.gemm_f32_nt(a, w, M, N, K, bias=b); y16 = ops.matmul_f16x3_nt(a, w, b)
    ref = a[:256].double().cpu() @ w.double().cpu().T + b.double().cpu()
    rms = ref.pow(2).mean().sqrt()
    e32 = (y32[:256].double().cpu() - ref).abs(); e16 = (y16[:256].double().cpu() - ref).abs()
    t32 = timeit(lambda: ops.gemm_f32_nt(a, w, M, N, K, bias=b)); t16 = timeit(lambda: ops.matmul_f16x3_nt(a, w, b))
    fl = 2.0 * M * K * N / 1e9
    print("linear %dx%dx%d: fp32 %.3f ms (%.0f TF/s) err mean %.2e max %.2e | f16x3 %.3f ms (%.0f TF/s equiv) err mean %.2e max %.2e" % (
        M, K, N, t32, fl / t32, e32.mean() / rms, e32.max() / rms, t16, fl / t16, e16.mean() / rms, e16.max() / rms))
