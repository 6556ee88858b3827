"""Diagnostic: edadm_qconv3_f16x3_direct (LDS-resident patch) against the implicit-GEMM form of the same three-product convolution
and against an fp64 convolution; timing of both at calibration-graph and decoder shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import torch.nn.functional as F
from edadm import ops

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


torch.manual_seed(0)
for (B, H, W, C, N, ups, res) in [(2, 64, 64, 128, 192, False, True), (4, 32, 32, 192, 384, False, False), (2, 64, 64, 64, 128, True, True),
                                  (8, 16, 16, 576, 576, False, True), (16, 8, 8, 960, 960, False, False), (32, 64, 64, 192, 192, False, False),
                                  (32, 64, 64, 384, 192, False, False), (16, 64, 64, 512, 512, False, True),
                                  (1, 128, 128, 64, 192, False, True), (2, 256, 256, 32, 128, False, True), (1, 128, 128, 64, 128, True, False),
                                  (1, 512, 128, 32, 128, False, False), (16, 128, 128, 256, 256, False, True), (16, 128, 128, 512, 512, True, False),
                                  (16, 256, 256, 128, 128, False, True), (16, 256, 256, 256, 256, True, False)]:
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = torch.randn(B, Hs, Ws, C, device=dev)
    w = torch.randn(N, 3, 3, C, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    r = torch.randn(B, H, W, N, device=dev) if res else None
    outs = {}
    for flag in (True, False):
        ops.F16X3_DIRECT = flag
        outs[flag] = ops.conv2d_f16x3_nhwc(x, w, bias, residual=r, stride=1, pad=1, ups=ups)
        t = timeit(lambda: ops.conv2d_f16x3_nhwc(x, w, bias, residual=r, stride=1, pad=1, ups=ups))
        outs[(flag, "t")] = t
    ops.F16X3_DIRECT = True
    if B * H * W * C <= 2 ** 24:
        xd = x.double().permute(0, 3, 1, 2)
        if ups:
            xd = F.interpolate(xd, scale_factor=2, mode="nearest")
        ref = F.conv2d(xd, w.double().permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1)
        if res:
            ref = ref + r.double()
        sc = ref.abs().max().item()
        e = [(outs[f].double() - ref).abs().max().item() / sc for f in (True, False)]
    else:
        e = [float("nan")] * 2
    d = (outs[True] - outs[False]).abs().max().item() / outs[False].abs().max().item()
    fl = 2.0 * B * H * W * N * 9 * C * 3
    print("B %2d %3dx%-3d C %4d N %4d ups %d res %d | direct %7.1f us (%6.1f TF f16) implicit %7.1f us (%6.1f) | err vs fp64 %.1e / %.1e  direct-implicit %.1e"
          % (B, H, W, C, N, ups, res, outs[(True, "t")], fl / outs[(True, "t")] / 1e6, outs[(False, "t")], fl / outs[(False, "t")] / 1e6, e[0], e[1], d))
