"""-m gpu: the fp32 contraction of the calibration graph (conv2d / linear forward + autograd backward
on the fp32-MFMA GEMM, im2col / col2im, split-K weight gradient) against torch fp64 on the host.
Tolerance: fp32 FMA chains over K <= 5184 terms -> 2e-5 relative to the result's scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cmp(a, b, tol=3e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape
    assert (a - b).abs().max() <= tol * max(1.0, float(b.abs().max())), float((a - b).abs().max())


@pytest.mark.parametrize("B,C,H,O,k,stride,pad", [(4, 32, 8, 48, 3, 1, 1), (2, 64, 16, 64, 3, 2, 1), (2, 64, 9, 32, 3, 2, 0),
                                                   (3, 64, 8, 32, 1, 1, 0), (2, 3, 16, 32, 3, 1, 1), (32, 192, 16, 192, 3, 1, 1),
                                                   (32, 64, 64, 64, 3, 1, 1), (32, 256, 8, 256, 3, 1, 1)])
# the last three take the f16 three-product path: implicit forward; forward + dgrad + gathered wgrad; split-K slabs (few tiles)
def test_conv2d_fwd_bwd(B, C, H, O, k, stride, pad):
    from edadm import contract
    g = torch.Generator().manual_seed(B * C + O)
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(O, C, k, k, generator=g) * 0.1
    b = torch.randn(O, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv2d(xd, wd, bd, stride=stride, padding=pad)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    out = contract.conv2d(xg, wg, bg, stride, pad)
    out.backward(gy.cuda())
    _cmp(out, ref), _cmp(xg.grad, xd.grad), _cmp(wg.grad, wd.grad, 1e-4), _cmp(bg.grad, bd.grad, 1e-4)


def test_linear_and_conv1d_fwd_bwd():
    from edadm import contract
    g = torch.Generator().manual_seed(5)
    for shape, O in (((100, 768), 192), ((4, 256, 96), 288), ((6, 10), 7), ((4096, 512), 512)):   # last: f16 three-product path
        x = torch.randn(*shape, generator=g)
        w = torch.randn(O, shape[-1], generator=g) * 0.1
        b = torch.randn(O, generator=g)
        xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
        ref = F.linear(xd, wd, bd)
        gy = torch.randn(ref.shape, generator=g)
        ref.backward(gy.double())
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        out = contract.linear(xg, wg, bg)
        out.backward(gy.cuda())
        _cmp(out, ref), _cmp(xg.grad, xd.grad), _cmp(wg.grad, wd.grad, 1e-4), _cmp(bg.grad, bd.grad, 1e-4)
    x = torch.randn(3, 32, 50, generator=g)
    w = torch.randn(96, 32, 1, generator=g) * 0.2
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv1d(xd, wd)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy.double())
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = contract.conv1d_k1(xg, wg)
    out.backward(gy.cuda())
    _cmp(out, ref), _cmp(xg.grad, xd.grad), _cmp(wg.grad, wd.grad, 1e-4)


@pytest.mark.parametrize("kind", ["normal", "outliers", "heavy_tail_small", "zeros"])
def test_f16_three_product_is_fp32_grade(kind):
    """edadm_split_f16 + edadm_qgemm_f16 (fp32 operands as hi + lo f16 terms, three products, fp32 accumulation) against
    fp64: no worse than the exact-fp32 MFMA path on the same data, whatever the operand statistics (activation outliers,
    gradients spread over many decades, all-zero rows)."""
    from edadm import ops
    g = torch.Generator().manual_seed(3)
    M, K, N = 4096, 1152, 384
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    if kind == "outliers":
        a = a * torch.where(torch.rand(M, K, generator=g) < 1e-3, 1000.0, 1.0)
    elif kind == "heavy_tail_small":
        a = a * 1e-7 * torch.exp(3 * torch.randn(M, K, generator=g))
        w = w * torch.exp(2 * torch.randn(N, 1, generator=g))
    elif kind == "zeros":
        a[:, ::2] = 0
        w[5] = 0
    b = torch.randn(N, generator=g) * float(a.abs().mean())
    ref = a.double() @ w.double().T + b.double()
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    y32 = ops.gemm_f32_nt(ad, wd, M, N, K, bias=bd).cpu().double()
    y16 = ops.matmul_f16x3_nt(ad, wd, bd).cpu().double()
    rms = float(ref.pow(2).mean().sqrt())
    e32, e16 = (y32 - ref).abs(), (y16 - ref).abs()
    print(kind, "fp32 mean %.2e max %.2e | f16x3 mean %.2e max %.2e (of rms)" % (e32.mean() / rms, e32.max() / rms, e16.mean() / rms, e16.max() / rms))
    assert e16.mean() <= 1.5 * e32.mean() + 1e-9 * rms and e16.max() <= 2.0 * e32.max() + 1e-7 * rms
    # the expansion itself: hi + lo reproduces x * 2^e to 2^-22 relative (or 2^-25 absolute for the smallest terms)
    planes, inv, _ = ops.split_f16(ad, M, 1, K, 0, False)
    p = planes.reshape(M, 3, K).float()
    assert torch.equal(p[:, 0], p[:, 2])
    back = (p[:, 0].double() + p[:, 1].double()) * float(inv[0])
    err = (back.cpu() - a.double()).abs()
    assert float((err - a.double().abs() * 2.0 ** -21).max()) <= 2.0 ** -24 * float(inv[0])


@pytest.mark.parametrize("B,H,W,C,N,ups,res", [(2, 64, 64, 128, 192, False, True), (4, 32, 32, 192, 384, False, False),
                                                (2, 64, 64, 64, 128, True, True), (8, 16, 16, 576, 576, False, True),
                                                (16, 8, 8, 960, 960, False, False), (1, 128, 128, 64, 192, False, True),
                                                (2, 256, 256, 32, 128, False, True), (1, 128, 128, 64, 128, True, False),
                                                (1, 512, 128, 32, 128, False, False), (3, 64, 64, 64, 320, False, False)])
def test_direct_three_product_convolution(B, H, W, C, N, ups, res):
    """edadm_qconv3_f16x3_direct (input patch resident in LDS; images wider than 64 pixels in 64-column blocks; the nearest-2x
    upsample read in place) against an fp64 convolution and against the implicit-GEMM form of the same three-product contraction:
    fp32-grade, and the two forms differ by accumulation order only."""
    import torch.nn.functional as F
    from edadm import ops
    torch.manual_seed(B + H + C + N)
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = torch.randn(B, Hs, Ws, C, device="cuda")
    w = torch.randn(N, 3, 3, C, device="cuda") * 0.05
    bias = torch.randn(N, device="cuda")
    r = torch.randn(B, H, W, N, device="cuda") if res else None
    assert ops.conv3_f16x3_direct_ok(B, H, W, C, N)
    out = {}
    for flag in (True, False):
        ops.F16X3_DIRECT = flag
        try:
            out[flag] = ops.conv2d_f16x3_nhwc(x, w, bias, residual=r, stride=1, pad=1, ups=ups)
        finally:
            ops.F16X3_DIRECT = True
    xd = x.double().permute(0, 3, 1, 2)
    if ups:
        xd = F.interpolate(xd, scale_factor=2, mode="nearest")
    ref = F.conv2d(xd, w.double().permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double()
    sc = ref.abs().max().item()
    e_d, e_i = ((out[f].double() - ref).abs().max().item() / sc for f in (True, False))
    assert e_d <= 5e-6 and e_d <= 3 * e_i + 1e-7, (e_d, e_i)


@pytest.mark.parametrize("B,C,H,W,O", [(4, 64, 32, 32, 192), (2, 192, 16, 16, 384), (8, 96, 16, 16, 576)])
def test_weight_gradient_orientation(B, C, H, W, O):
    """contract._wgrad_product forms the weight gradient as [9 C][O] when that wastes fewer rows of the 256 x 192 tiles and turns the
    result back: the same gradient as the [O][9 C] orientation up to the order of the fp32 slab sums, and as torch's in fp64."""
    from edadm import contract
    torch.manual_seed(B + C + O)
    x = torch.randn(B, C, H, W, device="cuda")
    w = (torch.randn(O, C, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    gy = torch.randn(B, O, H, W, device="cuda")
    grads = {}
    for flag in (True, False):
        contract.WGRAD_SWAP = flag
        try:
            w.grad = None
            (contract.conv2d(x, w, None, 1, 1) * gy).sum().backward()
            grads[flag] = w.grad.clone()
        finally:
            contract.WGRAD_SWAP = True
    wd = w.detach().double().requires_grad_(True)
    (F.conv2d(x.double(), wd, None, 1, 1) * gy.double()).sum().backward()
    sc = wd.grad.abs().max().item()
    for flag in (True, False):
        assert (grads[flag].double() - wd.grad).abs().max().item() <= 3e-6 * sc, flag
