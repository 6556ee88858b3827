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
                                                   (3, 64, 8, 32, 1, 1, 0), (2, 3, 16, 32, 3, 1, 1), (32, 192, 16, 192, 3, 1, 1)])
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
    for shape, O in (((100, 768), 192), ((4, 256, 96), 288), ((6, 10), 7)):
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
