"""-m gpu: the H1 training-graph kernels (csrc/train_ops.hip through edadm/train_ops.py) against the plain PyTorch fp32
ops they replace in the calibration graph -- forward values and input gradients.  Tolerances: 2e-5 of the tensor's range
forward, 5e-5 backward (fp32 reductions in another order; the GroupNorm statistics here are accumulated in fp64)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(name, got, ref, tol):
    rng = float(ref.abs().max())
    err = float((got - ref).abs().max())
    print("%-34s max err %.2e of range" % (name, err / max(rng, 1e-30)))
    assert err <= tol * max(rng, 1e-30), (name, err, rng)


@pytest.mark.parametrize("shape,G,silu", [((4, 64, 8, 8), 32, True), ((3, 192, 16, 16), 32, False), ((2, 96, 5, 7), 32, True),
                                           ((2, 64, 50), 32, False), ((2, 384, 32, 32), 32, True)])
def test_group_norm_silu_forward_backward(shape, G, silu):
    from edadm import train_ops as T
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 1.7 + 0.3).cuda().requires_grad_(True)
    norm = nn.GroupNorm(G, shape[1], eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(shape[1], generator=g))
        norm.bias.copy_(0.2 * torch.randn(shape[1], generator=g))
    gy = torch.randn(shape, generator=g).cuda()
    y = T.group_norm(x, norm, silu=silu)
    (y * gy).sum().backward()
    assert norm.weight.grad is None and norm.bias.grad is None       # affine gradients: not produced, by design (never trained)
    gx, x.grad = x.grad.clone(), None
    ref = F.group_norm(x, G, norm.weight, norm.bias, 1e-6)
    if silu:
        ref = ref * torch.sigmoid(ref)
    (ref * gy).sum().backward()
    close("group_norm%s fwd %s" % ("+silu" if silu else "", shape), y.detach(), ref.detach(), 2e-5)
    close("group_norm%s bwd %s" % ("+silu" if silu else "", shape), gx, x.grad, 5e-5)


@pytest.mark.parametrize("shape,G,silu", [((4, 64, 8, 8), 32, True), ((3, 192, 16, 16), 32, False), ((2, 96, 5, 7), 32, True),
                                           ((2, 384, 32, 32), 32, True), ((64, 576, 16, 16), 32, True), ((2, 1024, 9, 3), 32, False)])
def test_group_norm_silu_channels_last(shape, G, silu):
    """The NHWC form (edadm_gn_fwd_nhwc / _bwd_nhwc: chunked per-channel partials) on channels_last tensors, as the calibration
    graph feeds it between convolutions: against torch, and against the NCHW kernel on the same values (same arithmetic per
    element; the group moments are summed in another order)."""
    from edadm import train_ops as T
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x0 = (torch.randn(shape, generator=g) * 1.7 + 0.3).cuda()
    norm = nn.GroupNorm(G, shape[1], eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(shape[1], generator=g))
        norm.bias.copy_(0.2 * torch.randn(shape[1], generator=g))
    gy = torch.randn(shape, generator=g).cuda()
    res = {}
    for name, fmt in (("nhwc", torch.channels_last), ("nchw", torch.contiguous_format)):
        x = x0.clone(memory_format=fmt).requires_grad_(True)
        y = T.group_norm(x, norm, silu=silu)
        if name == "nhwc":
            assert y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()     # stays in the layout
        (y * gy.contiguous(memory_format=fmt)).sum().backward()
        res[name] = (y.detach(), x.grad.clone())
        if name == "nhwc":
            assert x.grad.is_contiguous(memory_format=torch.channels_last)
    x = x0.clone().requires_grad_(True)
    ref = F.group_norm(x, G, norm.weight, norm.bias, 1e-6)
    if silu:
        ref = ref * torch.sigmoid(ref)
    (ref * gy).sum().backward()
    close("nhwc group_norm fwd %s" % (shape,), res["nhwc"][0], ref.detach(), 2e-5)
    close("nhwc group_norm bwd %s" % (shape,), res["nhwc"][1], x.grad, 5e-5)
    close("nhwc vs nchw kernel fwd", res["nhwc"][0], res["nchw"][0], 2e-6)
    close("nhwc vs nchw kernel bwd", res["nhwc"][1], res["nchw"][1], 5e-6)


def test_convolutional_unit_runs_in_one_layout():
    """contract.CHANNELS_LAST: convolutions hand on channels_last tensors, element-wise kernels and GroupNorm follow the memory order:
    a ResBlock's forward + backward launches no NCHW <-> NHWC conversion between its operators, and gives the NCHW graph's values."""
    from torch.profiler import profile, ProfilerActivity
    from edadm import contract, train_ops as T
    torch.manual_seed(3)
    conv1, conv2 = nn.Conv2d(64, 128, 3, padding=1).cuda(), nn.Conv2d(128, 128, 3, padding=1).cuda()
    n1, n2 = nn.GroupNorm(32, 64).cuda(), nn.GroupNorm(32, 128).cuda()
    x0 = torch.randn(16, 64, 64, 64, device="cuda")          # large enough for the implicit input-gradient path (no weight transposes)

    def run(x):
        h = contract.conv2d(T.group_norm(x, n1, silu=True), conv1.weight, conv1.bias, 1, 1)
        h = contract.conv2d(T.group_norm(h, n2, silu=True), conv2.weight, conv2.bias, 1, 1)
        return h

    outs = {}
    for flag in (True, False):
        contract.CHANNELS_LAST = flag
        try:
            x = (x0.clone(memory_format=torch.channels_last) if flag else x0.clone()).requires_grad_(True)
            run(x).square().sum().backward()                   # warm-up
            x.grad = None
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                y = run(x)
                y.square().sum().backward()
                torch.cuda.synchronize()
            names = [e.key for e in prof.key_averages()]
            conv = sum(e.count for e in prof.key_averages() if "k_nchw_to_nhwc" in e.key)
            outs[flag] = (y.detach().contiguous(), x.grad.contiguous(), conv)
        finally:
            contract.CHANNELS_LAST = True
    assert outs[True][2] == 0, "layout conversions in the channels_last graph"
    assert outs[False][2] >= 6
    close("unit output, one layout vs NCHW", outs[True][0], outs[False][0], 2e-6)
    close("unit input gradient, one layout vs NCHW", outs[True][1], outs[False][1], 5e-6)


@pytest.mark.parametrize("rows,C", [(64, 32), (1000, 384), (96, 960), (10, 1280), (33, 2000)])
def test_layer_norm_forward_backward(rows, C):
    from edadm import train_ops as T
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(4, rows // 4 if rows % 4 == 0 else rows, C, generator=g) * 2 - 0.5).cuda().requires_grad_(True)
    ln = nn.LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(C, generator=g))
        ln.bias.copy_(0.2 * torch.randn(C, generator=g))
    gy = torch.randn(x.shape, generator=g).cuda()
    y = T.layer_norm(x, ln)
    (y * gy).sum().backward()
    gx, x.grad = x.grad.clone(), None
    ref = F.layer_norm(x, (C,), ln.weight, ln.bias, ln.eps)
    (ref * gy).sum().backward()
    close("layer_norm fwd C=%d" % C, y.detach(), ref.detach(), 2e-5)
    close("layer_norm bwd C=%d" % C, gx, x.grad, 5e-5)


def test_geglu_silu_softmax_forward_backward():
    from edadm import train_ops as T
    g = torch.Generator().manual_seed(5)
    h = (torch.randn(3, 40, 2 * 96, generator=g) * 2).cuda().requires_grad_(True)
    gy = torch.randn(3, 40, 96, generator=g).cuda()
    y = T.geglu(h)
    (y * gy).sum().backward()
    gh, h.grad = h.grad.clone(), None
    a, gate = h.chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    (ref * gy).sum().backward()
    close("geglu fwd", y.detach(), ref.detach(), 2e-5)
    close("geglu bwd", gh, h.grad, 5e-5)

    x = (torch.randn(7, 130, generator=g) * 3).cuda().requires_grad_(True)
    gy = torch.randn(7, 130, generator=g).cuda()
    y = T.silu(x)
    (y * gy).sum().backward()
    gx, x.grad = x.grad.clone(), None
    ref = x * torch.sigmoid(x)
    (ref * gy).sum().backward()
    close("silu fwd", y.detach(), ref.detach(), 2e-6)
    close("silu bwd", gx, x.grad, 2e-5)

    for cols in (16, 77, 256, 1024, 4100):
        s = (torch.randn(6, 9, cols, generator=g) * 4).cuda().requires_grad_(True)
        gy = torch.randn(6, 9, cols, generator=g).cuda()
        p = T.softmax(s)
        (p * gy).sum().backward()
        gs, s.grad = s.grad.clone(), None
        ref = s.softmax(dim=-1)
        (ref * gy).sum().backward()
        close("softmax fwd cols=%d" % cols, p.detach(), ref.detach(), 2e-6)
        close("softmax bwd cols=%d" % cols, gs, s.grad, 2e-5)


@pytest.mark.parametrize("Z,M,N,K", [(8, 64, 64, 32), (16, 256, 77, 40), (3, 100, 36, 24), (64, 1024, 1024, 8), (2, 16, 16, 192)])
def test_attention_products_forward_backward(Z, M, N, K):
    """bmm_nt (the attention products of quant_block.py:204-235,427-446 on the exact-fp32 MFMA) and the batched transpose,
    with both operand gradients, against einsum."""
    from edadm import train_ops as T
    g = torch.Generator().manual_seed(Z + M + N + K)
    a = torch.randn(Z, M, K, generator=g).cuda().requires_grad_(True)
    b = torch.randn(Z, N, K, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(Z, M, N, generator=g).cuda()
    y = T.bmm_nt(a, b, 0.37)
    (y * gy).sum().backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    ref = torch.einsum("zmk,znk->zmn", a.double(), b.double()) * 0.37
    (ref * gy.double()).sum().backward()
    close("bmm_nt fwd", y.detach().double(), ref.detach(), 1e-5)
    close("bmm_nt dA", ga.double(), a.grad.double(), 1e-5)
    close("bmm_nt dB", gb.double(), b.grad.double(), 1e-5)
    v = torch.randn(Z, N, 24, generator=g).cuda().requires_grad_(True)
    t = T.transpose12(v)
    assert t.shape == (Z, 24, N) and torch.equal(t, v.detach().transpose(1, 2))
    (t * t.detach()).sum().backward()
    assert torch.equal(v.grad, v.detach())


def test_training_graph_has_no_stock_norm_or_blas_kernels(golden):
    """One reconstruction iteration of a transformer block and a ResBlock under the torch profiler: no GroupNorm / LayerNorm /
    SiLU / GELU / softmax kernel of ATen and no rocBLAS / hipBLASLt GEMM runs -- they all go through libedadm.so."""
    from torch.profiler import profile, ProfilerActivity
    from helpers import build_ldm, quantize_like_reference
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, ctx), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    qnn.set_quant_state(True, True)
    x = x.clone().requires_grad_(True)
    qnn.model(x, t, ctx).sum().backward()            # warm-up (code objects, allocator)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        qnn.model(x, t, ctx).sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if getattr(e, "device_time_total", 0) > 0 or getattr(e, "cuda_time_total", 0) > 0]
    kernels = [n for n in names if not n.startswith("aten::") and not n.startswith("hip") and not n.startswith("Memcpy")]
    stock = ("GroupNorm", "group_norm", "layer_norm", "LayerNorm", "RowwiseMoments", "silu", "Gelu", "gelu", "softmax", "Softmax")
    bad = [n for n in kernels if "Cijk_" in n or "rocblas" in n or "hipblas" in n
           or (("at::native" in n or "at_cuda" in n) and any(k in n for k in stock))]
    ours = [n for n in kernels if n.startswith(("k_gn_", "k_ln_", "k_geglu_", "k_silu", "k_softmax", "k_transpose_batched"))]
    assert len(ours) >= 6, ours
    print("device kernels of one forward+backward:", len(kernels))
    assert not bad, bad
