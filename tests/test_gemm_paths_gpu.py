"""-m gpu: the three K4 kernel structures of edadm_qgemm_i8 / _q (4-wave tile, 8-wave tile, persistent
wave-specialised kernel) and their epilogues (LDS-staged, register-direct fp32, transposed-accumulator quantised)
on shapes big enough to reach each of them.  Integer accumulation bit-exact (scale 1, bias 0); the scaled
epilogue against fp64 at 1e-5; quantised outputs against the stand-alone quantiser kernels on the fp32 output
of the same GEMM (codes may differ by one where the fp32 value sits within rounding of a code boundary:
<= 1e-4 of the elements, never by more than one code)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a device"
    from edadm import ops as _ops
    return _ops


def _mk(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
    W = torch.randint(-8, 9, (N, K), generator=g, dtype=torch.int8)
    scale = torch.rand(N, generator=g) * 1e-3 + 1e-4
    bias = torch.randn(N, generator=g)
    return A, W, scale, bias, g


# M multiple of 256 with >= 224 tiles -> persistent kernel; 57344 + 100 -> 8-wave kernel with an edge tile;
# 3000 -> 4-wave kernel
@pytest.mark.parametrize("M,N,K,rpb", [(57344, 192, 256, 64), (57344, 384, 448, 1024), (57444, 192, 320, 100),
                                         (3000, 192, 192, 48), (65536, 128, 256, 4096)])
def test_qgemm_paths_fp32(ops, M, N, K, rpb):
    A, W, scale, bias, g = _mk(M, N, K, M + N + K)
    nb = (M + rpb - 1) // rpb
    rowadd = torch.randn(nb, N, generator=g)
    res = torch.randn(M, N, generator=g)
    Ad, Wd = A.cuda(), W.cuda()
    out = torch.empty(M, N, device="cuda")
    ops.qgemm_i8(Ad, Wd, M, N, K, torch.ones(N).cuda(), torch.zeros(N).cuda(), out)
    acc = (Ad.float() @ Wd.float().t())                    # |acc| < 2^24: exact in fp32
    assert torch.equal(out, acc)
    ops.qgemm_i8(Ad, Wd, M, N, K, scale.cuda(), bias.cuda(), out, rowadd=rowadd.cuda(), rows_per_batch=rpb, residual=res.cuda())
    ref = acc.double() * scale.double().cuda() + bias.double().cuda() + rowadd.double().cuda()[torch.arange(M, device="cuda") // rpb] \
        + res.double().cuda()
    assert (out.double() - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))
    ops.qgemm_i8(Ad, Wd, M, N, K, scale.cuda(), bias.cuda(), out, residual=res.cuda())
    ref = acc.double() * scale.double().cuda() + bias.double().cuda() + res.double().cuda()
    assert (out.double() - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,H,Cin,Cout,stride", [(56, 32, 192, 192, 1), (64, 32, 64, 384, 1), (224, 32, 128, 192, 2)])
def test_qgemm_conv_persistent(ops, B, H, Cin, Cout, stride):
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randint(-128, 128, (B, H, H, Cin), generator=g, dtype=torch.int8)
    w = torch.randint(-8, 9, (Cout, 3, 3, Cin), generator=g, dtype=torch.int8)
    padval = 5
    Ho = H // stride
    M, K = B * Ho * Ho, 9 * Cin
    geom = ops.make_geom(B, H, H, Cin, Ho, Ho, 3, 3, stride, 1, False, padval)
    out = torch.empty(M, Cout, device="cuda")
    ops.qgemm_i8(x.cuda(), w.reshape(Cout, K).cuda(), M, Cout, K, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda(), out, geom=geom)
    xp = F.pad(x.cuda().permute(0, 3, 1, 2).float(), (1, 1, 1, 1), value=padval)
    ref = torch.zeros(B, Cout, Ho, Ho, device="cuda")
    for c0 in range(0, Cin, 64):                            # fp32 conv in channel slices keeps every partial sum exact
        ref += F.conv2d(xp[:, c0:c0 + 64].double(), w.cuda().permute(0, 3, 1, 2)[:, c0:c0 + 64].double(), stride=stride).float()
    assert torch.equal(out, ref.permute(0, 2, 3, 1).reshape(M, Cout))


@pytest.mark.parametrize("M,N,K", [(57344, 384, 256), (57444, 192, 192), (3000, 384, 128)])
def test_qgemm_quantised_outputs(ops, M, N, K):
    A, W, scale, bias, g = _mk(M, N, K, 3 * M + N + K)
    scale = scale * 4
    Ad, Wd, sd, bd = A.cuda(), W.cuda(), scale.cuda(), bias.cuda()
    f32 = torch.empty(M, N, device="cuda")
    ops.qgemm_i8(Ad, Wd, M, N, K, sd, bd, f32)
    qp = ops.qp_tensor([(0.037, 119.0, 255.0)], "cuda")

    def codes_close(got, want, name):
        d = (got.int() - want.int()).abs()
        assert int(d.max()) <= 1, name
        assert float((d > 0).float().mean()) <= 1e-4, (name, float((d > 0).float().mean()))

    codes_close(ops.qgemm_i8_q(Ad, Wd, M, N, K, sd, bd, 2, qp), ops.quant_i8(f32, qp), "int8")
    h = ops.qgemm_i8_q(Ad, Wd, M, N, K, sd, bd, 1, qp)
    codes_close(h.float(), ops.quant_f16(f32, qp).float(), "f16")
    # the fused GEGLU reads interleaved (value, gate) columns; the stand-alone kernel takes [values | gates]
    split = torch.cat([f32[:, 0::2], f32[:, 1::2]], 1).contiguous()
    fused, alone = ops.qgemm_i8_q(Ad, Wd, M, N, K, sd, bd, 3, qp), ops.geglu_quant_i8(split, qp)
    codes_close(fused, alone, "geglu")
    # the epilogue's fast GELU (Abramowitz-Stegun erf + reciprocal multiply) is accepted only outside a guard band around
    # the rounding boundaries, inside it the exact form (erf to < 1 ulp, IEEE division) decides: the codes are those of
    # the stand-alone kernel, which only knows the exact form -- every one of them
    assert torch.equal(fused, alone), int((fused != alone).sum())


@pytest.mark.parametrize("B,Nk,N,K", [(56, 1024, 384, 384), (50, 64, 960, 960), (7, 256, 576, 192)])
def test_qgemm_transposed_f16_output(ops, B, Nk, N, K):
    """out_mode 4: the f16 operand of mode 1, stored transposed per image ([B][N][Nk]) -- bit-identical codes."""
    M = B * Nk
    A, W, scale, bias, g = _mk(M, N, K, M + 7 * N + K)
    qp = ops.qp_tensor([(0.041, 123.0, 255.0)], "cuda")
    args = (A.cuda(), W.cuda(), M, N, K, (scale * 4).cuda(), bias.cuda())
    assert ops.vt_mode_ok(M, N, Nk)
    plain = ops.qgemm_i8_q(*args, 1, qp)                                   # [M][N]
    tr = ops.qgemm_i8_q(*args, 4, qp, rows_per_batch=Nk)                   # [B][N][Nk]
    assert tr.shape == (B, N, Nk)
    assert torch.equal(tr, plain.reshape(B, Nk, N).transpose(1, 2))


# the small cases take 128-pixel tiles (fewer than 200 workgroups of 256 pixels); (100, 16, 16, 128, 960) is large enough for
# 256-pixel tiles at that level, (6, 8, 8, ...) fits 128-pixel tiles only (two images each)
@pytest.mark.parametrize("B,H,W,Cin,N", [(3, 64, 64, 192, 192), (5, 32, 32, 384, 384), (6, 16, 16, 576, 192), (8, 8, 8, 960, 384),
                                          (2, 64, 64, 64, 192), (4, 32, 32, 1152, 384), (2, 16, 16, 128, 576),
                                          (100, 16, 16, 128, 960), (6, 8, 8, 192, 192), (3, 16, 32, 64, 192),
                                          (4, 32, 32, 128, 128), (3, 16, 16, 256, 256), (2, 32, 32, 640, 640), (8, 8, 8, 1280, 1280),
                                          (2, 64, 64, 320, 640), (2, 64, 64, 320, 320), (1, 16, 16, 320, 320), (2, 32, 32, 640, 320),
                                          (4, 8, 8, 64, 192)])
def test_direct_conv3_equals_the_implicit_gemm_bit_for_bit(ops, B, H, W, Cin, N):
    """edadm_qconv3_i8_direct (input patch of a 256-pixel tile resident in LDS, weights streamed) against edadm_qgemm_i8's
    implicit-GEMM gather on the same operands: integer accumulation, same epilogue arithmetic -> identical fp32 bits, with
    and without the per-image row-add and the fp32 residual; and against an fp32 reference convolution."""
    assert ops.conv3_direct_ok(B, H, W, Cin, N)
    g = torch.Generator().manual_seed(B * H + Cin + N)
    x = torch.randint(-128, 128, (B, H, W, Cin), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, 3, 3, Cin), generator=g, dtype=torch.int8).cuda()
    scale = (torch.rand(N, generator=g) * 1e-3 + 1e-4).cuda()
    bias = torch.randn(N, generator=g).cuda()
    M, K = B * H * W, 9 * Cin
    padval = -17
    rowadd = torch.randn(B, N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    wdc = ops.conv3_pack_w(w.reshape(N, K), N, Cin)
    geom = ops.make_geom(B, H, W, Cin, H, W, 3, 3, 1, 1, False, padval)
    ones, zeros = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    # exact integer sums first (scale 1, bias 0): against an fp32 convolution in 64-channel slices (every partial sum exact)
    got = ops.qconv3_i8_direct(x, wdc, B, H, W, Cin, N, padval, ones, zeros, torch.empty(M, N, device="cuda"))
    xp = F.pad(x.permute(0, 3, 1, 2).float(), (1, 1, 1, 1), value=padval)
    ref = torch.zeros(B, N, H, W, device="cuda")
    for c0 in range(0, Cin, 64):
        ref += F.conv2d(xp[:, c0:c0 + 64].double(), w.permute(0, 3, 1, 2)[:, c0:c0 + 64].double()).float()
    assert torch.equal(got, ref.permute(0, 2, 3, 1).reshape(M, N))
    for ra, rs in ((None, None), (rowadd, None), (None, res), (rowadd, res)):
        if ra is not None and H * W < 64:
            continue
        a = ops.qgemm_i8(x, w.reshape(N, K), M, N, K, scale, bias, torch.empty(M, N, device="cuda"), geom=geom, rowadd=ra,
                         rows_per_batch=H * W, residual=rs)
        b = ops.qconv3_i8_direct(x, wdc, B, H, W, Cin, N, padval, scale, bias, torch.empty(M, N, device="cuda"), rowadd=ra,
                                 rows_per_batch=H * W, residual=rs)
        assert torch.equal(a, b), (ra is not None, rs is not None, float((a - b).abs().max()))


@pytest.mark.parametrize("B,Hin,Cin,N", [(3, 32, 384, 384), (5, 16, 576, 192), (8, 4, 960, 192), (2, 8, 128, 384), (4, 16, 1280, 1280),
                                         (2, 16, 128, 128)])
def test_direct_conv3_over_the_folded_upsample(ops, B, Hin, Cin, N):
    """The Upsample convolution (openaimodel.py:110-118): 3x3 over the nearest-2x upsampled tensor, which neither kernel
    writes -- the direct kernel's patch loader and the implicit GEMM's gather both read pixel (y / 2, x / 2); same bits."""
    H = 2 * Hin
    assert ops.conv3_direct_ok(B, H, H, Cin, N)
    g = torch.Generator().manual_seed(B + Hin + Cin)
    x = torch.randint(-128, 128, (B, Hin, Hin, Cin), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, 3, 3, Cin), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-3 + 1e-4).cuda(), torch.randn(N, generator=g).cuda()
    M, K, padval = B * H * H, 9 * Cin, 9
    geom = ops.make_geom(B, Hin, Hin, Cin, H, H, 3, 3, 1, 1, True, padval)
    a = ops.qgemm_i8(x, w.reshape(N, K), M, N, K, scale, bias, torch.empty(M, N, device="cuda"), geom=geom)
    b = ops.qconv3_i8_direct(x, ops.conv3_pack_w(w.reshape(N, K), N, Cin), B, H, H, Cin, N, padval, scale, bias,
                             torch.empty(M, N, device="cuda"), ups=True)
    assert torch.equal(a, b), float((a - b).abs().max())
    xu = x.permute(0, 3, 1, 2).float().repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = F.conv2d(F.pad(xu, (1, 1, 1, 1), value=padval).double(), w.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1).reshape(M, N)
    got = ops.qconv3_i8_direct(x, ops.conv3_pack_w(w.reshape(N, K), N, Cin), B, H, H, Cin, N, padval, torch.ones(N, device="cuda"),
                               torch.zeros(N, device="cuda"), torch.empty(M, N, device="cuda"), ups=True)
    assert torch.equal(got.double(), ref)


@pytest.mark.parametrize("B,H,Cin,N", [(24, 64, 192, 192), (6, 16, 576, 576), (8, 8, 960, 384), (4, 32, 128, 256), (8, 8, 640, 1280), (2, 32, 320, 320), (4, 16, 640, 320)])
def test_direct_conv3_groupnorm_partials(ops, B, H, Cin, N):
    """edadm_qconv3_i8_direct with gn_ws: per-channel (sum, sum of squares) of every 64-row slab of the output, summed in
    the epilogue's registers (128-pixel tiles -- the last two cases -- split a slab over two waves, which add up in the same
    order); reduced by edadm_groupnorm_final_cat[_rep2] they give the statistics of the two-pass kernels on
    the same output to 2e-5 (fp32 partial sums in another order, combined in fp64) -- alone, as the second half of a skip
    concatenation, and as the half-batch part of a guidance pair read periodically.  The output itself does not change."""
    g = torch.Generator().manual_seed(B + H + N)
    x = torch.randint(-128, 128, (B, H, H, Cin), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, 9 * Cin), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-2 + 1e-3).cuda(), torch.randn(N, generator=g).cuda()
    M, HW = B * H * H, H * H
    wdc = ops.conv3_pack_w(w, N, Cin)
    res = torch.randn(M, N, generator=g).cuda()
    out0 = ops.qconv3_i8_direct(x, wdc, B, H, H, Cin, N, 3, scale, bias, torch.empty(M, N, device="cuda"), residual=res)
    rows = 64
    assert ops.conv3_direct_tile(B, H, H, Cin, N) == (256 if H == 64 else 128)
    ws = torch.full((M // rows, N, 2), float("nan"), device="cuda")
    out = ops.qconv3_i8_direct(x, wdc, B, H, H, Cin, N, 3, scale, bias, torch.empty(M, N, device="cuda"), residual=res, gn_ws=ws)
    assert torch.equal(out, out0) and torch.isfinite(ws).all()
    G = 32
    xo = out.reshape(B, HW, N)
    tol = lambda got, want: ((got - want).abs() <= 2e-5 * want.abs().clamp_min(1.0)).all()
    assert tol(ops.groupnorm_final(ws, N, None, 0, B, HW, G, 1e-5, rows1=rows), ops.groupnorm_stats(xo, G, 1e-5))
    other = torch.randn(B, HW, 64, generator=g).cuda() * 3
    ws_o = torch.empty(M // 64, 64, 2, device="cuda")
    o3 = other.reshape(M // 64, 64, 64)
    ws_o[..., 0], ws_o[..., 1] = o3.sum(1), (o3 * o3).sum(1)
    assert tol(ops.groupnorm_final(ws_o, 64, ws, N, B, HW, G, 1e-5, rows2=rows), ops.groupnorm_stats(ops.Cat(other, xo), G, 1e-5))
    assert tol(ops.groupnorm_final(ws, N, ws_o, 64, B, HW, G, 1e-5, rows1=rows), ops.groupnorm_stats(ops.Cat(xo, other), G, 1e-5))
    # guidance pair: the second part holds half the images and is read periodically
    big = torch.cat([other, other * 0.5])
    ws_b = torch.cat([ws_o, ws_o * torch.tensor([0.5, 0.25], device="cuda")])
    got = ops.groupnorm_final(ws_b, 64, ws, N, 2 * B, HW, G, 1e-5, B2=B, rows2=rows)
    want = ops.groupnorm_stats(ops.Cat(big, xo), G, 1e-5)
    assert tol(got, want)


def test_direct_conv3_partials_do_not_depend_on_the_tile(ops):
    """The same images convolved inside a small batch (128-pixel tiles: two waves per 64-row slab) and inside a large one
    (256-pixel tiles: one wave per slab) give the same output AND the same GroupNorm partials, bit for bit: the sampling
    loop evaluates the prefix of a guidance pair at half the batch and must reproduce the doubled evaluation exactly."""
    g = torch.Generator().manual_seed(5)
    Cin = N = 192
    H = 64
    big, small = 16, 4
    assert ops.conv3_direct_tile(big, H, H, Cin, N) == 256 and ops.conv3_direct_tile(small, H, H, Cin, N) == 128
    x = torch.randint(-128, 128, (big, H, H, Cin), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, 9 * Cin), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-2 + 1e-3).cuda(), torch.randn(N, generator=g).cuda()
    rowadd = torch.randn(big, N, generator=g).cuda()
    wdc = ops.conv3_pack_w(w, N, Cin)

    def run(xs, ra):
        B = xs.shape[0]
        M = B * H * H
        ws = torch.full((M // 64, N, 2), float("nan"), device="cuda")
        out = ops.qconv3_i8_direct(xs, wdc, B, H, H, Cin, N, 0, scale, bias, torch.empty(M, N, device="cuda"), rowadd=ra,
                                   rows_per_batch=H * H, gn_ws=ws)
        return out, ws

    ob, wb = run(x, rowadd)
    os_, ws_ = run(x[:small].contiguous(), rowadd[:small].contiguous())
    n = small * H * H
    assert torch.equal(ob[:n], os_)
    assert torch.equal(wb[:n // 64], ws_)


@pytest.mark.parametrize("B,H,K,N,res", [(4, 32, 384, 384, True), (8, 16, 576, 576, True), (16, 8, 960, 960, True), (2, 64, 192, 192, False)])
def test_dense_layer_groupnorm_partials_from_the_epilogue(ops, B, H, K, N, res):
    """edadm_qgemm_i8_gn on a dense layer (a transformer's proj_out with its residual, attention.py:247-275): the output of the plain
    launch bit for bit, and per-channel partials of every 64-row slab that reduce to the two-pass statistics of that output (2e-5:
    fp32 sums in another order, combined in fp64) -- alone and as one half of a skip concatenation."""
    g = torch.Generator().manual_seed(B + H + N)
    HW = H * H
    M = B * HW
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, K), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-2 + 1e-3).cuda(), torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    assert ops.qgemm_i8_gn_ok(M, N, HW)
    out0 = ops.qgemm_i8(a, w, M, N, K, scale, bias, torch.empty(M, N, device="cuda"), residual=r)
    ws = torch.full((M // 64, N, 2), float("nan"), device="cuda")
    out = ops.qgemm_i8(a, w, M, N, K, scale, bias, torch.empty(M, N, device="cuda"), residual=r, gn_ws=ws, gn_hw=HW)
    assert torch.equal(out, out0) and torch.isfinite(ws).all()
    o3 = out.reshape(M // 64, 64, N).double()
    assert ((ws[..., 0].double() - o3.sum(1)).abs() <= 1e-5 * o3.abs().sum(1).clamp_min(1.0)).all()
    assert ((ws[..., 1].double() - (o3 * o3).sum(1)).abs() <= 1e-5 * (o3 * o3).sum(1).clamp_min(1.0)).all()
    G = 32
    xo = out.reshape(B, HW, N)
    tol = lambda got, want: ((got - want).abs() <= 2e-5 * want.abs().clamp_min(1.0)).all()
    assert tol(ops.groupnorm_final(ws, N, None, 0, B, HW, G, 1e-5, rows1=64), ops.groupnorm_stats(xo, G, 1e-5))
    other = torch.randn(B, HW, 64, generator=g).cuda() * 3
    ws_o = torch.empty(M // 64, 64, 2, device="cuda")
    o64 = other.reshape(M // 64, 64, 64)
    ws_o[..., 0], ws_o[..., 1] = o64.sum(1), (o64 * o64).sum(1)
    assert tol(ops.groupnorm_final(ws, N, ws_o, 64, B, HW, G, 1e-5, rows1=64), ops.groupnorm_stats(ops.Cat(xo, other), G, 1e-5))


@pytest.mark.parametrize("B,H,Cin,N,stride", [(4, 64, 192, 192, 2), (8, 32, 384, 384, 2), (16, 16, 576, 576, 2), (3, 32, 192, 384, 1)])
def test_implicit_gemm_groupnorm_partials_equal_the_direct_convolutions(ops, B, H, Cin, N, stride):
    """The implicit-GEMM kernel's partials (a Downsample convolution, openaimodel.py:143-170) -- and, for a stride-1 3x3 layer both
    kernels can take, THE SAME BITS as edadm_qconv3_i8_direct's: one order of sums per 64-row slab whichever kernel produced the
    tensor (a layer may change kernels with the batch size; the sampling loop's half-batch prefix must not change statistics)."""
    g = torch.Generator().manual_seed(B + H + N + stride)
    x = torch.randint(-128, 128, (B, H, H, Cin), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, 9 * Cin), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-2 + 1e-3).cuda(), torch.randn(N, generator=g).cuda()
    Ho = H // stride
    M, HW, K = B * Ho * Ho, Ho * Ho, 9 * Cin
    if not ops.qgemm_i8_gn_ok(M, N, HW):
        pytest.skip("shape outside edadm_qgemm_i8_gn_ok")
    geom = ops.make_geom(B, H, H, Cin, Ho, Ho, 3, 3, stride, 1, False, 5)
    rowadd = torch.randn(B, N, generator=g).cuda()
    out0 = ops.qgemm_i8(x, w, M, N, K, scale, bias, torch.empty(M, N, device="cuda"), geom=geom, rowadd=rowadd, rows_per_batch=HW)
    ws = torch.full((M // 64, N, 2), float("nan"), device="cuda")
    out = ops.qgemm_i8(x, w, M, N, K, scale, bias, torch.empty(M, N, device="cuda"), geom=geom, rowadd=rowadd, rows_per_batch=HW,
                       gn_ws=ws, gn_hw=HW)
    assert torch.equal(out, out0) and torch.isfinite(ws).all()
    tol = lambda got, want: ((got - want).abs() <= 2e-5 * want.abs().clamp_min(1.0)).all()
    assert tol(ops.groupnorm_final(ws, N, None, 0, B, HW, 32, 1e-5, rows1=64), ops.groupnorm_stats(out.reshape(B, HW, N), 32, 1e-5))
    if stride == 1 and ops.conv3_direct_ok(B, H, H, Cin, N):
        ws_d = torch.full((M // 64, N, 2), float("nan"), device="cuda")
        out_d = ops.qconv3_i8_direct(x, ops.conv3_pack_w(w, N, Cin), B, H, H, Cin, N, 5, scale, bias, torch.empty(M, N, device="cuda"),
                                     rowadd=rowadd, rows_per_batch=HW, gn_ws=ws_d)
        assert torch.equal(out_d, out) and torch.equal(ws_d, ws)


@pytest.mark.parametrize("B,H,Cin,N", [(50, 32, 384, 384), (100, 16, 576, 576), (25, 64, 192, 192)])
def test_direct_conv3_tail_retiling_same_bits_and_partials(ops, B, H, Cin, N):
    """launch_conv3_direct (csrc/gemm.hip): when the 256-pixel tiles leave a mostly empty last round of workgroups, the rows beyond the
    last full round go to a second launch of 128-pixel tiles.  Output and GroupNorm partials are those of the implicit GEMM on the same
    layer, bit for bit (integer sums; one order of partial sums per 64-row slab whichever tile owns it)."""
    g = torch.Generator().manual_seed(B + H + N)
    x = torch.randint(-128, 128, (B, H, H, Cin), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, 9 * Cin), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-2 + 1e-3).cuda(), torch.randn(N, generator=g).cuda()
    M, HW, K = B * H * H, H * H, 9 * Cin
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = (M // 256) * (N // 192)
    assert ops.conv3_direct_tile(B, H, H, Cin, N) == 256 and tiles > ncu and 0 < tiles % ncu <= 0.6 * ncu      # the split is taken
    rowadd = torch.randn(B, N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    geom = ops.make_geom(B, H, H, Cin, H, H, 3, 3, 1, 1, False, 7)
    ws_g = torch.full((M // 64, N, 2), float("nan"), device="cuda")
    ref = ops.qgemm_i8(x, w, M, N, K, scale, bias, torch.empty(M, N, device="cuda"), geom=geom, rowadd=rowadd, rows_per_batch=HW,
                       residual=res, gn_ws=ws_g, gn_hw=HW)
    ws_d = torch.full((M // 64, N, 2), float("nan"), device="cuda")
    got = ops.qconv3_i8_direct(x, ops.conv3_pack_w(w, N, Cin), B, H, H, Cin, N, 7, scale, bias, torch.empty(M, N, device="cuda"),
                               rowadd=rowadd, rows_per_batch=HW, residual=res, gn_ws=ws_d)
    assert torch.equal(got, ref)
    assert torch.equal(ws_d, ws_g)


@pytest.mark.parametrize("M,N,K,mode,res", [(65536, 384, 384, 1, False), (65536, 384, 384, 2, False), (32768, 384, 1536, 2, True),
                                            (16384, 4608, 576, 3, False), (65536, 384, 384, 4, False), (40960, 256, 320, 1, False)])
def test_persistent_quantised_output_kernel_bit_identical_to_per_tile_launches(ops, M, N, K, mode, res):
    """k_gemm_ntq (csrc/gemm.hip): launches of >= 512 full tiles with a quantised output walk their tiles persistently and request
    the next tile's operands in front of the epilogue; the same rows in slices of fewer than 512 tiles take k_gemm_nt (one tile per
    workgroup).  Integer accumulation and one epilogue: the same bytes, for f16 codes, int8 codes (+ residual), GEGLU and the
    transposed f16 form (quant_layer.py:406-437 + the consumer's quantiser, quant_block.py:128-162)."""
    g = torch.Generator().manual_seed(M // 128 + N + K + mode)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-8, 9, (N, K), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 2e-3 + 1e-4).cuda(), torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    oqp = ops.qp_tensor([(0.03, 117.0, 255.0)], torch.device("cuda"))
    tn = 192 if N % 192 == 0 else 128
    assert (M // 128) * (N // tn) >= 512
    rpb = 1024 if mode == 4 else 1
    big = ops.qgemm_i8_q(a, w, M, N, K, scale, bias, mode, oqp, residual=r, rows_per_batch=rpb)
    rows = 128 * max(1, 511 // (N // tn))                     # slices of < 512 tiles
    if mode == 4:
        rows = max(rpb, rows // rpb * rpb)
    parts = []
    for m0 in range(0, M, rows):
        m1 = min(M, m0 + rows)
        parts.append(ops.qgemm_i8_q(a[m0:m1], w, m1 - m0, N, K, scale, bias, mode, oqp, residual=None if r is None else r[m0:m1],
                                    rows_per_batch=rpb))
    small = torch.cat(parts)
    assert big.shape == small.shape and torch.equal(big, small)
    assert float(big.float().abs().sum()) > 0


@pytest.mark.parametrize("M,N,K1,K2", [(1024, 192, 192, 192), (2048, 384, 576, 384), (640, 960, 960, 960), (4096, 192, 384, 192), (256, 576, 1152, 384)])
def test_split_quantiser_layer_in_one_launch_bit_identical_to_two(M, N, K1, K2):
    """K4s (csrc/gemm.hip, edadm_qgemm_i8_split2): the 1x1 skip convolution over [h | skip] with two activation / weight quantisers
    (quant_layer.py:415-427) as ONE launch with two accumulator sets against the two-launch form whose second launch accumulates
    through the residual port: the same fp32 bits, with and without a bias."""
    from edadm import ops
    g = torch.Generator().manual_seed(M + N + K1 + K2)
    A = torch.randint(-128, 128, (M, K1 + K2), generator=g, dtype=torch.int8).cuda()
    W1 = torch.randint(-8, 8, (N, K1), generator=g, dtype=torch.int8).cuda()
    W2 = torch.randint(-8, 8, (N, K2), generator=g, dtype=torch.int8).cuda()
    s1, s2 = (torch.rand(N, generator=g) * 1e-3 + 1e-4).cuda(), (torch.rand(N, generator=g) * 2e-3 + 1e-4).cuda()
    bias = torch.randn(N, generator=g).cuda()
    assert ops.qgemm_i8_split2_ok(M, N, K1, K2)
    for b in (bias, None):
        two = torch.empty(M, N, device="cuda")
        ops.qgemm_i8(A[:, :K1], W1, M, N, K1, s1, b, two, lda=K1 + K2)
        ops.qgemm_i8(A[:, K1:], W2, M, N, K2, s2, None, two, lda=K1 + K2, residual=two)
        one = ops.qgemm_i8_split2(A, W1, W2, M, N, K1, K2, s1, s2, b, torch.empty(M, N, device="cuda"))
        assert torch.equal(one, two), float((one - two).abs().max())
    # exact integer reference of the first rows
    ref = (A[:64, :K1].double() @ W1.double().t()) * s1.double() + bias.double() + (A[:64, K1:].double() @ W2.double().t()) * s2.double()
    one = ops.qgemm_i8_split2(A, W1, W2, M, N, K1, K2, s1, s2, bias, torch.empty(M, N, device="cuda"))
    assert float((one[:64].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("M,K,specs", [
    (65536, 384, [(384, 2), (384, 2), (384, 1)]),            # q / k / v of the 32 x 32 self-attention: int8, int8, f16 codes
    (25600, 576, [(576, 1), (576, 1), (576, 4)]),            # the 16 x 16 level: f16 codes, v transposed per image
    (16384, 384, [(3072, 3)]),                               # GEGLU projection 384 -> 3072 (pairs), 256 CUs x uneven row shares
    (12800, 576, [(4608, 3)]),                               # GEGLU projection 576 -> 4608: 240 of the 256 workgroups
    (32768, 384, [(384, 1), (192, 2), (576, 4), (192, 3)]),  # four problems of different widths and all four output forms
    (102400, 384, [(3072, 3)]),                              # the production shape of the 32 x 32 GEGLU projection
])
def test_grouped_weight_resident_kernel_bit_identical_to_separate_launches(ops, M, K, specs):
    """k_gemm_br (csrc/gemm.hip, edadm_qgemm_i8_grouped_q): the problems of one launch -- each with its OWN int8 operand, weights,
    scales, output form and consuming quantiser (the q / k / v QuantModules of ldm/modules/attention.py:168-176 keep separate input
    quantisers, quant_layer.py:406-437) -- give the bytes of one edadm_qgemm_i8_q launch each (k_gemm_ntq / k_gemm_p / k_gemm_nt:
    integer accumulation, the same register-direct epilogue)."""
    from edadm import lib
    g = torch.Generator().manual_seed(M // 128 + K + 7 * len(specs))
    assert ops.qgemm_i8_grouped_q_ok(M, sum(n for n, _ in specs), K)
    probs, refs = [], []
    for i, (N, mode) in enumerate(specs):
        a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
        w = torch.randint(-8, 9, (N, K), generator=g, dtype=torch.int8).cuda()
        scale = (torch.rand(N, generator=g) * 2e-3 + 1e-4).cuda()
        bias = torch.randn(N, generator=g).cuda() if i != 1 else None
        oqp = ops.qp_tensor([(0.03 + 0.004 * i, 117.0 + i, 255.0)], torch.device("cuda"))
        rpb = 1024 if mode == 4 else 0
        probs.append(dict(A=a, W=w, N=N, scale=scale, bias=bias, out_mode=mode, oqp=oqp, rows_per_batch=rpb))
        refs.append(ops.qgemm_i8_q(a, w, M, N, K, scale, bias, mode, oqp, rows_per_batch=rpb or 1))
    import ctypes as _ct
    lib.load().edadm_diag_launch_kernels((_ct.c_int32 * 8)())       # forget the reference launches' tags
    lib.CALLS = {}
    try:
        outs = ops.qgemm_i8_grouped_q(probs, M, K)
        calls = dict(lib.CALLS)
    finally:
        lib.CALLS = None
    assert calls == {"edadm_qgemm_i8_grouped_q": 1}
    import ctypes
    buf = (ctypes.c_int32 * 8)()
    n = lib.load().edadm_diag_launch_kernels(buf)
    assert n >= 1 and buf[n - 1] == 7                       # launch tag 7 = k_gemm_br
    ops.device_status()                                     # a hand-off that never arrived would raise here
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and o.dtype == r.dtype and torch.equal(o, r)
        assert float(o.float().abs().sum()) > 0
    # the launch again (hand-off words start from zero every launch; nothing carried over)
    again = ops.qgemm_i8_grouped_q(probs, M, K)
    assert all(torch.equal(x, y) for x, y in zip(again, refs))


def test_grouped_kernel_rejects_what_it_cannot_take(ops):
    from edadm import lib
    a = torch.zeros(1024, 384, dtype=torch.int8, device="cuda")
    w = torch.zeros(192, 384, dtype=torch.int8, device="cuda")
    s = torch.ones(192, device="cuda")
    oqp = ops.qp_tensor([(0.03, 117.0, 255.0)], torch.device("cuda"))
    assert not ops.qgemm_i8_grouped_q_ok(1024, 192, 384)          # too few tiles to fill the chip
    assert not ops.qgemm_i8_grouped_q_ok(102400, 3072, 960)       # the weight block does not fit LDS
    assert not ops.qgemm_i8_grouped_q_ok(102400, 3000, 384)
    with pytest.raises(lib.EdadmError):
        ops.qgemm_i8_grouped_q([dict(A=a, W=w, N=192, scale=s, bias=None, out_mode=2, oqp=oqp)], 1000, 384)
    bad = dict(A=a, W=w, N=192, scale=s, bias=None, out_mode=2, oqp=oqp, lda=380)      # rows not 16-byte aligned
    with pytest.raises(lib.EdadmError):
        ops.qgemm_i8_grouped_q([bad], 1024, 384)
    with pytest.raises(lib.EdadmError):
        ops.qgemm_i8_grouped_q([dict(A=a, W=w, N=192, scale=s, bias=None, out_mode=2, oqp=oqp)], 1024, 448)
