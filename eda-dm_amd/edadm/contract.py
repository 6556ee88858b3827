"""fp32 contraction of the calibration graph (H1) on hand-written kernels: `conv2d` / `conv1d(k=1)` /
`linear` forward and autograd backward (quant_layer.py:434 and the `loss.backward()` of block_recon.py:197).

    forward   Y[m][o]  = X(m, k) . W[o][k]          implicit GEMM over NHWC x, k = (ky, kx, c)
    dX        stride-1 "same" convolutions: the same implicit GEMM of dY with the flipped, transposed filter;
              otherwise dcols = dY[m][o] . W^T[k][o] -> col2im
    dW        dW[o][k] = sum_m dY^T[o][m] . X^T[k][m]   split over m into S slabs, summed in order

Large products run on the f16 MFMA as three products of two-term f16 expansions with fp32 accumulation
(edadm_split_f16 / edadm_transpose_split_f16 + edadm_qgemm_f16x3 / edadm_gemm_f16x3_nt: fp32-grade results at about
twice the exact-fp32 MFMA's rate); small or narrow ones on the exact-fp32 MFMA
(edadm_gemm_f32_nt, edadm_conv2d_f32_nhwc, edadm_im2col_f32 / edadm_col2im_f32).  EDADM_F16X3=0: exact path only.
"""
import os

import torch

from . import ops

IMPLICIT_DGRAD = True      # input gradients of stride-1 convolutions as implicit GEMMs (False: im2col GEMM + col2im; tools only)
# large products run as ONE f16-MFMA GEMM over the two-term f16 expansion of both fp32 operands (three products,
# fp32 accumulation: fp32-grade result at 2x the exact-fp32 MFMA's rate, csrc/elem.hip); 0 = exact-fp32 MFMA only
F16X3 = True
# Executed fp32-equivalent flops (2 M N K per product: forward, input gradient, weight gradient), counted on the host as
# the products are issued -- bench.py reads the counter around one captured reconstruction iteration for H1's roofline
FLOPS = [0.0]
F16X3_MIN_NK = 256 * 256      # smallest weight matrix (N x K) whose linear product takes the three-product f16 path (tools/recon_time.py: the 384-wide
                              # layers of the 32x32 transformer blocks 17.5 -> 16.7 ms per iteration; smaller ones are neutral)
# Convolutions hand their result (and input gradient) on in the layout they compute in: logical NCHW tensors with NHWC strides
# (torch channels_last), and take such tensors without a conversion pass.  The element-wise operators of the graph run on memory
# order (ops.mem_view), GroupNorm has an NHWC form (edadm_gn_fwd_nhwc): a convolutional unit's iteration keeps one layout from
# its cached inputs to its loss (edadm/recon.py stores those caches in NHWC) instead of converting around every convolution --
# 29 conversion passes, 10 % of a 64x64 ResBlock iteration.  False: NCHW between operators (tools / A-B tests).
CHANNELS_LAST = True
# few-tile 3x3 convolutions (the 8x8 / 16x16 levels) on the direct three-product kernel too instead of im2col + split-K GEMM (+ col2im):
# measured neutral at 16x16 and slower at 8x8 (tools/recon_time.py DIRECT_SMALL=1: ResBlock 960 @ 8x8 2.72 -> 2.84 ms) -- off
DIRECT_SMALL = False


def _f16x3_linear(M, N, K):
    """worth the two conversion passes: enough output columns per converted operand byte"""
    return F16X3 and K % 16 == 0 and N >= 256 and N * K >= F16X3_MIN_NK and M >= 2048


def _split(M, O, K):
    """Number of slabs along the reduction (row) axis for the weight gradient.  Slabs of ~2048 rows: the [128 + 192][2 x 2048] f16
    operand slabs of the workgroups that run together then share one L2 (4 MiB per XCD) instead of streaming 4-8 MiB slabs past
    it -- measured best or within 5 % of best on every unit shape of the LDM-4 walk (tools/wgrad_split.py: 131072 x 192 x 3456
    1107 -> 774 us, 8192 x 576 x 10368 555 -> 409); more slabs while the launch has fewer workgroups than the 256 CUs hold."""
    tiles = ((O + 127) // 128) * ((K + 127) // 128)
    ok = lambda s: s <= 64 and M % (s * 16) == 0 and M // s >= 256
    s = 1
    while M // (s * 2) >= 2048 and ok(s * 2):
        s *= 2
    while tiles * s < 256 and ok(s * 2):
        s *= 2
    return s


def _amax(t):
    """partial maxima of an operand that may be expanded to f16 terms (shared by every expansion of the tensor)"""
    return ops.absmax_parts(t) if F16X3 and t.numel() >= (1 << 17) and t.numel() % 4 == 0 else None


def _matmul_nt(a2d, w2d, bias=None, amax=None):
    """[M][K] . [N][K]^T; few-tile shapes (the 8x8 / 16x16 levels) are split along K into slabs so that
    the launch still covers the 256 CUs, and the slabs are summed in a fixed order."""
    M, K = a2d.shape
    N = w2d.shape[0]
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    S = 1
    while tiles * S < 192 and K % (S * 2 * 16) == 0 and K // (S * 2) >= 512:
        S *= 2
    if S == 1:
        if _f16x3_linear(M, N, K):
            return ops.matmul_f16x3_nt(a2d, w2d, bias, amax=amax)
        return ops.gemm_f32_nt(a2d, w2d, M, N, K, bias=bias)
    Ks = K // S
    if F16X3 and Ks % 16 == 0 and M >= 1024 and N >= 256:
        # the same slabs on the f16 three-product path: the [hi x16 | lo x16] layout cuts along K at any multiple of 16
        wb, inv_b, _ = ops.split_f16(w2d, N, 1, K, 2, True)
        xa, _, comb = ops.split_f16(a2d, M, 1, K, 2, False, other=inv_b, N=N, amax=amax)
        slabs = ops.gemm_f16x3_nt(xa, 2 * K, 2 * Ks, wb, 2 * K, 2 * Ks, S, M, N, 2 * Ks)
        out = ops.sum_slabs(slabs).mul_(comb)
        return out if bias is None else out.add_(bias)
    slabs = ops.gemm_f32_nt(a2d, w2d, M, N, Ks, lda=K, ldb=K, batch=S, strideA=Ks, strideB=Ks)
    out = ops.sum_slabs(slabs)
    return out if bias is None else out.add_(bias)


WGRAD_SWAP = True
# The weight gradient of a layer depends on nothing the rest of backward produces: its kernels (operand transposition / expansion, the
# split-K product, the slab sum) go to a second HIP stream, forked after dY is known and joined at the end of the layer's backward, next
# to the input gradient on the current stream.  Same kernels on the same data -- same bits.  Inside a graph capture the fork / join
# become parallel branches of the captured iteration.  Measured (tools/recon_time.py, SIDE_WGRAD=1 / 0): ResBlocks 3-5 % faster at
# every level (192 @ 64x64 5.78 -> 5.53 ms, 960 @ 8x8 2.66 -> 2.54, the up block 384 -> 192 @ 64x64 14.28 -> 13.74).
SIDE_WGRAD = True
SIDE_LINEAR_MAX_ROWS = 4096
_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(dev.index)
    if s is None:
        s = _SIDE[dev.index] = torch.cuda.Stream(dev)
    return s


class _Fork:
    """with _Fork(t, on): ...  runs the body on the side stream of t's device, ordered after everything enqueued so far on the current
    stream; join() makes the current stream wait for it."""

    def __init__(self, t, on):
        self.on = bool(on) and t.is_cuda
        if self.on:
            self.cur = torch.cuda.current_stream(t.device)
            self.side = _side_stream(t.device)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.cur)
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.on:
            self.ctx.__exit__(*a)
        return False

    def join(self):
        if self.on:
            self.cur.wait_stream(self.side)


def _wgrad_product(gt, at, S, O, K, Ms):
    """sum over the S slabs of gt [O][..] . at [K][..]^T -> [O][K].  The 8-wave kernel's tiles are 256 rows x 192 columns: with the
    outputs as rows, O = 192 fills 75 % of a tile's rows (384: two tiles at 75 %, 576: three) -- when the other orientation wastes
    less, the product is formed transposed ([K][O]: 256-row tiles over 9 C, O as one or more full 192-column blocks) and the small
    result turned back."""
    def waste(rows, cols):
        return (((rows + 255) // 256) * 256 / rows) * (((cols + 191) // 192) * 192 / cols)
    if WGRAD_SWAP and waste(K, O) < 0.9 * waste(O, K):
        slabs = ops.gemm_f16x3_nt(at, at.shape[1], 2 * Ms, gt, gt.shape[1], 2 * Ms, S, K, O, 2 * Ms)
        return (ops.sum_slabs(slabs) if S > 1 else slabs[0]).t().contiguous()      # [K][O] -> [O][K]: a weight-sized copy
    slabs = ops.gemm_f16x3_nt(gt, gt.shape[1], 2 * Ms, at, at.shape[1], 2 * Ms, S, O, K, 2 * Ms)
    return ops.sum_slabs(slabs) if S > 1 else slabs[0]


def _wgrad(gy2d, a2d, amax_g=None, amax_a=None, conv=None):
    """dW[o][k] = gy2d^T . a2d  (gy2d [M][O], a2d [M][K]).  conv = (x_nhwc, (KH, KW, stride, pad, Ho, Wo)): a2d is the
    im2col of x_nhwc and is never materialised (f16 three-product path only)."""
    M, O = gy2d.shape
    if conv is not None:
        xh, (KH, KW, stride, pad, Ho, Wo) = conv
        K = KH * KW * xh.shape[-1]
        S = _split(M, O, K)
        if F16X3 and xh.shape[-1] % 64 == 0 and (M // S) % 16 == 0 and M % 4 == 0 and amax_a is not None:
            Ms = M // S
            gt, inv_g = ops.transpose_split_f16(gy2d, Ms, 2, amax=amax_g)
            at, inv_a = ops.transpose_split_f16(xh, Ms, 2, amax=amax_a, conv=(KH, KW, stride, pad, Ho, Wo))
            return _wgrad_product(gt, at, S, O, K, Ms).mul_(inv_g * inv_a)
        a2d = ops.im2col_f32(xh, KH, KW, stride, pad, Ho, Wo)
    K = a2d.shape[1]
    if M % 4:
        amax_g = amax_a = None                                   # the reduction length must be a multiple of 4 floats: zero rows add 0
        Mp = (M + 3) // 4 * 4
        gy2d = torch.cat([gy2d, gy2d.new_zeros(Mp - M, O)])
        a2d = torch.cat([a2d, a2d.new_zeros(Mp - M, K)])
        M = Mp
    S = _split(M, O, K)
    Ms = M // S
    if F16X3 and Ms % 16 == 0 and M >= 2048 and O * K >= 128 * 128:
        # transposition, slab cut and two-term f16 expansion of both operands in one pass each; S slabs of K = 3 Ms on
        # the f16 MFMA; the two power-of-two scales come off after the ordered slab sum
        gt, inv_g = ops.transpose_split_f16(gy2d, Ms, 2, amax=amax_g)    # [O][S][Ms / 16][2][16]
        at, inv_a = ops.transpose_split_f16(a2d, Ms, 2, amax=amax_a)     # [K][S][Ms / 16][2][16]
        return _wgrad_product(gt, at, S, O, K, Ms).mul_(inv_g * inv_a)
    gyT, aT = ops.transpose_f32(gy2d), ops.transpose_f32(a2d)            # [O][M], [K][M]
    if S == 1:
        return ops.gemm_f32_nt(gyT, aT, O, K, M)
    slabs = ops.gemm_f32_nt(gyT, aT, O, K, Ms, lda=M, ldb=M, batch=S, strideA=Ms, strideB=Ms)
    return ops.sum_slabs(slabs)


def _pad4(t2d):
    """K (row length) must be a multiple of 4 floats for the 16-byte staging: zero-pad the tail."""
    K = t2d.shape[1]
    if K % 4 == 0:
        return t2d, K
    Kp = (K + 3) // 4 * 4
    out = torch.zeros(t2d.shape[0], Kp, dtype=t2d.dtype, device=t2d.device)
    out[:, :K] = t2d
    return out, K


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        x2p, K = _pad4(x2)
        wp, _ = _pad4(weight.contiguous())
        ctx.px = _amax(x2p)
        FLOPS[0] += 2.0 * x2p.shape[0] * wp.shape[0] * K
        out = _matmul_nt(x2p, wp, bias, amax=ctx.px)
        ctx.save_for_backward(x2p, wp)
        ctx.meta = (x.shape, K, bias is not None)
        return out.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2p, wp = ctx.saved_tensors
        xshape, K, has_bias = ctx.meta
        gy2 = gy.reshape(-1, gy.shape[-1]).contiguous()
        gy2p, O = _pad4(gy2)
        pg = _amax(gy2p) if gy2p is gy2 else None
        gx = gw = gb = None
        FLOPS[0] += 2.0 * gy2.shape[0] * wp.shape[0] * K * (int(ctx.needs_input_grad[0]) + int(ctx.needs_input_grad[1]))
        # linear layers: only where the two halves leave the GPU under-filled (the 8x8 level: tools/recon_time.py, transformer block
        # 960 @ 8x8 5.08 -> 4.88 ms; at 16x16 / 32x32 the side stream costs 1 %)
        fork = _Fork(gy2, SIDE_WGRAD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and gy2.shape[0] <= SIDE_LINEAR_MAX_ROWS)
        if ctx.needs_input_grad[1]:
            with fork:
                gw = _wgrad(gy2, x2p, amax_g=pg, amax_a=ctx.px)[:, :K].contiguous()
        if ctx.needs_input_grad[0]:
            wT, _ = _pad4(ops.transpose_f32(wp))                                   # [Kp][O]
            gx = _matmul_nt(gy2p, wT, amax=pg)[:, :K].reshape(xshape)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy2.sum(0)
        fork.join()
        return gx, gw, gb


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        B, C, H, W = x.shape
        O, _, KH, KW = weight.shape
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        xh = x.permute(0, 2, 3, 1) if ops.is_cl(x) else ops.nchw_to_nhwc(x.contiguous())
        if C % 4:                                            # tiny-Cin first layer: pad channels to 4
            Cp = (C + 3) // 4 * 4
            xp = torch.zeros(B, H, W, Cp, dtype=x.dtype, device=x.device)
            xp[..., :C] = xh
            wp = torch.zeros(O, Cp, KH, KW, dtype=x.dtype, device=x.device)
            wp[:, :C] = weight
            xh, wsrc = xp, wp
        else:
            Cp, wsrc = C, weight
        one = KH == 1 and KW == 1 and stride == 1 and pad == 0
        w4 = wsrc.permute(0, 2, 3, 1).contiguous()                                  # [O][KH][KW][Cp]
        # forward as an implicit GEMM (edadm_conv2d_f32_nhwc): no im2col matrix, the gather re-reads x through L2;
        # the im2col matrix is only rebuilt in backward, for the weight gradient
        M = B * Ho * Wo
        px = None
        FLOPS[0] += 2.0 * M * O * KH * KW * C
        direct = (DIRECT_SMALL and F16X3 and KH == 3 and KW == 3 and stride == 1 and pad == 1 and ops.f16x3_conv_ok(xh, w4)
                  and ops.conv3_f16x3_direct_ok(B, H, W, Cp, O))
        if direct or ((M + 127) // 128) * ((O + 127) // 128) >= 128:
            if F16X3 and ops.f16x3_conv_ok(xh, w4):
                px = _amax(xh)
                out = ops.conv2d_f16x3_nhwc(xh, w4, bias, stride=stride, pad=pad, amax=px)   # [B][Ho][Wo][O]
            else:
                out = ops.conv2d_f32_nhwc(xh, w4, bias, stride=stride, pad=pad)
        else:                                                # few tiles (8x8 / 16x16 levels): im2col + split-K GEMM
            cols = xh.reshape(M, Cp) if one else ops.im2col_f32(xh, KH, KW, stride, pad, Ho, Wo)
            px = _amax(xh)                                   # max |cols| = max |x|
            out = _matmul_nt(cols, w4.reshape(O, KH * KW * Cp), bias, amax=px).reshape(B, Ho, Wo, O)
        ctx.save_for_backward(xh, w4.reshape(O, KH * KW * Cp))
        ctx.px = px if px is not None else _amax(xh)
        ctx.meta = (B, C, Cp, H, W, O, KH, KW, stride, pad, Ho, Wo, one, bias is not None)
        return out.permute(0, 3, 1, 2) if CHANNELS_LAST else ops.nhwc_to_nchw(out)

    @staticmethod
    def backward(ctx, gy):
        xh, w2 = ctx.saved_tensors
        B, C, Cp, H, W, O, KH, KW, stride, pad, Ho, Wo, one, has_bias = ctx.meta
        gyh = (gy.permute(0, 2, 3, 1) if ops.is_cl(gy) else ops.nchw_to_nhwc(gy.contiguous())).reshape(B * Ho * Wo, O)
        pg = _amax(gyh)
        gx = gw = gb = None
        FLOPS[0] += 2.0 * B * Ho * Wo * O * KH * KW * C * (int(ctx.needs_input_grad[0]) + int(ctx.needs_input_grad[1]))
        fork = _Fork(gyh, SIDE_WGRAD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1])
        if ctx.needs_input_grad[1]:
            with fork:
                if one:
                    gw2 = _wgrad(gyh, xh.reshape(B * H * W, Cp), amax_g=pg, amax_a=ctx.px)
                else:                                                                     # [O][KH*KW*Cp]
                    gw2 = _wgrad(gyh, None, amax_g=pg, amax_a=ctx.px, conv=(xh, (KH, KW, stride, pad, Ho, Wo)))
                gw = gw2.reshape(O, KH, KW, Cp)[..., :C].permute(0, 3, 1, 2).contiguous()
        if ctx.needs_input_grad[0]:
            M = B * Ho * Wo
            big = ((B * H * W + 127) // 128) * ((Cp + 127) // 128) >= 512      # few tiles: the split-K GEMM + col2im wins
            if (DIRECT_SMALL and F16X3 and KH == 3 and KW == 3 and stride == 1 and pad == 1 and O % 16 == 0
                    and ops.conv3_f16x3_direct_ok(B, Ho, Wo, O, Cp)):
                big = True                                   # the direct kernel does not need many tiles to beat im2col + col2im
            if IMPLICIT_DGRAD and (not one) and stride == 1 and KH == KW and 2 * pad == KH - 1 and O % 4 == 0 and big:
                # input gradient of a stride-1 "same" convolution = the same convolution of gy with the spatially
                # flipped, transposed filter: one implicit GEMM, no [M][K] gradient-of-columns matrix, no col2im pass
                wf = w2.reshape(O, KH, KW, Cp).flip(1, 2).permute(3, 1, 2, 0).contiguous()      # [Cp][KH][KW][O]
                gy4 = gyh.reshape(B, Ho, Wo, O)
                if F16X3 and ops.f16x3_conv_ok(gy4, wf):
                    dxh = ops.conv2d_f16x3_nhwc(gy4, wf, None, stride=1, pad=pad, amax=pg)
                else:
                    dxh = ops.conv2d_f32_nhwc(gy4, wf, None, stride=1, pad=pad)
            else:
                gyp, _ = _pad4(gyh)
                w2t, _ = _pad4(ops.transpose_f32(w2))                                 # [K][O]
                dcols = _matmul_nt(gyp, w2t, amax=pg if gyp is gyh else None)           # [M][K]
                dxh = dcols.reshape(B, H, W, Cp) if one else ops.col2im_f32(dcols, B, H, W, Cp, KH, KW, stride, pad, Ho, Wo)
            dxh = dxh[..., :C].contiguous() if Cp != C else dxh
            gx = dxh.permute(0, 3, 1, 2) if CHANNELS_LAST else ops.nhwc_to_nchw(dxh)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gyh.sum(0)
        fork.join()
        return gx, gw, gb, None, None


def linear(x, weight, bias=None):
    return _LinearFn.apply(x, weight, bias)


def conv2d(x, weight, bias=None, stride=1, padding=0):
    return _Conv2dFn.apply(x, weight, bias, int(stride), int(padding))


def conv1d_k1(x, weight, bias=None):
    """Conv1d with kernel 1 over [B][C][L] (attention qkv / proj_out of the legacy AttentionBlock)."""
    y = _LinearFn.apply(x.permute(0, 2, 1), weight[:, :, 0], bias)
    return y.permute(0, 2, 1)
