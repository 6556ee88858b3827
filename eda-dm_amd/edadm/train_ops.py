"""Autograd front end of the H1 training-graph kernels (csrc/train_ops.hip): the non-contraction ops the reconstruction
loop differentiates through -- GroupNorm (+ SiLU), LayerNorm, GEGLU, SiLU, softmax and the attention products -- forward
and input gradient on libedadm.so instead of stock torch / hipBLASLt kernels (north_star: "fused GroupNorm+SiLU and softmax
... as HIP kernels").  The normalisation affines get no gradient: block_recon.py:44-108 never trains them.

Device tensors only (ops raise on host tensors: the product has no CPU compute path)."""
import ctypes

import torch
import torch.nn.functional as F

from . import lib, ops
from .ops import _pf, _stream


# ----------------------------------------------------------------------------- GroupNorm (+ SiLU), NCHW / [B, C, T]
class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, silu):
        x = x.float()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        stats = torch.empty(B * G, 2, dtype=torch.float32, device=x.device)
        cl = ops.is_cl(x) and C % 4 == 0 and C <= 1024 and B <= 65535
        if cl:
            # NHWC memory order (the convolutions' own layout, edadm/contract.py CHANNELS_LAST): chunked per-channel partials
            xm = x.permute(0, 2, 3, 1)
            y = torch.empty_like(xm)
            ws = ops.workspace(x.device, lib.load().edadm_gn_nhwc_ws_floats(B, HW, C, int(G)))
            lib.call("edadm_gn_fwd_nhwc", _pf(xm), _pf(g), _pf(b), _pf(y), _pf(stats), _pf(ws), B, C, HW, int(G), float(eps),
                     1 if silu else 0, _stream())
            ctx.save_for_backward(xm, g, b, stats)
            ctx.meta = (B, C, HW, int(G), bool(silu), True)
            return y.permute(0, 3, 1, 2)
        x = x.contiguous()
        y = torch.empty_like(x)
        lib.call("edadm_gn_fwd_nchw", _pf(x), _pf(g), _pf(b), _pf(y), _pf(stats), B, C, HW, int(G), float(eps), 1 if silu else 0,
                 _stream())
        ctx.save_for_backward(x, g, b, stats)
        ctx.meta = (B, C, HW, int(G), bool(silu), False)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, stats = ctx.saved_tensors
        B, C, HW, G, silu, cl = ctx.meta
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if cl:
                dym = ops.mem_like(dy, True)                 # no copy when the gradient arrives in NHWC order
                ws = ops.workspace(x.device, lib.load().edadm_gn_nhwc_ws_floats(B, HW, C, G))
                lib.call("edadm_gn_bwd_nhwc", _pf(dym), _pf(x), _pf(g), _pf(b), _pf(stats), _pf(dx), _pf(ws), B, C, HW, G,
                         1 if silu else 0, _stream())
                dx = dx.permute(0, 3, 1, 2)
            else:
                dy = dy.contiguous()
                lib.call("edadm_gn_bwd_nchw", _pf(dy), _pf(x), _pf(g), _pf(b), _pf(stats), _pf(dx), B, C, HW, G, 1 if silu else 0,
                         _stream())
        return dx, None, None, None, None, None


def group_norm(x, norm, silu=False):
    """nn.GroupNorm `norm` applied to x [B, C, ...] (+ x * sigmoid(x) of the result in the same pass)."""
    return _GroupNormFn.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu)


# ----------------------------------------------------------------------------- LayerNorm over the last dimension
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        x2 = x.reshape(-1, C).contiguous().float()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = torch.empty_like(x2)
        stats = torch.empty(x2.shape[0], 2, dtype=torch.float32, device=x.device)
        lib.call("edadm_ln_fwd", _pf(x2), _pf(g), _pf(b), _pf(y), _pf(stats), x2.shape[0], C, float(eps), _stream())
        ctx.save_for_backward(x2, g, stats)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, stats = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            d2 = dy.reshape(x2.shape).contiguous()
            dx = torch.empty_like(x2)
            lib.call("edadm_ln_bwd", _pf(d2), _pf(x2), _pf(g), _pf(stats), _pf(dx), x2.shape[0], x2.shape[1], _stream())
            dx = dx.reshape(dy.shape)
        return dx, None, None, None


def layer_norm(x, ln):
    return _LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)


# ----------------------------------------------------------------------------- GEGLU: a * gelu(gate), (a | gate) = halves of a row
class _GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        inner = h.shape[-1] // 2
        h2 = h.reshape(-1, 2 * inner).contiguous().float()
        out = torch.empty(h2.shape[0], inner, dtype=torch.float32, device=h.device)
        lib.call("edadm_geglu_fwd", _pf(h2), _pf(out), h2.shape[0], inner, _stream())
        ctx.save_for_backward(h2)
        ctx.shape = h.shape
        return out.reshape(tuple(h.shape[:-1]) + (inner,))

    @staticmethod
    def backward(ctx, dy):
        (h2,) = ctx.saved_tensors
        inner = h2.shape[1] // 2
        d2 = dy.reshape(-1, inner).contiguous()
        dh = torch.empty_like(h2)
        lib.call("edadm_geglu_bwd", _pf(d2), _pf(h2), _pf(dh), h2.shape[0], inner, _stream())
        return dh.reshape(ctx.shape)


def geglu(h):
    return _GegluFn.apply(h)


# ----------------------------------------------------------------------------- SiLU
class _SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x, cl = ops.mem_view(x.float())
        ctx.save_for_backward(x)
        ctx.cl = cl
        return ops.mem_restore(ops.silu(x), cl)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        lib.call("edadm_silu_bwd", _pf(ops.mem_like(dy, ctx.cl)), _pf(x), _pf(dx), x.numel(), _stream())
        return ops.mem_restore(dx, ctx.cl)


def silu(x):
    return _SiluFn.apply(x)


# ----------------------------------------------------------------------------- softmax over the last dimension
class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s):
        cols = s.shape[-1]
        s2 = s.reshape(-1, cols).contiguous().float()
        if cols % 4 == 0 and cols <= 4096:
            p = ops.softmax_f32(s2)
        else:
            p = torch.empty_like(s2)
            lib.call("edadm_softmax_fwd_any", _pf(s2), _pf(p), s2.shape[0], cols, _stream())
        ctx.save_for_backward(p)
        return p.reshape(s.shape)

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        d2 = dp.reshape(p.shape).contiguous()
        dx = torch.empty_like(p)
        lib.call("edadm_softmax_bwd", _pf(d2), _pf(p), _pf(dx), p.shape[0], p.shape[1], _stream())
        return dx.reshape(dp.shape)


def softmax(s):
    return _SoftmaxFn.apply(s)


# ----------------------------------------------------------------------------- attention products
def _transpose_raw(x):
    Z, R, C = x.shape
    out = torch.empty(Z, C, R, dtype=torch.float32, device=x.device)
    for z0 in range(0, Z, 65535):
        zs = min(65535, Z - z0)
        lib.call("edadm_transpose_batched_f32", _pf(x[z0:z0 + zs]), _pf(out[z0:z0 + zs]), zs, R, C, _stream())
    return out


class _TransposeFn(torch.autograd.Function):
    """[Z, R, C] -> [Z, C, R] (contiguous)."""

    @staticmethod
    def forward(ctx, x):
        return _transpose_raw(x.contiguous().float())

    @staticmethod
    def backward(ctx, dy):
        return _transpose_raw(dy.contiguous())


def transpose12(x):
    return _TransposeFn.apply(x)


def _nt(a, b, alpha=1.0):
    """a [Z, M, K], b [Z, N, K] (contiguous) -> alpha * a . b^T [Z, M, N] on the exact-fp32 MFMA (edadm_gemm_f32_nt)."""
    Z, M, K = a.shape
    N = b.shape[1]
    from . import contract
    contract.FLOPS[0] += 2.0 * Z * M * N * K
    if K % 4:                                       # the staging moves 16-byte pieces: zero columns add nothing
        a, b = F.pad(a, (0, 4 - K % 4)), F.pad(b, (0, 4 - K % 4))
        K = a.shape[2]
    out = torch.empty(Z, M, N, dtype=torch.float32, device=a.device)
    ops.gemm_f32_nt(a, b, M, N, K, alpha=alpha, out=out, batch=Z, strideA=M * K, strideB=N * K, strideC=M * N)
    return out


class _BmmNtFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = a.contiguous().float(), b.contiguous().float()
        ctx.save_for_backward(a, b)
        ctx.alpha = float(alpha)
        return _nt(a, b, alpha)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _nt(g, _transpose_raw(b), ctx.alpha)                    # [Z, M, N] . [Z, K, N]^T
        if ctx.needs_input_grad[1]:
            db = _nt(_transpose_raw(g), _transpose_raw(a), ctx.alpha)    # [Z, N, M] . [Z, K, M]^T
        return da, db, None


def bmm_nt(a, b, alpha=1.0):
    """Batched a . b^T with autograd: a [Z, M, K], b [Z, N, K] -> [Z, M, N]."""
    return _BmmNtFn.apply(a, b, alpha)
