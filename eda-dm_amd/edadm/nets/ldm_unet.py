"""FP32 definition of the latent-diffusion UNet family the reference quantises (LDM-4 ImageNet,
LDM-8 LSUN, Stable-Diffusion v1): same constructor keywords and state_dict layout as
ldm/modules/diffusionmodules/openaimodel.py:447-783 (`UNetModel`), with the blocks of
openaimodel.py:74-406 and ldm/modules/attention.py:37-287 it is assembled from, so that
`qdiff.QuantModel` can rewrite it in place and reference checkpoints load unchanged.

Gradient checkpointing: the FP blocks below never recompute (a 288 GB MI355X keeps the activations; without
stochastic masks a recompute is transparent), so their `use_checkpoint` / `checkpoint` arguments only travel to the
quantised wrappers.  Those (qdiff/quant_block.py) DO honour the flag through `checkpoint` below, because the
reference's hand-written checkpoint (util.py:102-148) is not transparent during reconstruction: its backward re-runs
the block, every training-mode activation quantizer draws a NEW prob-mask in that second forward
(quant_layer.py:271-275), and the per-module loss sees outputs without a graph -- observable behaviour a drop-in has to
reproduce (fixture G18).  Quantised sampling does not execute this graph: edadm/engine.py compiles it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from edadm import train_ops as T     # GroupNorm (+ SiLU), LayerNorm, GEGLU, softmax, attention products on libedadm.so


# ----------------------------------------------------------------------------- small helpers
class _RecomputeInBackward(torch.autograd.Function):
    """forward: `fn` without a graph; backward: `fn` again WITH a graph (fresh quantizer masks), then the gradients of that
    second evaluation for the block inputs and the listed parameters."""

    @staticmethod
    def forward(ctx, fn, n_in, *args):
        ctx.fn, ctx.n_in, ctx.args = fn, n_in, args
        with torch.no_grad():
            return fn(*args[:n_in])

    @staticmethod
    def backward(ctx, *gout):
        n_in, args = ctx.n_in, ctx.args
        ins = [a.detach().requires_grad_(True) if torch.is_tensor(a) and a.is_floating_point() else a for a in args[:n_in]]
        with torch.enable_grad():
            out = ctx.fn(*ins)
        wrt = [a for a in ins if torch.is_tensor(a) and a.requires_grad] + [p for p in args[n_in:] if p.requires_grad]
        grads = iter(torch.autograd.grad(out, wrt, gout, allow_unused=True))
        gin = [next(grads) if torch.is_tensor(a) and a.requires_grad else None for a in ins]
        gpar = [next(grads) if p.requires_grad else None for p in args[n_in:]]
        ctx.args = None
        return (None, None) + tuple(gin) + tuple(gpar)


def checkpoint(func, inputs, params=None, flag=False):
    """util.py:102-115 of the reference.  Without autograd (sampling, activation caching) or without the flag: a plain call."""
    if not flag or not torch.is_grad_enabled():
        return func(*inputs)
    return _RecomputeInBackward.apply(func, len(inputs), *(tuple(inputs) + tuple(params or ())))


def conv_nd(dims, *args, **kwargs):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dims](*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def avg_pool_nd(dims, *args, **kwargs):
    return {1: nn.AvgPool1d, 2: nn.AvgPool2d, 3: nn.AvgPool3d}[dims](*args, **kwargs)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return T.group_norm(x, self)


class _GroupNorm(nn.GroupNorm):
    """nn.GroupNorm on the HIP forward / backward kernels (same parameters, same state_dict keys)."""

    def forward(self, x):
        return T.group_norm(x, self)


class _LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return T.layer_norm(x, self)


class _SiLU(nn.SiLU):
    def forward(self, x):
        return T.silu(x)


def norm_silu(seq, x):
    """(GroupNorm, SiLU) at the head of `seq` in ONE pass (the fused K5 of the training graph)."""
    assert isinstance(seq[0], nn.GroupNorm) and isinstance(seq[1], nn.SiLU)
    return T.group_norm(x, seq[0], silu=True)


def normalization(channels):
    return GroupNorm32(32, channels)


def Normalize(in_channels):
    return _GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


_FREQS = {}


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """cos | sin sinusoidal table (util.py:151-171).  The frequency vector is evaluated on the host exactly as the reference does
    (torch.exp on CPU, then moved) but ONCE per (dim, period, device): a host-to-device copy per forward is what kept the FP
    forward out of a HIP graph (scripts/calibration.py: the TDAC trajectories replay one)."""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    half = dim // 2
    key = (half, max_period, str(timesteps.device))
    if key not in _FREQS:
        _FREQS[key] = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
    freqs = _FREQS[key]
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def exists(v):
    return v is not None


def default(v, d):
    return v if v is not None else (d() if callable(d) else d)


# ----------------------------------------------------------------------------- transformer pieces
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return T.geglu(self.proj(x))                # a * gelu(gate), (a | gate) = proj(x).chunk(2, -1)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, default(dim_out, dim)))

    def forward(self, x):
        return self.net(x)


class CrossQKMatMul(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, q, k):
        return T.bmm_nt(q, k, self.scale)           # einsum("bid,bjd->bij") * scale


class CrossSMVMatMul(nn.Module):
    def forward(self, attn, v):
        return T.bmm_nt(attn, T.transpose12(v))     # einsum("bij,bjd->bid")


def split_heads(t, h):
    b, n, hd = t.shape
    return t.reshape(b, n, h, hd // h).permute(0, 2, 1, 3).reshape(b * h, n, hd // h)


def merge_heads(t, h):
    bh, n, d = t.shape
    return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.qk_matmul = CrossQKMatMul(self.scale)
        self.smv_matmul = CrossSMVMatMul()
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def forward(self, x, context=None, mask=None):
        context = default(context, x)
        q, k, v = (split_heads(t, self.heads) for t in (self.to_q(x), self.to_k(context), self.to_v(context)))
        sim = self.qk_matmul(q, k)
        if mask is not None:
            m = mask.reshape(mask.shape[0], -1)[:, None, :].repeat_interleave(self.heads, 0)
            sim.masked_fill_(~m, -torch.finfo(sim.dtype).max)
        out = self.smv_matmul(T.softmax(sim), v)
        return self.to_out(merge_heads(out, self.heads))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1, self.norm2, self.norm3 = _LayerNorm(dim), _LayerNorm(dim), _LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)
             for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, in_channels, 1))

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        t = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(b, h * w, -1)
        for blk in self.transformer_blocks:
            t = blk(t, context)
        t = t.reshape(b, h, w, -1).permute(0, 3, 1, 2)
        return self.proj_out(t) + x


# ----------------------------------------------------------------------------- UNet pieces
class TimestepBlock(nn.Module):
    """Marker: forward(x, emb)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb, context=None, split=0):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb, split=split)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        if use_conv:
            self.conv = conv_nd(dims, channels, self.out_channels, 3, padding=padding)

    def forward(self, x):
        assert x.shape[1] == self.channels
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        if use_conv:
            self.op = conv_nd(dims, channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert channels == self.out_channels
            self.op = avg_pool_nd(dims, kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.op(x)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint = use_conv, use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), _SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False, dims), Upsample(channels, False, dims)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False, dims), Downsample(channels, False, dims)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(
            _SiLU(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), _SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def forward(self, x, emb, split=0):
        return resblock_forward(self, x, emb, split)


def resblock_forward(blk, x, emb, split=0):
    """Shared by ResBlock and qdiff.QuantResBlock (openaimodel.py:248-278, quant_block.py:86-116)."""
    if blk.updown:
        h = norm_silu(blk.in_layers, x)
        h, x = blk.h_upd(h), blk.x_upd(x)
        h = blk.in_layers[-1](h)
    else:
        h = blk.in_layers[-1](norm_silu(blk.in_layers, x))
    e = blk.emb_layers(emb).type(h.dtype)
    while e.dim() < h.dim():
        e = e[..., None]
    if blk.use_scale_shift_norm:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = blk.out_layers[1:](blk.out_layers[0](h) * (1 + scale) + shift)
    else:
        h = blk.out_layers[2:](norm_silu(blk.out_layers, h + e))
    if split > 0:
        return blk.skip_connection(x, split=split) + h
    return blk.skip_connection(x) + h


class QKMatMul(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale = None

    def forward(self, q, k):
        return T.bmm_nt(T.transpose12(q * self.scale), T.transpose12(k * self.scale))     # einsum("bct,bcs->bts")


class SMVMatMul(nn.Module):
    def forward(self, weight, v):
        return T.bmm_nt(v, weight)                  # einsum("bts,bcs->bct")


class QKVAttentionLegacy(nn.Module):
    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads
        self.qkv_matmul, self.smv_matmul = QKMatMul(), SMVMatMul()

    def forward(self, qkv):
        bs, width, length = qkv.shape
        ch = width // (3 * self.n_heads)
        q, k, v = qkv.reshape(bs * self.n_heads, ch * 3, length).split(ch, dim=1)
        self.qkv_matmul.scale = 1 / math.sqrt(math.sqrt(ch))
        weight = T.softmax(self.qkv_matmul(q, k))
        return self.smv_matmul(weight, v).reshape(bs, -1, length)


class QKVAttention(nn.Module):
    """The new attention order (q | k | v split BEFORE the heads, openaimodel.py:413-444).  No shipped configuration sets
    `use_new_attention_order`, the reference's quantisation hooks wrap QKVAttentionLegacy only (quant_block.py:119-162), and the int8
    executor has no graph for it -- so the products run on the library's kernels (T.bmm_nt / T.softmax, the same operators the legacy
    class uses, differentiable for the reconstruction loop) and the frozen engine refuses the module (edadm/engine.py) rather than
    falling back to stock torch operators silently."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def forward(self, qkv):
        bs, width, length = qkv.shape
        ch = width // (3 * self.n_heads)
        q, k, v = qkv.chunk(3, dim=1)
        scale = 1 / math.sqrt(math.sqrt(ch))
        q = (q * scale).reshape(bs * self.n_heads, ch, length)
        k = (k * scale).reshape(bs * self.n_heads, ch, length)
        w = T.softmax(T.bmm_nt(T.transpose12(q), T.transpose12(k)))              # einsum("bct,bcs->bts")
        return T.bmm_nt(v.reshape(bs * self.n_heads, ch, length), w).reshape(bs, -1, length)   # einsum("bts,bcs->bct")


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False,
                 use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.use_checkpoint = use_checkpoint
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.attention = QKVAttention(self.num_heads) if use_new_attention_order else QKVAttentionLegacy(self.num_heads)
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))

    def forward(self, x):
        b, c, *spatial = x.shape
        xf = x.reshape(b, c, -1)
        h = self.proj_out(self.attention(self.qkv(self.norm(xf))))
        return (xf + h).reshape(b, c, *spatial)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        if use_spatial_transformer:
            assert context_dim is not None
        if context_dim is not None:
            assert use_spatial_transformer
            if not isinstance(context_dim, int):
                context_dim = list(context_dim) if hasattr(context_dim, "__iter__") else int(context_dim)
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        assert num_heads != -1 or num_head_channels != -1
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_resolutions, self.dropout, self.channel_mult = attention_resolutions, dropout, channel_mult
        self.conv_resample, self.num_classes, self.use_checkpoint = conv_resample, num_classes, use_checkpoint
        self.dtype = torch.float32
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = n_embed is not None
        self.split_shortcut = False
        mc, ted = model_channels, model_channels * 4
        self.time_embed = nn.Sequential(linear(mc, ted), _SiLU(), linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)

        def res(cin, cout, **kw):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(ch, heads_arg):
            nonlocal num_heads
            if num_head_channels == -1:
                dim_head = ch // num_heads
            else:
                num_heads = ch // num_head_channels
                dim_head = num_head_channels
            if legacy:
                dim_head = ch // num_heads if use_spatial_transformer else num_head_channels
            if use_spatial_transformer:
                return SpatialTransformer(ch, num_heads, dim_head, depth=transformer_depth, context_dim=context_dim)
            return AttentionBlock(ch, use_checkpoint=use_checkpoint,
                                  num_heads=num_heads if heads_arg is None else heads_arg,
                                  num_head_channels=dim_head, use_new_attention_order=use_new_attention_order)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, mc, 3, padding=1))])
        chans, ch, ds = [mc], mc, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * mc)]
                ch = mult * mc
                if ds in attention_resolutions:
                    layers.append(attn(ch, None))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, dims=dims,
                                                                              out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, None), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + chans.pop(), mc * mult)]
                ch = mc * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown
                                  else Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), _SiLU(),
                                 zero_module(conv_nd(dims, mc, out_channels, 3, padding=1)))
        if self.predict_codebook_ids:
            self.id_predictor = nn.Sequential(normalization(ch), conv_nd(dims, mc, n_embed, 1))

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        assert (y is not None) == (self.num_classes is not None)
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        if self.num_classes is not None:
            emb = emb + self.label_emb(y)
        hs, h = [], x.type(self.dtype)
        for module in self.input_blocks:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for module in self.output_blocks:
            split = h.shape[1] if self.split_shortcut else 0
            h = module(torch.cat([h, hs.pop()], dim=1), emb, context, split=split)
        h = h.type(x.dtype)
        return self.id_predictor(h) if self.predict_codebook_ids else self.out[2:](norm_silu(self.out, h))
