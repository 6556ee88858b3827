"""Quantised block wrappers — API of the reference's qdiff/quant_block.py.  Each wrapper re-uses
the FP block's sub-modules (already turned into QuantModules by QuantModel) and adds the
activation quantizers of the attention products (q, k, v and the softmax output, `sm_abit`).
These module graphs run the FP and fake-quant passes of calibration; sampling runs the compiled
int8 executor (edadm/engine.py), which reads the same modules for its parameters."""
import logging
from types import MethodType

import torch as th
import torch.nn as nn

from qdiff.quant_layer import QuantModule, UniformAffineQuantizer, StraightThrough
from edadm.nets.ldm_unet import (AttentionBlock, ResBlock, TimestepBlock, QKMatMul, SMVMatMul,
                                 BasicTransformerBlock, resblock_forward, split_heads, merge_heads, checkpoint)
from edadm.nets.ddpm_unet import ResnetBlock, AttnBlock, nonlinearity
from edadm import train_ops as T

logger = logging.getLogger(__name__)


class BaseQuantBlock(nn.Module):
    def __init__(self, act_quant_params: dict = {}):
        super().__init__()
        self.use_weight_quant = False
        self.use_act_quant = False
        self.can_recon = True
        self.split = 0
        self.act_quantizer = UniformAffineQuantizer(**act_quant_params)
        self.activation_function = StraightThrough()
        self.ignore_reconstruction = False

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, QuantModule):
                m.set_quant_state(weight_quant, act_quant)


class QuantResBlock(BaseQuantBlock, TimestepBlock):
    """LDM ResBlock (quant_block.py:46-116)."""

    def __init__(self, res: ResBlock, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        for name in ("channels", "emb_channels", "dropout", "out_channels", "use_conv", "use_checkpoint",
                     "use_scale_shift_norm", "updown"):
            setattr(self, name, getattr(res, name))
        self.in_layers = res.in_layers
        self.h_upd, self.x_upd = res.h_upd, res.x_upd
        self.emb_layers, self.out_layers = res.emb_layers, res.out_layers
        self.skip_connection = res.skip_connection
        self.split = 0

    def forward(self, x, emb=None, split=0):
        # the split only travels down while the skip convolution has not been split yet (:72-84)
        first = split != 0 and not isinstance(self.skip_connection, nn.Identity) and self.skip_connection.split == 0
        if emb is None:
            x, emb = x
        if first:
            self.split = split
        sp = self.split if first else 0
        # quant_block.py:78-84: checkpointed when the config says so (Stable Diffusion: use_checkpoint True)
        return checkpoint(lambda a, b: resblock_forward(self, a, b, sp), (x, emb), list(self.parameters()), self.use_checkpoint)


class QuantQKMatMul(BaseQuantBlock):
    def __init__(self, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.scale = None
        self.use_act_quant = False
        self.act_quantizer_q = UniformAffineQuantizer(**act_quant_params)
        self.act_quantizer_k = UniformAffineQuantizer(**act_quant_params)

    def forward(self, q, k):
        if self.use_act_quant:                      # einsum("bct,bcs->bts")
            return T.bmm_nt(T.transpose12(self.act_quantizer_q(q * self.scale)), T.transpose12(self.act_quantizer_k(k * self.scale)))
        return T.bmm_nt(T.transpose12(q * self.scale), T.transpose12(k * self.scale))

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_act_quant = act_quant


class QuantSMVMatMul(BaseQuantBlock):
    def __init__(self, act_quant_params: dict = {}, sm_abit=8):
        super().__init__(act_quant_params)
        self.use_act_quant = False
        self.act_quantizer_v = UniformAffineQuantizer(**act_quant_params)
        pw = act_quant_params.copy()
        pw['n_bits'], pw['symmetric'], pw['always_zero'] = sm_abit, False, True
        self.act_quantizer_w = UniformAffineQuantizer(**pw)

    def forward(self, weight, v):
        if self.use_act_quant:                      # einsum("bts,bcs->bct")
            return T.bmm_nt(self.act_quantizer_v(v), self.act_quantizer_w(weight))
        return T.bmm_nt(v, weight)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_act_quant = act_quant


class QuantAttentionBlock(BaseQuantBlock):
    def __init__(self, attn: AttentionBlock, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.channels, self.num_heads, self.use_checkpoint = attn.channels, attn.num_heads, attn.use_checkpoint
        self.norm, self.qkv, self.attention, self.proj_out = attn.norm, attn.qkv, attn.attention, attn.proj_out

    def forward(self, x):
        # quant_block.py:180-182: checkpointed with the flag hard-wired to True (see edadm/nets/ldm_unet.py `checkpoint`)
        return checkpoint(self._forward, (x,), list(self.parameters()), True)

    def _forward(self, x):
        b, c, *spatial = x.shape
        xf = x.reshape(b, c, -1)
        h = self.proj_out(self.attention(self.qkv(self.norm(xf))))
        return (xf + h).reshape(b, c, *spatial)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, (QuantModule, QuantQKMatMul, QuantSMVMatMul)):
                m.set_quant_state(weight_quant, act_quant)


def cross_attn_forward(self, x, context=None, mask=None):
    """CrossAttention.forward with the four attention quantizers (quant_block.py:204-235)."""
    h = self.heads
    context = x if context is None else context
    q, k, v = (split_heads(t, h) for t in (self.to_q(x), self.to_k(context), self.to_v(context)))
    if self.use_act_quant:                          # einsum('bid,bjd->bij') * scale
        sim = T.bmm_nt(self.act_quantizer_q(q), self.act_quantizer_k(k), self.scale)
    else:
        sim = T.bmm_nt(q, k, self.scale)
    if mask is not None:
        m = mask.reshape(mask.shape[0], -1)[:, None, :].repeat_interleave(h, 0)
        sim.masked_fill_(~m, -th.finfo(sim.dtype).max)
    attn = T.softmax(sim)
    if self.use_act_quant:                          # einsum('bij,bjd->bid')
        out = T.bmm_nt(self.act_quantizer_w(attn), T.transpose12(self.act_quantizer_v(v)))
    else:
        out = T.bmm_nt(attn, T.transpose12(v))
    return self.to_out(merge_heads(out, h))


class QuantBasicTransformerBlock(BaseQuantBlock):
    def __init__(self, tran: BasicTransformerBlock, act_quant_params: dict = {}, sm_abit: int = 8):
        super().__init__(act_quant_params)
        self.attn1, self.ff, self.attn2 = tran.attn1, tran.ff, tran.attn2
        self.norm1, self.norm2, self.norm3 = tran.norm1, tran.norm2, tran.norm3
        self.checkpoint = tran.checkpoint
        pw = act_quant_params.copy()
        pw['n_bits'], pw['always_zero'] = sm_abit, True
        for a in (self.attn1, self.attn2):
            a.act_quantizer_q = UniformAffineQuantizer(**act_quant_params)
            a.act_quantizer_k = UniformAffineQuantizer(**act_quant_params)
            a.act_quantizer_v = UniformAffineQuantizer(**act_quant_params)
        self.attn1.act_quantizer_w = UniformAffineQuantizer(**pw)
        self.attn2.act_quantizer_w = UniformAffineQuantizer(**pw)
        for a in (self.attn1, self.attn2):
            a.forward = MethodType(cross_attn_forward, a)
            a.use_act_quant = False

    def forward(self, x, context=None):
        if context is None and isinstance(x, (tuple, list)):
            x, context = x
        # quant_block.py:275: checkpointed unless QuantModel.set_grad_ckpt(False) (which all conditional scripts call)
        return checkpoint(self._forward, (x, context), list(self.parameters()), self.checkpoint)

    def _forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.attn1.use_act_quant = act_quant
        self.attn2.use_act_quant = act_quant
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, QuantModule):
                m.set_quant_state(weight_quant, act_quant)


class QuantResnetBlock(BaseQuantBlock):
    """DDPM ResnetBlock (quant_block.py:300-348)."""

    def __init__(self, res: ResnetBlock, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.in_channels, self.out_channels = res.in_channels, res.out_channels
        self.use_conv_shortcut = res.use_conv_shortcut
        self.norm1, self.conv1, self.temb_proj = res.norm1, res.conv1, res.temb_proj
        self.norm2, self.dropout, self.conv2 = res.norm2, res.dropout, res.conv2
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = res.conv_shortcut
            else:
                self.nin_shortcut = res.nin_shortcut
        self.split = 0

    def forward(self, x, temb=None, split=0):
        if split != 0:
            self.split = split
        h = self.conv1(self.norm1(x, silu=True))
        h = h + self.temb_proj(nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.norm2(h, silu=True)))
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x, split=self.split)
        return x + h


class QuantAttnBlock(BaseQuantBlock):
    """DDPM single-head AttnBlock (quant_block.py:398-451)."""

    def __init__(self, attn: AttnBlock, act_quant_params: dict = {}, sm_abit=8):
        super().__init__(act_quant_params)
        self.in_channels = attn.in_channels
        self.norm, self.q, self.k, self.v, self.proj_out = attn.norm, attn.q, attn.k, attn.v, attn.proj_out
        self.act_quantizer_q = UniformAffineQuantizer(**act_quant_params)
        self.act_quantizer_k = UniformAffineQuantizer(**act_quant_params)
        self.act_quantizer_v = UniformAffineQuantizer(**act_quant_params)
        pw = act_quant_params.copy()
        pw['n_bits'] = sm_abit
        self.act_quantizer_w = UniformAffineQuantizer(**pw)

    def forward(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        q = T.transpose12(q.reshape(b, c, h * w))                  # [b, hw, c]
        k = k.reshape(b, c, h * w)
        if self.use_act_quant:
            q, k = self.act_quantizer_q(q), self.act_quantizer_k(k)
        p = T.softmax(T.bmm_nt(q, T.transpose12(k), int(c) ** (-0.5)))      # [b, i, j]: the reference's w_ before its permute
        v = v.reshape(b, c, h * w)
        if self.use_act_quant:
            # the reference quantises w_.permute(0, 2, 1): a per-tensor quantiser, so the codes are those of p
            v, p = self.act_quantizer_v(v), self.act_quantizer_w(p)
        h_ = T.bmm_nt(v, p).reshape(b, c, h, w)                   # bmm(v, w_): h[c, i] = sum_j v[c, j] p[i, j]
        return x + self.proj_out(h_)


def get_specials(quant_act=False):
    specials = {ResBlock: QuantResBlock, BasicTransformerBlock: QuantBasicTransformerBlock,
                ResnetBlock: QuantResnetBlock, AttnBlock: QuantAttnBlock}
    if quant_act:
        specials[QKMatMul] = QuantQKMatMul
        specials[SMVMatMul] = QuantSMVMatMul
    else:
        specials[AttentionBlock] = QuantAttentionBlock
    return specials
