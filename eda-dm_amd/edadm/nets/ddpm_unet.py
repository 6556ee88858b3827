"""FP32 definition of the CIFAR-10 DDPM UNet the reference quantises (interface and
state_dict layout of ddim/models/diffusion.py:199-392: `Model(config)`, attributes `temb.dense`,
`conv_in`, `down[i].block/attn/downsample`, `mid.block_1/attn_1/block_2`, `up[i]...`, `norm_out`,
`conv_out`, and `config.split_shortcut`).  This module graph is what `qdiff.QuantModel` rewrites in
place; it is used directly only for the FP passes of calibration.  Quantised sampling does not run
this graph: it is compiled into the int8 executor of edadm/engine.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from edadm import train_ops as T     # GroupNorm (+ SiLU), softmax, attention products on libedadm.so (forward and backward)


def get_timestep_embedding(timesteps, embedding_dim):
    """sin | cos sinusoidal table (diffusion.py:6-24)."""
    assert timesteps.dim() == 1
    half = embedding_dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    arg = timesteps.float()[:, None] * freq.to(timesteps.device)[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def nonlinearity(x):
    return T.silu(x)                                # x * sigmoid(x) (diffusion.py:27-29)


class _GroupNorm(nn.GroupNorm):
    """nn.GroupNorm on the HIP forward / backward kernels (same parameters, same state_dict keys)."""

    def forward(self, x, silu=False):
        return T.group_norm(x, self, silu=silu)


def Normalize(in_channels):
    return _GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)   # asymmetric pad done in forward

    def forward(self, x):
        if self.with_conv:
            return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))
        return F.avg_pool2d(x, 2, 2)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x, temb=None, split=0):
        if temb is None:
            x, temb = x
        h = self.conv1(self.norm1(x, silu=True))
        h = h + self.temb_proj(nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.norm2(h, silu=True)))
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                x = self.conv_shortcut(x)
            else:
                x = self.nin_shortcut(x, split) if split != 0 else self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        qt, kt = T.transpose12(q.reshape(b, c, h * w)), T.transpose12(k.reshape(b, c, h * w))      # [b, hw, c]
        p = T.softmax(T.bmm_nt(qt, kt, int(c) ** (-0.5)))                  # softmax_j(q_i . k_j / sqrt(c))
        h_ = T.bmm_nt(v.reshape(b, c, h * w), p).reshape(b, c, h, w)      # h[c, i] = sum_j v[c, j] p[i, j]
        return x + self.proj_out(h_)


class Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.config.split_shortcut = False
        self.config.change_block_recon = False
        m = config.model
        ch, out_ch, ch_mult = m.ch, m.out_ch, tuple(m.ch_mult)
        self.ch, self.temb_ch = ch, ch * 4
        self.num_resolutions, self.num_res_blocks = len(ch_mult), m.num_res_blocks
        self.resolution, self.in_channels = config.data.image_size, m.in_channels
        if getattr(m, "type", "simple") == "bayesian":
            self.logvar = nn.Parameter(torch.zeros(config.diffusion.num_diffusion_timesteps))

        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(m.in_channels, ch, 3, 1, 1)

        res, in_mult = self.resolution, (1,) + ch_mult
        self.down = nn.ModuleList()
        cin = None
        for lvl in range(self.num_resolutions):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            cin, cout = ch * in_mult[lvl], ch * ch_mult[lvl]
            for _ in range(self.num_res_blocks):
                stage.block.append(ResnetBlock(in_channels=cin, out_channels=cout, temb_channels=self.temb_ch,
                                               dropout=m.dropout))
                cin = cout
                if res in m.attn_resolutions:
                    stage.attn.append(AttnBlock(cin))
            if lvl != self.num_resolutions - 1:
                stage.downsample = Downsample(cin, m.resamp_with_conv)
                res //= 2
            self.down.append(stage)

        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=cin, out_channels=cin, temb_channels=self.temb_ch, dropout=m.dropout)
        self.mid.attn_1 = AttnBlock(cin)
        self.mid.block_2 = ResnetBlock(in_channels=cin, out_channels=cin, temb_channels=self.temb_ch, dropout=m.dropout)

        self.up = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            cout, skip = ch * ch_mult[lvl], ch * ch_mult[lvl]
            for j in range(self.num_res_blocks + 1):
                if j == self.num_res_blocks:
                    skip = ch * in_mult[lvl]
                stage.block.append(ResnetBlock(in_channels=cin + skip, out_channels=cout, temb_channels=self.temb_ch,
                                               dropout=m.dropout))
                cin = cout
                if res in m.attn_resolutions:
                    stage.attn.append(AttnBlock(cin))
            if lvl != 0:
                stage.upsample = Upsample(cin, m.resamp_with_conv)
                res *= 2
            self.up.insert(0, stage)

        self.norm_out = Normalize(cin)
        self.conv_out = nn.Conv2d(cin, out_ch, 3, 1, 1)

    def forward(self, x, t=None, context=None):
        if t is None:
            x, t = x
        assert x.shape[2] == x.shape[3] == self.resolution
        temb = self.temb.dense[1](nonlinearity(self.temb.dense[0](get_timestep_embedding(t, self.ch))))
        hs = [self.conv_in(x)]
        for lvl, stage in enumerate(self.down):
            for j in range(self.num_res_blocks):
                h = stage.block[j](hs[-1], temb)
                if len(stage.attn) > 0:
                    h = stage.attn[j](h)
                hs.append(h)
            if lvl != self.num_resolutions - 1:
                hs.append(stage.downsample(hs[-1]))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(hs[-1], temb)), temb)
        for lvl in reversed(range(self.num_resolutions)):
            stage = self.up[lvl]
            for j in range(self.num_res_blocks + 1):
                if self.config.split_shortcut:
                    h = stage.block[j](torch.cat([h, hs.pop()], dim=1), temb, split=h.size(1))
                else:
                    h = stage.block[j](torch.cat([h, hs.pop()], dim=1), temb)
                if len(stage.attn) > 0:
                    h = stage.attn[j](h)
            if lvl != 0:
                h = stage.upsample(h)
        return self.conv_out(self.norm_out(h, silu=True))
