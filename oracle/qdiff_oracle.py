"""ORACLE — test infrastructure only.  CPU restatement (PyTorch-CPU float32 tensor arithmetic,
explicit closed-form gradients) of the reference's hot path: the uniform-affine / AdaRound
quantizers, the quantized layer and block graphs, the DDPM and LDM UNets, the block/layer
reconstruction loop and the DDIM stepping maths.

Nothing in the product imports this file: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may.  Parity status: PINNED — every function below is checked against golden
vectors captured by importing the reference itself (tests/golden/make_golden.py ->
tests/golden/g*.npz; tests/test_oracle_golden.py).

Each function cites the reference file:line it restates (paths relative to the reference root).
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

EPS = torch.tensor(1e-8, dtype=torch.float32)


# ======================================================================================
# K3 — scale / zero-point search                           qdiff/quant_layer.py:95-226
# ======================================================================================
def calculate_qparams(min_val, max_val, n_levels):
    """quant_layer.py:95-105."""
    quant_min, quant_max = 0, n_levels - 1
    min_neg = torch.min(min_val, torch.zeros_like(min_val))
    max_pos = torch.max(max_val, torch.zeros_like(max_val))
    scale = (max_pos - min_neg) / float(quant_max - quant_min)
    scale = torch.max(scale, EPS)
    zp = quant_min - torch.round(min_neg / scale)
    zp = torch.clamp(zp, quant_min, quant_max)
    return scale, zp


def quantize_minmax(x, x_max, x_min, n_levels, channel_wise):
    """quant_layer.py:108-118."""
    delta, zp = calculate_qparams(x_min, x_max, n_levels)
    if channel_wise:
        shp = [1] * x.dim()
        shp[0] = x.shape[0]
        delta, zp = delta.reshape(shp), zp.reshape(shp)
    x_int = torch.round(x / delta)
    x_q = torch.clamp(x_int + zp, 0, n_levels - 1)
    return (x_q - zp) * delta


def _score(x, xq, channel_wise):
    """quant_layer.py:87-93 (p = 2.4)."""
    e = (x - xq).abs().pow(2.4)
    return torch.flatten(e, 1).mean(1) if channel_wise else e.mean()


def search_1d(x, n_bits, one_side, channel_wise, num=100):
    """quant_layer.py:150-213: 100 clip thresholds; per-tensor path batches 8 candidates and
    takes argmin (first minimum); per-channel path keeps strict-< best."""
    n_levels = 2 ** n_bits
    if channel_wise:
        x_min, x_max = torch.aminmax(torch.flatten(x, 1), dim=1)
    else:
        x_min, x_max = torch.aminmax(x)
    xrange = torch.max(x_min.abs(), x_max)
    if not channel_wise:
        thres = xrange / num * torch.arange(1, num + 1)
        new_min = torch.zeros_like(thres) if one_side == "pos" else -thres
        new_max = torch.zeros_like(thres) if one_side == "neg" else thres
        scale = (new_max - new_min) / float(n_levels - 1)
        scale = torch.max(scale, EPS)
        zp = -torch.round(new_min / scale)
        zp = torch.clamp(zp, 0, n_levels - 1).view(-1, 1)
        scale = scale.view(-1, 1)
        scores = []
        xf = x.reshape(1, -1)
        for i in range(0, num, 8):
            x_int = (xf / scale[i:i + 8]).round()
            x_int = torch.max(torch.min(x_int, n_levels - 1 - zp[i:i + 8]), -zp[i:i + 8])
            x_sim = x_int * scale[i:i + 8]
            scores.append((x_sim - xf).abs().pow(2.4).mean(1))
        ind = torch.argmin(torch.hstack(scores))
        return new_min[ind], new_max[ind]
    best_score = torch.zeros_like(x_min) + 1e10
    best_min, best_max = x_min.clone(), x_max.clone()
    for i in range(1, num + 1):
        thres = xrange / num * i
        new_min = torch.zeros_like(x_min) if one_side == "pos" else -thres
        new_max = torch.zeros_like(x_max) if one_side == "neg" else thres
        xq = quantize_minmax(x, new_max, new_min, n_levels, channel_wise)
        score = _score(x, xq, channel_wise)
        better = score < best_score
        best_min = torch.where(better, new_min, best_min)
        best_max = torch.where(better, new_max, best_max)
        best_score = torch.min(score, best_score)
    return best_min, best_max


def search_2d(x, n_bits, channel_wise, num=100):
    """quant_layer.py:120-147: 100 ranges x 2^b zero points."""
    n_levels = 2 ** n_bits
    if channel_wise:
        x_min, x_max = torch.aminmax(torch.flatten(x, 1), dim=1)
        x_max = torch.max(x_max, torch.zeros_like(x_max))
        x_min = torch.min(x_min, torch.zeros_like(x_min))
    else:
        x_min, x_max = torch.aminmax(x)
    xrange = x_max - x_min
    best_score = torch.zeros_like(x_min) + 1e10
    best_min, best_max = x_min.clone(), x_max.clone()
    for i in range(1, num + 1):
        tmp_min = torch.zeros_like(x_min)
        tmp_max = xrange / num * i
        tmp_delta = (tmp_max - tmp_min) / (2 ** n_bits - 1)
        for zp in range(0, n_levels):
            new_min = tmp_min - zp * tmp_delta
            new_max = tmp_max - zp * tmp_delta
            xq = quantize_minmax(x, new_max, new_min, n_levels, channel_wise)
            score = _score(x, xq, channel_wise)
            better = score < best_score
            best_min = torch.where(better, new_min, best_min)
            best_max = torch.where(better, new_max, best_max)
            best_score = torch.min(best_score, score)
    return best_min, best_max


# ======================================================================================
# K1 — activation / weight fake-quant with explicit gradients   quant_layer.py:246-276
# ======================================================================================
def fake_quant_fwd(x, delta, zp, n_levels, mask=None):
    """quant_layer.py:266-276.  Returns (out, codes).  `mask` = (rand < prob) when training."""
    x_int = torch.round(x / delta) + zp
    codes = torch.clamp(x_int, 0, n_levels - 1)
    out = (codes - zp) * delta
    if mask is not None:
        out = torch.where(mask, out, x)
    return out, codes


def fake_quant_bwd(gy, x, delta, zp, n_levels, mask=None):
    """Closed form of autograd through quant_layer.py:19-23,266-276:
    round_ste passes 1, clamp passes where 0 <= x_int <= L (inclusive), so
      dx     = gy * inrange
      ddelta = sum gy * ((codes - zp) - inrange * x / delta)
    and with the prob mask only masked-in elements take the quantized branch."""
    xs = x / delta
    x_int = torch.round(xs) + zp
    inr = ((x_int >= 0) & (x_int <= n_levels - 1)).to(x.dtype)
    codes = torch.clamp(x_int, 0, n_levels - 1)
    # autograd order of operations: d(out)/d(codes) = gy*delta, then /delta through x/delta, so
    # gx = (gy*delta)/delta (not bit-identical to gy) and the delta path is -(gy*delta)*((x/delta)/delta)
    gq = gy * delta * inr
    gd_el = gy * (codes - zp) - gq * (xs / delta)
    gx = gq / delta
    if mask is not None:
        gx = torch.where(mask, gx, gy)
        gd_el = torch.where(mask, gd_el, torch.zeros_like(gd_el))
    if delta.numel() == 1:
        gd = gd_el.sum().reshape(delta.shape)
    else:
        dims = [i for i in range(x.dim()) if delta.shape[i] == 1]
        gd = gd_el.sum(dim=dims, keepdim=True)
    return gx, gd


class _FakeQuantFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, delta, zp, n_levels, mask):
        ctx.save_for_backward(x, delta, zp)
        ctx.n_levels, ctx.mask = n_levels, mask
        return fake_quant_fwd(x, delta, zp, n_levels, mask)[0]

    @staticmethod
    def backward(ctx, gy):
        x, delta, zp = ctx.saved_tensors
        gx, gd = fake_quant_bwd(gy, x, delta, zp, ctx.n_levels, ctx.mask)
        return gx, gd, None, None, None


class OQ:
    """Restatement of UniformAffineQuantizer (quant_layer.py:36-357), scale_method='mse' only."""

    def __init__(self, n_bits=8, symmetric=False, channel_wise=False, scale_method="mse", leaf_param=False,
                 always_zero=False, prob=1.0, name=""):
        if scale_method not in ("mse", "max"):
            raise NotImplementedError
        self.scale_method, self.always_zero = scale_method, always_zero
        self.name = name
        self.n_bits, self.n_levels = n_bits, 2 ** n_bits
        self.sym, self.channel_wise, self.leaf_param = symmetric, channel_wise, leaf_param
        self.delta = self.zero_point = None
        self.inited = False
        self.running_min = self.running_max = None
        self.one_side_dist = None
        self.prob, self.is_training = prob, False
        self.mask_fn = None     # test hook: returns the uniform tensor torch.rand_like would have

    def bitwidth_refactor(self, b):
        self.n_bits, self.n_levels = b, 2 ** b

    def get_min_max(self, x):
        """quant_layer.py:215-226 (+ EMA 79-85)."""
        if self.one_side_dist is None:
            self.one_side_dist = "pos" if x.min() >= 0.0 else "neg" if x.max() <= 0.0 else "no"
        if self.one_side_dist != "no" or self.sym:
            mn, mx = search_1d(x, self.n_bits, self.one_side_dist, self.channel_wise)
        else:
            mn, mx = search_2d(x, self.n_bits, self.channel_wise)
        if self.leaf_param:
            if self.running_min is None:
                self.running_min, self.running_max = mn, mx
            self.running_min = 0.1 * mn + 0.9 * self.running_min
            self.running_max = 0.1 * mx + 0.9 * self.running_max
            return self.running_min, self.running_max
        return mn, mx

    def max_rule(self, x):
        """scale_method='max', init_quantization_scale_2 (quant_layer.py:278-330): Python-float (double) arithmetic on the
        extrema of each channel (weights) or of the tensor; symmetric -> absmax / n_levels with zero_point 0."""
        xf = x.detach().reshape(x.shape[0], -1) if self.channel_wise else x.detach().reshape(1, -1)
        mn, mx = xf.min(1)[0].double().numpy(), xf.max(1)[0].double().numpy()
        lo, hi = np.minimum(mn, 0.0), np.maximum(mx, 0.0)
        delta = np.maximum(np.abs(lo), hi) / self.n_levels if self.sym else (mx - mn) / (self.n_levels - 1)
        delta = np.where(delta < 1e-8, 1e-8, delta)
        zp = np.zeros_like(delta) if (self.sym or self.always_zero) else np.round(-lo / delta)
        return torch.as_tensor(delta.astype(np.float32)), torch.as_tensor(zp.astype(np.float32) + 0.0)

    def init_scale(self, x):
        with torch.no_grad():
            if self.scale_method == "max":
                delta, zp = self.max_rule(x)
                if not self.channel_wise:
                    delta, zp = delta.reshape(()), zp.reshape(())
            else:
                mn, mx = self.get_min_max(x.detach())
                delta, zp = calculate_qparams(mn, mx, self.n_levels)
        if self.channel_wise:
            shp = [1] * x.dim()
            shp[0] = x.shape[0]
            delta, zp = delta.reshape(shp), zp.reshape(shp)
        self.delta, self.zero_point = delta.clone(), zp

    def __call__(self, x):
        if not self.inited:
            self.init_scale(x)
        mask = None
        if self.is_training and self.prob < 1.0:
            u = self.mask_fn(x) if self.mask_fn is not None else torch.rand_like(x)
            mask = u < self.prob
        return _FakeQuantFn.apply(x, self.delta, self.zero_point, self.n_levels, mask)


# ======================================================================================
# K2 — AdaRound                                       qdiff/adaptive_rounding.py:9-78
# ======================================================================================
GAMMA, ZETA = -0.1, 1.1


def adaround_init_alpha(w, delta):
    """adaptive_rounding.py:66-72."""
    x_floor = torch.floor(w / delta)
    rest = (w / delta) - x_floor
    return -torch.log((ZETA - GAMMA) / (rest - GAMMA) - 1)


def adaround_fwd(w, alpha, delta, zp, n_levels, soft):
    """adaptive_rounding.py:49-61,63-64."""
    x_floor = torch.floor(w / delta)
    if soft:
        h = torch.clamp(torch.sigmoid(alpha) * (ZETA - GAMMA) + GAMMA, 0, 1)
    else:
        h = (alpha >= 0).float()
    x_int = x_floor + h
    xq = torch.clamp(x_int + zp, 0, n_levels - 1)
    return (xq - zp) * delta


def adaround_bwd(gy, w, alpha, delta, zp, n_levels):
    """d out / d alpha through the soft path: delta * [0<=x_int+zp<=L] * [0<=s<=1] * (zeta-gamma) sig(1-sig)."""
    x_floor = torch.floor(w / delta)
    sig = torch.sigmoid(alpha)
    s = sig * (ZETA - GAMMA) + GAMMA
    h = torch.clamp(s, 0, 1)
    v = x_floor + h + zp
    inr = ((v >= 0) & (v <= n_levels - 1)).to(w.dtype)
    ins = ((s >= 0) & (s <= 1)).to(w.dtype)
    return gy * delta * inr * ins * (ZETA - GAMMA) * sig * (1 - sig)


class _AdaRoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, alpha, delta, zp, n_levels):
        ctx.save_for_backward(w, alpha, delta, zp)
        ctx.n_levels = n_levels
        return adaround_fwd(w, alpha, delta, zp, n_levels, True)

    @staticmethod
    def backward(ctx, gy):
        w, alpha, delta, zp = ctx.saved_tensors
        return None, adaround_bwd(gy, w, alpha, delta, zp, ctx.n_levels), None, None, None


class OAdaRound:
    """AdaRoundQuantizer (adaptive_rounding.py:9-78), round_mode 'learned_hard_sigmoid'."""

    def __init__(self, uaq, weight):
        self.name = uaq.name
        self.n_bits, self.n_levels, self.sym = uaq.n_bits, uaq.n_levels, uaq.sym
        self.delta, self.zero_point = uaq.delta, uaq.zero_point
        self.soft_targets = False
        self.alpha = adaround_init_alpha(weight.clone(), self.delta).requires_grad_(True)
        self.inited = True

    def __call__(self, w):
        if self.soft_targets:
            return _AdaRoundFn.apply(w, self.alpha, self.delta, self.zero_point, self.n_levels)
        return adaround_fwd(w, self.alpha.detach(), self.delta, self.zero_point, self.n_levels, False)


# ======================================================================================
# K7 — reconstruction loss; temperature decay   quant_layer.py:26-33, block_recon.py:305-323
# ======================================================================================
def lp_loss(pred, tgt, p=2.0, reduction="none"):
    if reduction == "none":
        return (pred - tgt).abs().pow(p).sum(1).mean()
    return (pred - tgt).abs().pow(p).mean()


def lp_loss_grad(pred, tgt):
    """d/dpred of lp_loss(p=2): 2 (pred - tgt) / (numel / C)."""
    denom = pred.numel() / pred.shape[1]
    return 2.0 * (pred - tgt) / denom


def linear_temp_decay(t, t_max, rel_start_decay, start_b, end_b):
    start_decay = rel_start_decay * t_max
    if t < start_decay:
        return start_b
    rel_t = (t - start_decay) / (t_max - start_decay)
    return end_b + (start_b - end_b) * max(0.0, (1 - rel_t))


# ======================================================================================
# K8 — Adam + cosine annealing (torch.optim.Adam / CosineAnnealingLR as used at
# block_recon.py:112-117,199-206; lr_t = lr0 * (1 + cos(pi t / T)) / 2, eta_min 0)
# ======================================================================================
class OAdam:
    def __init__(self, params, lr, t_max, betas=(0.9, 0.999), eps=1e-8):
        self.params, self.lr0, self.t_max = list(params), lr, t_max
        self.b1, self.b2, self.eps = betas[0], betas[1], eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0       # optimiser steps taken
        self.epoch = 0   # scheduler steps taken

    def lr(self):
        return self.lr0 * (1 + math.cos(math.pi * self.epoch / self.t_max)) / 2

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        self.t += 1
        lr = self.lr()
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        with torch.no_grad():
            for p, m, v in zip(self.params, self.m, self.v):
                if p.grad is None:
                    continue
                g = p.grad
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-(lr / bc1))
        self.epoch += 1


# ======================================================================================
# a4 — QuantModule                                         quant_layer.py:360-446
# ======================================================================================
class OLayer:
    def __init__(self, name, kind, weight, bias, wq, aq, stride=1, padding=0):
        self.name, self.kind = name, kind
        self.weight, self.bias = weight, bias
        self.stride, self.padding = stride, padding
        self.wq_params, self.aq_params = dict(wq), dict(aq)
        self.weight_quantizer = OQ(name=name + ".weight_quantizer", **wq)
        self.act_quantizer = OQ(name=name + ".act_quantizer", **aq)
        self.weight_quantizer_0 = self.act_quantizer_0 = None
        self.split = 0
        self.use_weight_quant = self.use_act_quant = False
        self.disable_act_quant = False
        self.hook_out = None
        self.record = False

    def quantizers(self):
        out = [self.weight_quantizer, self.act_quantizer]
        if self.split:
            out += [self.weight_quantizer_0, self.act_quantizer_0]
        return out

    def set_quant_state(self, w, a):
        self.use_weight_quant, self.use_act_quant = w, a

    def __call__(self, x, split=0):
        if split != 0 and self.split != 0:
            assert split == self.split
        elif split != 0:
            self.split = split
            self.weight_quantizer_0 = OQ(name=self.name + ".weight_quantizer_0", **self.wq_params)
            self.act_quantizer_0 = OQ(name=self.name + ".act_quantizer_0", **self.aq_params)
        if not self.disable_act_quant and self.use_act_quant:
            if self.split:
                x = torch.cat([self.act_quantizer(x[:, :self.split]), self.act_quantizer_0(x[:, self.split:])], 1)
            else:
                x = self.act_quantizer(x)
        if self.use_weight_quant:
            if self.split:
                w = torch.cat([self.weight_quantizer(self.weight[:, :self.split]),
                               self.weight_quantizer_0(self.weight[:, self.split:])], 1)
            else:
                w = self.weight_quantizer(self.weight)
        else:
            w = self.weight
        if self.kind == "conv2d":
            out = F.conv2d(x, w, self.bias, stride=self.stride, padding=self.padding)
        elif self.kind == "conv1d":
            out = F.conv1d(x, w, self.bias, stride=self.stride, padding=self.padding)
        else:
            out = F.linear(x, w, self.bias)
        if self.record:
            self.hook_out = out
        return out


def silu(x):
    return x * torch.sigmoid(x)


class ONorm:
    def __init__(self, w, b, eps, groups=32):
        self.w, self.b, self.eps, self.groups = w, b, eps, groups

    def __call__(self, x):
        return F.group_norm(x.float(), self.groups, self.w, self.b, self.eps)


class OLayerNorm:
    def __init__(self, w, b):
        self.w, self.b = w, b

    def __call__(self, x):
        return F.layer_norm(x, (x.shape[-1],), self.w, self.b, 1e-5)


class OBlock:
    """BaseQuantBlock (quant_block.py:20-43)."""
    is_block = True

    def __init__(self, name):
        self.name = name
        self.use_weight_quant = self.use_act_quant = False
        self.split = 0

    def layers(self):
        return []

    def extra_quantizers(self):
        return []

    def set_quant_state(self, w, a):
        self.use_weight_quant, self.use_act_quant = w, a
        for l in self.layers():
            l.set_quant_state(w, a)


class _Builder:
    def __init__(self, sd, wq, aq, sm_abit=8):
        self.sd, self.wq, self.aq, self.sm_abit = sd, wq, aq, sm_abit

    def t(self, key):
        return torch.as_tensor(np.asarray(self.sd[key])).float().clone()

    def has(self, key):
        return key in self.sd

    def layer(self, prefix, name, kind, stride=1, padding=0):
        b = self.t(prefix + ".bias") if self.has(prefix + ".bias") else None
        return OLayer(name, kind, self.t(prefix + ".weight"), b, self.wq, self.aq, stride, padding)

    def norm(self, prefix, eps):
        return ONorm(self.t(prefix + ".weight"), self.t(prefix + ".bias"), eps)

    def ln(self, prefix):
        return OLayerNorm(self.t(prefix + ".weight"), self.t(prefix + ".bias"))

    def aq_w(self, symmetric=None):
        p = dict(self.aq)
        p["n_bits"] = self.sm_abit
        if symmetric is not None:
            p["symmetric"] = symmetric
        return p


# --------------------------------------------------------------------------------------
# CIFAR DDPM blocks        quant_block.py:300-348 (QuantResnetBlock), :398-451 (QuantAttnBlock)
# --------------------------------------------------------------------------------------
class OResnetBlock(OBlock):
    def __init__(self, B, sdp, name, cin, cout):
        super().__init__(name)
        self.cin, self.cout = cin, cout
        self.norm1 = B.norm(sdp + ".norm1", 1e-6)
        self.conv1 = B.layer(sdp + ".conv1", name + ".conv1", "conv2d", 1, 1)
        self.temb_proj = B.layer(sdp + ".temb_proj", name + ".temb_proj", "linear")
        self.norm2 = B.norm(sdp + ".norm2", 1e-6)
        self.conv2 = B.layer(sdp + ".conv2", name + ".conv2", "conv2d", 1, 1)
        self.nin = B.layer(sdp + ".nin_shortcut", name + ".nin_shortcut", "conv2d", 1, 0) if cin != cout else None

    def layers(self):
        return [self.conv1, self.temb_proj, self.conv2] + ([self.nin] if self.nin else [])

    def __call__(self, x, temb, split=0):
        if split != 0:
            self.split = split
        h = self.conv1(silu(self.norm1(x)))
        h = h + self.temb_proj(silu(temb))[:, :, None, None]
        h = self.conv2(silu(self.norm2(h)))
        if self.nin is not None:
            x = self.nin(x, split=self.split)
        return x + h


class OAttnBlock(OBlock):
    def __init__(self, B, sdp, name, c):
        super().__init__(name)
        self.norm = B.norm(sdp + ".norm", 1e-6)
        self.q = B.layer(sdp + ".q", name + ".q", "conv2d")
        self.k = B.layer(sdp + ".k", name + ".k", "conv2d")
        self.v = B.layer(sdp + ".v", name + ".v", "conv2d")
        self.proj_out = B.layer(sdp + ".proj_out", name + ".proj_out", "conv2d")
        self.act_quantizer_q = OQ(name=name + ".act_quantizer_q", **B.aq)
        self.act_quantizer_k = OQ(name=name + ".act_quantizer_k", **B.aq)
        self.act_quantizer_v = OQ(name=name + ".act_quantizer_v", **B.aq)
        self.act_quantizer_w = OQ(name=name + ".act_quantizer_w", **B.aq_w())

    def layers(self):
        return [self.q, self.k, self.v, self.proj_out]

    def extra_quantizers(self):
        return [self.act_quantizer_q, self.act_quantizer_k, self.act_quantizer_v, self.act_quantizer_w]

    def __call__(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        q = q.reshape(b, c, h * w).permute(0, 2, 1)
        k = k.reshape(b, c, h * w)
        if self.use_act_quant:
            q, k = self.act_quantizer_q(q), self.act_quantizer_k(k)
        w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
        w_ = F.softmax(w_, dim=2)
        v = v.reshape(b, c, h * w)
        w_ = w_.permute(0, 2, 1)
        if self.use_act_quant:
            v, w_ = self.act_quantizer_v(v), self.act_quantizer_w(w_)
        h_ = torch.bmm(v, w_).reshape(b, c, h, w)
        return x + self.proj_out(h_)


def ddpm_timestep_embedding(t, dim):
    """ddim/models/diffusion.py:6-24 (sin | cos)."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = t.float()[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


class _Net:
    """Common traversal helpers of QuantModel (quant_model.py:14-95)."""

    def units(self):
        raise NotImplementedError

    def all_layers(self):
        out = []
        for kind, u in self.units():
            out += [u] if kind == "layer" else u.layers()
        return out

    def all_quantizers(self):
        """named_modules order: BaseQuantBlock's own (unused) act_quantizer is not modelled."""
        out = []
        for kind, u in self.units():
            if kind == "layer":
                out += u.quantizers()
            else:
                out += u.ordered_quantizers() if hasattr(u, "ordered_quantizers") else \
                    sum([l.quantizers() for l in u.layers()], []) + u.extra_quantizers()
        return out

    def set_quant_state(self, w, a):
        for kind, u in self.units():
            u.set_quant_state(w, a)

    def set_first_last_layer_to_8bit(self):
        """quant_model.py:77-88: first and last WEIGHT quantizer in module order (the first is the
        time-embedding Linear, not conv_in) and the second-to-last act quantizer."""
        ls = self.all_layers()
        ls[0].weight_quantizer.bitwidth_refactor(8)
        ls[-1].weight_quantizer.bitwidth_refactor(8)
        self.second_last_act_quantizer().bitwidth_refactor(8)

    def disable_network_output_quantization(self):
        self.all_layers()[-1].disable_act_quant = True

    def load_qparams(self, d, prefix="qp/model."):
        """Load (delta, zero_point, n_bits) captured from the reference; marks quantizers inited."""
        for q in self.all_quantizers():
            k = prefix + q.name
            if k + "/delta" in d:
                q.delta = torch.as_tensor(np.asarray(d[k + "/delta"])).float()
                q.zero_point = torch.as_tensor(np.asarray(d[k + "/zero_point"])).float()
                q.bitwidth_refactor(int(d[k + "/n_bits"]))
                q.inited = True


class ODDPM(_Net):
    """ddim/models/diffusion.py:199-392 wrapped as QuantModel does (quant_model.py:26-62)."""

    def __init__(self, sd, ch, ch_mult, nres, attn_res, res, wq, aq, sm_abit=8, in_ch=3):
        B = _Builder(sd, wq, aq, sm_abit)
        self.ch, self.nlev, self.nres, self.res = ch, len(ch_mult), nres, res
        self.split_shortcut = False
        self.dense0 = B.layer("temb.dense.0", "temb.dense.0", "linear")
        self.dense1 = B.layer("temb.dense.1", "temb.dense.1", "linear")
        self.conv_in = B.layer("conv_in", "conv_in", "conv2d", 1, 1)
        in_mult = (1,) + tuple(ch_mult)
        cur = res
        self.down = []
        for i in range(self.nlev):
            lvl = dict(block=[], attn=[], down=None)
            bin_, bout = ch * in_mult[i], ch * ch_mult[i]
            for j in range(nres):
                lvl["block"].append(OResnetBlock(B, "down.%d.block.%d" % (i, j), "down.%d.block.%d" % (i, j), bin_, bout))
                bin_ = bout
                if cur in attn_res:
                    lvl["attn"].append(OAttnBlock(B, "down.%d.attn.%d" % (i, j), "down.%d.attn.%d" % (i, j), bin_))
            if i != self.nlev - 1:
                lvl["down"] = B.layer("down.%d.downsample.conv" % i, "down.%d.downsample.conv" % i, "conv2d", 2, 0)
                cur //= 2
            self.down.append(lvl)
        self.mid1 = OResnetBlock(B, "mid.block_1", "mid.block_1", bin_, bin_)
        self.mid_attn = OAttnBlock(B, "mid.attn_1", "mid.attn_1", bin_)
        self.mid2 = OResnetBlock(B, "mid.block_2", "mid.block_2", bin_, bin_)
        self.up = [None] * self.nlev
        for i in reversed(range(self.nlev)):
            lvl = dict(block=[], attn=[], up=None)
            bout = ch * ch_mult[i]
            skip = ch * ch_mult[i]
            for j in range(nres + 1):
                if j == nres:
                    skip = ch * in_mult[i]
                lvl["block"].append(OResnetBlock(B, "up.%d.block.%d" % (i, j), "up.%d.block.%d" % (i, j), bin_ + skip, bout))
                bin_ = bout
                if cur in attn_res:
                    lvl["attn"].append(OAttnBlock(B, "up.%d.attn.%d" % (i, j), "up.%d.attn.%d" % (i, j), bin_))
            if i != 0:
                lvl["up"] = B.layer("up.%d.upsample.conv" % i, "up.%d.upsample.conv" % i, "conv2d", 1, 1)
                cur *= 2
            self.up[i] = lvl
        self.norm_out = B.norm("norm_out", 1e-6)
        self.conv_out = B.layer("conv_out", "conv_out", "conv2d", 1, 1)

    def units(self):
        """Execution order == recon order (recon_block_Qmodel.py:26-94; G11 fixture)."""
        u = [("layer", self.dense0), ("layer", self.dense1), ("layer", self.conv_in)]
        for lvl in self.down:
            for j, b in enumerate(lvl["block"]):
                u.append(("block", b))
                if lvl["attn"]:
                    u.append(("block", lvl["attn"][j]))
            if lvl["down"] is not None:
                u.append(("layer", lvl["down"]))
        u += [("block", self.mid1), ("block", self.mid_attn), ("block", self.mid2)]
        for i in reversed(range(self.nlev)):
            lvl = self.up[i]
            for j, b in enumerate(lvl["block"]):
                u.append(("block", b))
                if lvl["attn"]:
                    u.append(("block", lvl["attn"][j]))
            if lvl["up"] is not None:
                u.append(("layer", lvl["up"]))
        u.append(("layer", self.conv_out))
        return u

    def second_last_act_quantizer(self):
        # module order puts up.0 (lowest level) last inside `up`; its last block's last QuantModule
        last = self.up[0]["block"][-1]
        return last.layers()[-1].act_quantizer

    def __call__(self, x, t, context=None, stop_at=None):
        temb = ddpm_timestep_embedding(t, self.ch)
        temb = self.dense1(silu(self.dense0(temb)))
        hs = [self.conv_in(x)]
        for i, lvl in enumerate(self.down):
            for j in range(self.nres):
                h = lvl["block"][j](hs[-1], temb)
                if lvl["attn"]:
                    h = lvl["attn"][j](h)
                hs.append(h)
            if lvl["down"] is not None:
                hs.append(lvl["down"](F.pad(hs[-1], (0, 1, 0, 1))))
        h = self.mid2(self.mid_attn(self.mid1(hs[-1], temb)), temb)
        for i in reversed(range(self.nlev)):
            lvl = self.up[i]
            for j in range(self.nres + 1):
                split = h.size(1) if self.split_shortcut else 0
                h = lvl["block"][j](torch.cat([h, hs.pop()], 1), temb, split=split)
                if lvl["attn"]:
                    h = lvl["attn"][j](h)
            if lvl["up"] is not None:
                h = lvl["up"](F.interpolate(h, scale_factor=2.0, mode="nearest"))
        return self.conv_out(silu(self.norm_out(h)))


# --------------------------------------------------------------------------------------
# LDM blocks   quant_block.py:46-116 (QuantResBlock), :119-162 (QK/SMV), :168-192
# (QuantAttentionBlock), :204-285 (cross_attn_forward, QuantBasicTransformerBlock)
# --------------------------------------------------------------------------------------
class OResBlock(OBlock):
    def __init__(self, B, sdp, name, cin, cout, scale_shift=False, up=False, down=False):
        super().__init__(name)
        self.scale_shift, self.up, self.down = scale_shift, up, down
        self.in_norm = B.norm(sdp + ".in_layers.0", 1e-5)
        self.in_conv = B.layer(sdp + ".in_layers.2", name + ".in_layers.2", "conv2d", 1, 1)
        self.emb = B.layer(sdp + ".emb_layers.1", name + ".emb_layers.1", "linear")
        self.out_norm = B.norm(sdp + ".out_layers.0", 1e-5)
        self.out_conv = B.layer(sdp + ".out_layers.3", name + ".out_layers.3", "conv2d", 1, 1)
        self.skip = B.layer(sdp + ".skip_connection", name + ".skip_connection", "conv2d", 1, 0) if cin != cout else None

    def layers(self):
        return [self.in_conv, self.emb, self.out_conv] + ([self.skip] if self.skip else [])

    def _updown(self, x):
        if self.up:
            return F.interpolate(x, scale_factor=2, mode="nearest")
        if self.down:
            return F.avg_pool2d(x, 2, 2)
        return x

    def __call__(self, x, emb, split=0):
        # quant_block.py:72-84: split only reaches _forward while the skip has not split yet
        if not (split != 0 and self.skip is not None and self.skip.split == 0):
            split = 0
        if split != 0:
            self.split = split
        if self.up or self.down:
            h = self._updown(silu(self.in_norm(x)))
            x = self._updown(x)
            h = self.in_conv(h)
        else:
            h = self.in_conv(silu(self.in_norm(x)))
        e = self.emb(silu(emb))[:, :, None, None]
        if self.scale_shift:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = self.out_norm(h) * (1 + scale) + shift
            h = self.out_conv(silu(h))
        else:
            h = self.out_conv(silu(self.out_norm(h + e)))
        if self.skip is None:
            return x + h
        return (self.skip(x, split=self.split) if split != 0 else self.skip(x)) + h


class OCrossAttn:
    def __init__(self, B, sdp, name, heads):
        self.heads = heads
        self.to_q = B.layer(sdp + ".to_q", name + ".to_q", "linear")
        self.to_k = B.layer(sdp + ".to_k", name + ".to_k", "linear")
        self.to_v = B.layer(sdp + ".to_v", name + ".to_v", "linear")
        self.to_out = B.layer(sdp + ".to_out.0", name + ".to_out.0", "linear")
        self.scale = None
        self.use_act_quant = False
        self.act_quantizer_q = OQ(name=name + ".act_quantizer_q", **B.aq)
        self.act_quantizer_k = OQ(name=name + ".act_quantizer_k", **B.aq)
        self.act_quantizer_v = OQ(name=name + ".act_quantizer_v", **B.aq)
        self.act_quantizer_w = OQ(name=name + ".act_quantizer_w", **B.aq_w())

    def layers(self):
        return [self.to_q, self.to_k, self.to_v, self.to_out]

    def extra(self):
        return [self.act_quantizer_q, self.act_quantizer_k, self.act_quantizer_v, self.act_quantizer_w]

    def __call__(self, x, context=None):
        h = self.heads
        q = self.to_q(x)
        context = x if context is None else context
        k, v = self.to_k(context), self.to_v(context)

        def sp(t):
            b, n, hd = t.shape
            return t.reshape(b, n, h, hd // h).permute(0, 2, 1, 3).reshape(b * h, n, hd // h)

        q, k, v = sp(q), sp(k), sp(v)
        scale = (q.shape[-1]) ** -0.5
        if self.use_act_quant:
            sim = torch.einsum("bid,bjd->bij", self.act_quantizer_q(q), self.act_quantizer_k(k)) * scale
        else:
            sim = torch.einsum("bid,bjd->bij", q, k) * scale
        attn = sim.softmax(dim=-1)
        if self.use_act_quant:
            out = torch.einsum("bij,bjd->bid", self.act_quantizer_w(attn), self.act_quantizer_v(v))
        else:
            out = torch.einsum("bij,bjd->bid", attn, v)
        bh, n, d = out.shape
        out = out.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)
        return self.to_out(out)


class OTransformerBlock(OBlock):
    def __init__(self, B, sdp, name, heads):
        super().__init__(name)
        self.attn1 = OCrossAttn(B, sdp + ".attn1", name + ".attn1", heads)
        self.ff0 = B.layer(sdp + ".ff.net.0.proj", name + ".ff.net.0.proj", "linear")
        self.ff2 = B.layer(sdp + ".ff.net.2", name + ".ff.net.2", "linear")
        self.attn2 = OCrossAttn(B, sdp + ".attn2", name + ".attn2", heads)
        self.norm1, self.norm2, self.norm3 = B.ln(sdp + ".norm1"), B.ln(sdp + ".norm2"), B.ln(sdp + ".norm3")

    def layers(self):
        return self.attn1.layers() + [self.ff0, self.ff2] + self.attn2.layers()

    def ordered_quantizers(self):
        # named_modules order: attn1 {to_q,to_k,to_v,to_out, aq_q,k,v,w}, ff, attn2 {...}
        out = sum([l.quantizers() for l in self.attn1.layers()], []) + self.attn1.extra()
        out += self.ff0.quantizers() + self.ff2.quantizers()
        out += sum([l.quantizers() for l in self.attn2.layers()], []) + self.attn2.extra()
        return out

    def extra_quantizers(self):
        return self.attn1.extra() + self.attn2.extra()

    def set_quant_state(self, w, a):
        super().set_quant_state(w, a)
        self.attn1.use_act_quant = self.attn2.use_act_quant = a

    def __call__(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        h = self.ff0(self.norm3(x))
        a, gate = h.chunk(2, dim=-1)
        return self.ff2(a * F.gelu(gate)) + x


class OQKMatMul(OBlock):
    def __init__(self, B, name):
        super().__init__(name)
        self.scale = None
        self.act_quantizer_q = OQ(name=name + ".act_quantizer_q", **B.aq)
        self.act_quantizer_k = OQ(name=name + ".act_quantizer_k", **B.aq)

    def extra_quantizers(self):
        return [self.act_quantizer_q, self.act_quantizer_k]

    def __call__(self, q, k):
        if self.use_act_quant:
            return torch.einsum("bct,bcs->bts", self.act_quantizer_q(q * self.scale), self.act_quantizer_k(k * self.scale))
        return torch.einsum("bct,bcs->bts", q * self.scale, k * self.scale)


class OSMVMatMul(OBlock):
    def __init__(self, B, name):
        super().__init__(name)
        self.act_quantizer_v = OQ(name=name + ".act_quantizer_v", **B.aq)
        self.act_quantizer_w = OQ(name=name + ".act_quantizer_w", **B.aq_w(symmetric=False))

    def extra_quantizers(self):
        return [self.act_quantizer_v, self.act_quantizer_w]

    def __call__(self, weight, v):
        if self.use_act_quant:
            return torch.einsum("bts,bcs->bct", self.act_quantizer_w(weight), self.act_quantizer_v(v))
        return torch.einsum("bts,bcs->bct", weight, v)


class OSpatialTransformer:
    def __init__(self, B, sdp, name, heads):
        self.norm = B.norm(sdp + ".norm", 1e-6)
        self.proj_in = B.layer(sdp + ".proj_in", name + ".proj_in", "conv2d")
        self.block = OTransformerBlock(B, sdp + ".transformer_blocks.0", name + ".transformer_blocks.0", heads)
        self.proj_out = B.layer(sdp + ".proj_out", name + ".proj_out", "conv2d")

    def units(self):
        return [("layer", self.proj_in), ("block", self.block), ("layer", self.proj_out)]

    def __call__(self, x, context):
        b, c, h, w = x.shape
        x_in = x
        x = self.proj_in(self.norm(x))
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        x = self.block(x, context)
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
        return self.proj_out(x) + x_in


class _Recompute(torch.autograd.Function):
    """The reference's hand-written gradient checkpoint (ldm/modules/diffusionmodules/util.py:102-148): forward without a
    graph, backward = a SECOND forward with a graph -- in which every training-mode quantizer draws a new prob-mask
    (quant_layer.py:271-275) -- and the gradients of that second evaluation."""

    @staticmethod
    def forward(ctx, fn, n_in, *args):
        ctx.fn, ctx.n_in, ctx.args = fn, n_in, args
        with torch.no_grad():
            return fn(*args[:n_in])

    @staticmethod
    def backward(ctx, *gout):
        ins = [a.detach().requires_grad_(True) for a in ctx.args[:ctx.n_in]]
        with torch.enable_grad():
            out = ctx.fn(*ins)
        grads = torch.autograd.grad(out, ins + list(ctx.args[ctx.n_in:]), gout, allow_unused=True)
        return (None, None) + tuple(grads)


class OLegacyAttention:
    """AttentionBlock + QKVAttentionLegacy with QuantQKMatMul/QuantSMVMatMul swapped in
    (openaimodel.py:281-406; quant_block.py:119-162; get_specials quant_act=True)."""

    def __init__(self, B, sdp, name, heads):
        self.heads = heads
        self.name = name
        self.norm = B.norm(sdp + ".norm", 1e-5)
        self.qkv = B.layer(sdp + ".qkv", name + ".qkv", "conv1d")
        self.qk = OQKMatMul(B, name + ".attention.qkv_matmul")
        self.smv = OSMVMatMul(B, name + ".attention.smv_matmul")
        self.proj_out = B.layer(sdp + ".proj_out", name + ".proj_out", "conv1d")
        # Change_LDM_model_attnblock (recon_block_Qmodel.py:11-16) wraps the whole AttentionBlock into ONE
        # QuantAttentionBlock (quant_block.py:165-201) after scale initialisation: the walk then reconstructs it as a block
        # whose trainables are q, k, v, w of the two matmul wrappers, then qkv / proj_out (block_recon.py:66-79)
        self.merged = False
        self.split = 0

    def units(self):
        if self.merged:
            return [("block", self)]
        return [("layer", self.qkv), ("block", self.qk), ("block", self.smv), ("layer", self.proj_out)]

    def layers(self):
        return [self.qkv, self.proj_out]

    def extra_quantizers(self):
        return self.qk.extra_quantizers() + self.smv.extra_quantizers()

    def set_quant_state(self, w, a):
        for u in (self.qkv, self.qk, self.smv, self.proj_out):
            u.set_quant_state(w, a)

    def trainables(self):
        out = []
        for l in self.layers():
            for q in l.quantizers():
                out += [t for t in (getattr(q, "alpha", None), q.delta) if torch.is_tensor(t) and t.requires_grad]
        for q in self.extra_quantizers():
            if torch.is_tensor(q.delta) and q.delta.requires_grad:
                out.append(q.delta)
        return out

    def __call__(self, x, context=None):
        # QuantAttentionBlock.forward is checkpointed with the flag hard-wired to True (quant_block.py:180-182)
        if self.merged and torch.is_grad_enabled():
            return _Recompute.apply(self._forward, 1, x, *self.trainables())
        return self._forward(x)

    def _forward(self, x, context=None):
        b, c, hh, ww = x.shape
        xf = x.reshape(b, c, -1)
        qkv = self.qkv(self.norm(xf))
        bs, width, length = qkv.shape
        ch = width // (3 * self.heads)
        q, k, v = qkv.reshape(bs * self.heads, ch * 3, length).split(ch, dim=1)
        self.qk.scale = 1 / math.sqrt(math.sqrt(ch))
        weight = torch.softmax(self.qk(q, k).float(), dim=-1)
        a = self.smv(weight, v).reshape(bs, -1, length)
        return (xf + self.proj_out(a)).reshape(b, c, hh, ww)


def change_ldm_model_attnblock(net):
    """Change_LDM_model_attnblock (recon_block_Qmodel.py:11-16) on an OUNet."""
    n = 0
    for mods in net.input_blocks + [net.middle] + net.output_blocks:
        for m in mods:
            if isinstance(m, OLegacyAttention):
                m.merged = True
                n += 1
    return n


def ldm_timestep_embedding(t, dim, max_period=10000):
    """ldm/modules/diffusionmodules/util.py:151-171 (cos | sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class OUNet(_Net):
    """openaimodel.py:447-783 wrapped by QuantModel; transformer (ImageNet/SD) or legacy
    attention (Church/Bedroom) variants."""

    def __init__(self, sd, wq, aq, sm_abit=8, *, in_channels, out_channels, model_channels, attention_resolutions,
                 num_res_blocks, channel_mult, num_heads=-1, num_head_channels=-1, use_spatial_transformer=False,
                 context_dim=None, use_scale_shift_norm=False, resblock_updown=False, image_size=None,
                 transformer_depth=1, legacy=True):
        # legacy: with a spatial transformer and num_head_channels == -1 both settings give dim_head = ch // num_heads
        # (openaimodel.py:578-585)
        B = _Builder(sd, wq, aq, sm_abit)
        mc = self.mc = int(model_channels)
        self.split_shortcut = False
        ss, ru = bool(use_scale_shift_norm), bool(resblock_updown)
        attention_resolutions = [int(a) for a in np.atleast_1d(attention_resolutions)]
        channel_mult = [int(c) for c in np.atleast_1d(channel_mult)]
        nres = int(num_res_blocks)
        num_heads, num_head_channels = int(num_heads), int(num_head_channels)
        st = bool(use_spatial_transformer)
        self.te0 = B.layer("time_embed.0", "time_embed.0", "linear")
        self.te2 = B.layer("time_embed.2", "time_embed.2", "linear")

        def attn(sdp, ch):
            heads = num_heads if num_head_channels == -1 else ch // num_head_channels
            return OSpatialTransformer(B, sdp, sdp, heads) if st else OLegacyAttention(B, sdp, sdp, heads)

        self.input_blocks = [[B.layer("input_blocks.0.0", "input_blocks.0.0", "conv2d", 1, 1)]]
        chans = [mc]
        ch, ds, idx = mc, 1, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(nres):
                mods = [OResBlock(B, "input_blocks.%d.0" % idx, "input_blocks.%d.0" % idx, ch, mult * mc, ss)]
                ch = mult * mc
                if ds in attention_resolutions:
                    mods.append(attn("input_blocks.%d.1" % idx, ch))
                self.input_blocks.append(mods)
                chans.append(ch)
                idx += 1
            if level != len(channel_mult) - 1:
                if ru:
                    mods = [OResBlock(B, "input_blocks.%d.0" % idx, "input_blocks.%d.0" % idx, ch, ch, ss, down=True)]
                else:
                    mods = [B.layer("input_blocks.%d.0.op" % idx, "input_blocks.%d.0.op" % idx, "conv2d", 2, 1)]
                self.input_blocks.append(mods)
                chans.append(ch)
                idx += 1
                ds *= 2
        self.middle = [OResBlock(B, "middle_block.0", "middle_block.0", ch, ch, ss), attn("middle_block.1", ch),
                       OResBlock(B, "middle_block.2", "middle_block.2", ch, ch, ss)]
        self.output_blocks = []
        idx = 0
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(nres + 1):
                ich = chans.pop()
                mods = [OResBlock(B, "output_blocks.%d.0" % idx, "output_blocks.%d.0" % idx, ch + ich, mc * mult, ss)]
                ch = mc * mult
                if ds in attention_resolutions:
                    mods.append(attn("output_blocks.%d.1" % idx, ch))
                if level and i == nres:
                    k = len(mods)
                    if ru:
                        mods.append(OResBlock(B, "output_blocks.%d.%d" % (idx, k), "output_blocks.%d.%d" % (idx, k),
                                              ch, ch, ss, up=True))
                    else:
                        mods.append(("upsample", B.layer("output_blocks.%d.%d.conv" % (idx, k),
                                                          "output_blocks.%d.%d.conv" % (idx, k), "conv2d", 1, 1)))
                    ds //= 2
                self.output_blocks.append(mods)
                idx += 1
        self.out_norm = B.norm("out.0", 1e-5)
        self.out_conv = B.layer("out.2", "out.2", "conv2d", 1, 1)

    @staticmethod
    def _mod_units(m):
        if isinstance(m, OLayer):
            return [("layer", m)]
        if isinstance(m, tuple):
            return [("layer", m[1])]
        if isinstance(m, OBlock):
            return [("block", m)]
        return m.units()

    def units(self):
        u = [("layer", self.te0), ("layer", self.te2)]
        for mods in self.input_blocks + [self.middle] + self.output_blocks:
            for m in mods:
                u += self._mod_units(m)
        u.append(("layer", self.out_conv))
        return u

    def second_last_act_quantizer(self):
        qs = [q for q in self.all_quantizers() if q.leaf_param]
        return qs[-2]

    @staticmethod
    def _run(mods, h, emb, ctx, split=0):
        for m in mods:
            if isinstance(m, OResBlock):
                h = m(h, emb, split=split)
            elif isinstance(m, OLayer):
                h = m(h)
            elif isinstance(m, tuple):
                h = m[1](F.interpolate(h, scale_factor=2, mode="nearest"))
            else:
                h = m(h, ctx)
        return h

    def __call__(self, x, t, context=None):
        emb = self.te2(silu(self.te0(ldm_timestep_embedding(t, self.mc))))
        hs = []
        h = x
        for mods in self.input_blocks:
            h = self._run(mods, h, emb, context)
            hs.append(h)
        h = self._run(self.middle, h, emb, context)
        for mods in self.output_blocks:
            split = h.shape[1] if self.split_shortcut else 0
            h = torch.cat([h, hs.pop()], dim=1)
            h = self._run(mods, h, emb, context, split=split)
        return self.out_conv(silu(self.out_norm(h)))


# ======================================================================================
# Scale-init drivers                               qdiff/set_quantize_params.py:9-71
# ======================================================================================
def cfg_double(batch):
    """(x, t, index, cond, uncond, ...) -> the classifier-free-guidance batch [x, x], [t, t], [uncond, cond] that the
    calibration forward of the conditional samplers evaluates (ddim_control.py:221-228, plms.py:213-220)."""
    return [torch.cat([batch[0]] * 2), torch.cat([batch[1]] * 2), torch.cat([batch[4], batch[3]])]


def set_weight_quantize_params(net, cali, batch_size=32, transform=None):
    """transform=cfg_double with batch_size=2: set_weight_quantize_params_Stable / _Conditional
    (qdiff_control/set_quantize_params_Stable.py:107-145: one guided forward of the first two samples)."""
    net.set_quant_state(True, False)
    for l in net.all_layers():
        l.weight_quantizer.inited = False
    with torch.no_grad():
        b = [c[:batch_size] for c in cali]
        net(*(transform(b) if transform else b))
    for q in net.all_quantizers():
        if not q.leaf_param:
            q.inited = True


def set_act_quantize_params(net, cali, batch_size=256, all_attention=True, transform=None):
    """transform=cfg_double: the Stable / Conditional drivers (set_quantize_params_Stable.py:12-105)."""
    net.set_quant_state(True, True)
    for q in net.all_quantizers():
        if q.leaf_param:
            q.inited = False
    batch_size = min(batch_size, cali[0].size(0))
    with torch.no_grad():
        for i in range(int(cali[0].size(0) / batch_size)):
            b = [c[i * batch_size:(i + 1) * batch_size] for c in cali]
            net(*(transform(b) if transform else b))
    for q in net.all_quantizers():
        q.inited = True


# ======================================================================================
# a9 — save_inp_oup_data                              qdiff/data_utils.py:7-171
# ======================================================================================
class _Stop(Exception):
    pass


def _capture(net, unit, args):
    """Run `net` until `unit` has produced its output; return (inputs tuple, output)."""
    store = {}
    orig = unit.__class__.__call__

    def hooked(self, *a, **k):
        out = orig(self, *a, **k)
        if self is unit:
            store["inp"] = tuple(t.detach() for t in a if isinstance(t, torch.Tensor))
            store["out"] = out.detach()
            raise _Stop
        return out

    unit.__class__.__call__ = hooked
    try:
        with torch.no_grad():
            try:
                net(*args)
            except _Stop:
                pass
    finally:
        unit.__class__.__call__ = orig
    return store["inp"], store["out"]


def save_inp_oup_data(net, unit, cali, act_quant=True, batch_size=32):
    """asym=True, input_prob=True variant (block_recon.py:126): returns
    (resblock, (inp_q[, temb_q]), (inp_fp[, temb_fp]), out_fp)."""
    iq, ifp, ofp = [], [], []
    for i in range(int(cali[0].size(0) / batch_size)):
        args = [c[i * batch_size:(i + 1) * batch_size] for c in cali]
        net.set_quant_state(False, False)
        inp_fp, out_fp = _capture(net, unit, args)
        net.set_quant_state(True, act_quant)
        inp_q, _ = _capture(net, unit, args)
        iq.append(inp_q), ifp.append(inp_fp), ofp.append(out_fp)
    n_in = len(iq[0])
    cat = lambda lst, j: torch.cat([x[j] for x in lst])
    return n_in == 2, tuple(cat(iq, j) for j in range(n_in)), tuple(cat(ifp, j) for j in range(n_in)), torch.cat(ofp)


# ======================================================================================
# a6/a7 — reconstruction loops            block_recon.py:13-232, layer_recon.py:13-129
# ======================================================================================
def _prepare_unit(unit, kind, act_quant, recon_w, recon_a):
    if kind == "attn_layer":
        # attn_layer_recon.py:44-63: no AdaRound, no layer step sizes -- only q, k, v, w of the QuantAttnBlock
        a_para, ordered = [], list(unit.extra_quantizers())
        if act_quant:
            for q in ordered:
                q.delta = q.delta.detach().clone().requires_grad_(True)
                if recon_a:
                    a_para.append(q.delta)
                    q.is_training = True
        return [], [], a_para, ordered
    layers = [unit] if kind == "layer" else unit.layers()
    w_para, a_para = [], []
    for l in layers:
        if l.split == 0:
            l.weight_quantizer = OAdaRound(l.weight_quantizer, l.weight)
            if recon_w:
                l.weight_quantizer.soft_targets = True
                w_para.append(l.weight_quantizer.alpha)
        else:
            l.weight_quantizer = OAdaRound(l.weight_quantizer, l.weight[:, :l.split])
            l.weight_quantizer_0 = OAdaRound(l.weight_quantizer_0, l.weight[:, l.split:])
            if recon_w:
                l.weight_quantizer.soft_targets = l.weight_quantizer_0.soft_targets = True
                w_para += [l.weight_quantizer.alpha, l.weight_quantizer_0.alpha]
    aqs = []
    if kind == "block":
        aqs += unit.extra_quantizers()         # q,k,v,w come first inside the module loop for attention blocks
    ordered = []
    if kind == "block" and unit.extra_quantizers():
        # block_recon.py:46-108 iterates block.modules(): the block itself first (its q/k/v/w), then children
        ordered += unit.extra_quantizers()
    for l in layers:
        if l.act_quantizer.delta is not None:
            ordered.append(l.act_quantizer)
            if l.split:
                ordered.append(l.act_quantizer_0)
    if act_quant:
        for q in ordered:
            q.delta = q.delta.detach().clone().requires_grad_(True)
            if recon_a:
                a_para.append(q.delta)
                q.is_training = True
    return layers, w_para, a_para, ordered


def reconstruct_unit(net, unit, kind, cali, batch_size=32, iters=20000, act_quant=False, lr_a=4e-5, lr_w=1e-2,
                     p=2.0, input_prob=1.0, recon_w=False, recon_a=False, add_loss=0.0, cache_batch=32,
                     rand_fn=None, trace=None, caches=None):
    """block_reconstruction / layer_reconstruction with asym=True, opt_mode='mse', round loss 'none';
    kind 'attn_layer' = AttnBlock_layer_reconstruction (attn_layer_recon.py:13-133: block output loss only, the
    attention step sizes the only trainables, cur_inp = cur_sym when input_prob == 1 as in the block loop)."""
    unit.set_quant_state(True, act_quant)
    layers, w_para, a_para, aqs = _prepare_unit(unit, kind, act_quant, recon_w, recon_a)
    w_opt = OAdam(w_para, lr_w, iters) if w_para else None
    a_opt = OAdam(a_para, lr_a, iters) if a_para else None
    # `caches`: the (two, inp_q, inp_fp, out_fp) a test captured from the reference's own walk, instead of re-deriving them
    two, inp_q, inp_fp, out_fp = caches if caches is not None else save_inp_oup_data(net, unit, cali, act_quant, cache_batch)
    sz = out_fp.size(0)
    for it in range(iters):
        idx = random.sample(range(sz), batch_size)
        cur_out = out_fp[idx]
        cur_inp, cur_sym = inp_q[0][idx], inp_fp[0][idx]
        if two:
            temb_inp, temb_sym = inp_q[1][idx], inp_fp[1][idx]
        if input_prob < 1.0:
            u = rand_fn(cur_inp) if rand_fn is not None else torch.rand_like(cur_inp)
            cur_inp = torch.where(u < input_prob, cur_inp, cur_sym)
        elif kind in ("block", "attn_layer"):
            cur_inp = cur_sym          # block_recon.py:144-145, attn_layer_recon.py:101-102 (the layer loop keeps cur_inp, layer_recon.py:106-107)
        for o in (w_opt, a_opt):
            if o:
                o.zero_grad()
        args_q = (cur_inp, temb_inp) if two else (cur_inp,)
        args_fp = (cur_sym, temb_sym) if two else (cur_sym,)
        out_quant = unit(*args_q)
        m_loss = 0.0
        if kind == "block":
            for l in layers:
                l.record = True
            unit.set_quant_state(False, False)
            with torch.no_grad():
                unit(*args_fp)
            module_r = [l.hook_out for l in layers]
            unit.set_quant_state(True, act_quant)
            unit(*args_q)
            module_q = [l.hook_out for l in layers]
            for l in layers:
                l.record = False
            for j in range(len(module_r) - 1):
                m_loss = m_loss + lp_loss(module_q[j], module_r[j], p=2)
        loss = lp_loss(out_quant, cur_out, p=p) + add_loss * m_loss
        loss.backward()
        for o in (w_opt, a_opt):
            if o:
                o.step()
        if trace is not None:
            trace(it, w_para, a_para, float(loss.detach()))
    for l in layers:
        l.weight_quantizer.soft_targets = False
        l.act_quantizer.is_training = False
        if l.split:
            l.weight_quantizer_0.soft_targets = False
            l.act_quantizer_0.is_training = False
    for q in aqs:
        q.is_training = False


# ======================================================================================
# K9 — DDIM stepping          ddim/functions/denoising.py:4-59; ddim_control.py:198-254
# ======================================================================================
def compute_alpha(beta, t):
    beta = torch.cat([torch.zeros(1), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def ddim_step(xt, et, at, at_next, eta=0.0, noise=None):
    """denoising.py:50-56."""
    x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
    c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
    c2 = ((1 - at_next) - c1 ** 2).sqrt()
    nz = noise if noise is not None else torch.zeros_like(xt)
    return at_next.sqrt() * x0_t + c1 * nz + c2 * et, x0_t


def generalized_steps(x, seq, model, b, eta=0.0, noise=None):
    n = x.size(0)
    seq_next = [-1] + list(seq[:-1])
    xs, x0s = [x], []
    for i, j in zip(reversed(seq), reversed(seq_next)):
        t = torch.ones(n) * i
        nt = torch.ones(n) * j
        at, atn = compute_alpha(b, t.long()), compute_alpha(b, nt.long())
        et = model(xs[-1], t)
        xn, x0 = ddim_step(xs[-1], et, at, atn, eta, noise)
        xs.append(xn), x0s.append(x0)
    return xs, x0s


def make_ddim_timesteps(num_ddim, num_ddpm):
    """util.py:46-60, 'uniform'."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """util.py:63-74."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def ldm_linear_betas(n, linear_start, linear_end):
    """util.py:21-26."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2).numpy()


def p_sample_ddim(x, e_cond, e_uncond, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """ddim_control.py:206-254 (CFG combine + update); scalars or per-sample [B,1,1,1] tensors."""
    e_t = e_uncond + scale * (e_cond - e_uncond)
    pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    nz = sigma_t * noise if noise is not None else 0.0
    return a_prev.sqrt() * pred_x0 + dir_xt + nz, pred_x0


def plms_sample(apply_model, x_T, cond, uncond, scale, alphacums, n_steps):
    """ldm/models/diffusion/plms.py:136-279: full PLMS run (eta 0) with classifier-free guidance.
    Returns (final x, [x after each step], [pred_x0 of each step])."""
    ts = make_ddim_timesteps(n_steps, len(alphacums))
    _, al, alp = make_ddim_sampling_parameters(np.asarray(alphacums, dtype=np.float32), ts, 0.0)
    al, alp = torch.tensor(al, dtype=torch.float32), torch.tensor(alp, dtype=torch.float32)
    b = x_T.shape[0]

    def eps(x, t):
        tt = torch.full((b,), int(t), dtype=torch.long)
        if uncond is None or scale == 1.0:
            return apply_model(x, tt, cond)
        eu, ec = apply_model(torch.cat([x] * 2), torch.cat([tt] * 2), torch.cat([uncond, cond])).chunk(2)
        return eu + scale * (ec - eu)

    def step(x, e, index):
        a_t = torch.full((b, 1, 1, 1), float(al[index]))
        a_prev = torch.full((b, 1, 1, 1), float(alp[index]))
        s1 = torch.full((b, 1, 1, 1), float(np.sqrt(1.0 - al[index].numpy())))
        pred_x0 = (x - s1 * e) / a_t.sqrt()
        return a_prev.sqrt() * pred_x0 + (1.0 - a_prev).sqrt() * e, pred_x0

    time_range = np.flip(ts)
    total = len(time_range)
    img, old, xs, x0s = x_T, [], [], []
    for i, t in enumerate(time_range):
        index = total - i - 1
        e_t = eps(img, t)
        if len(old) == 0:
            x_prev, _ = step(img, e_t, index)
            e_next = eps(x_prev, time_range[min(i + 1, total - 1)])
            e_p = (e_t + e_next) / 2
        elif len(old) == 1:
            e_p = (3 * e_t - old[-1]) / 2
        elif len(old) == 2:
            e_p = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            e_p = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        img, p0 = step(img, e_p, index)
        old.append(e_t)
        if len(old) >= 4:
            old.pop(0)
        xs.append(img)
        x0s.append(p0)
    return img, xs, x0s


# ======================================================================================
# TDAC scoring                                           scripts/calibration.py:45-92
# ======================================================================================
def tdac_allocate(feature_map, lam, n_samples, dense_r, fixup_ge=False):
    T = len(feature_map)
    dense = torch.zeros(T, dtype=torch.int16)
    for i in range(T):
        for j in range(T):
            if i != j and torch.mean((feature_map[i] - feature_map[j]) ** 2) <= dense_r:
                dense[i] = dense[i] + 1
    dn = (dense - dense.min()) / (dense.max() - dense.min())
    cd = torch.zeros(T)
    for i in range(T):
        for j in range(T):
            if i != j:
                cd[i] = cd[i] + torch.sum(1 - F.cosine_similarity(feature_map[i], feature_map[j], dim=1, eps=1e-6))
    cdn = (cd - cd.min()) / (cd.max() - cd.min())
    w = dn + lam * cdn
    prob = w / torch.sum(w)
    t_num = (prob * n_samples).round().to(torch.int64)
    t_error = int(n_samples - torch.sum(t_num))
    _, order = torch.sort(t_num, descending=True)
    if t_error >= 0:
        t_num[order[:t_error]] += 1
    else:
        for i in reversed(range(T)):
            if t_error == 0:
                break
            if (t_num[i] >= 0) if fixup_ge else (t_num[i] > 0):
                t_num[i] -= 1
                t_error += 1
    return dense, cd, w, t_num
