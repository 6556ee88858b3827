"""Quantizers and the quantised layer — API of the reference's qdiff/quant_layer.py
(`UniformAffineQuantizer` :36-357, `QuantModule` :360-446, `lp_loss` :26-33, `round_ste` :19-23),
re-implemented on the HIP kernels of libedadm.so:

  * fake-quant forward / STE + LSQ backward ........ edadm_fake_quant_fwd / _bwd       (K1)
  * MSE clip-range search (the O(numel x 100) part) . edadm_mse_scores_tensor/_channel   (K3)
  * reconstruction loss ............................ edadm_lp_loss_fwd / _bwd          (K7)

Device tensors only: there is no CPU code path (ops raise on host tensors).
"""
import logging
import warnings
import os
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from edadm import ops

logger = logging.getLogger(__name__)
_mask_rng = random.Random(0x5EEDED)      # seeds of the in-kernel mask RNG; separate from `random`


def seed_mask_rng(seed):
    _mask_rng.seed(seed)


class StraightThrough(nn.Module):
    def __init__(self, channel_num: int = 1):
        super().__init__()

    def forward(self, input):
        return input


def round_ste(x):
    return (x.round() - x).detach() + x


# ----------------------------------------------------------------------------- K7
class _LpLoss2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt):
        # a sum over all elements: evaluated in pred's memory order (NHWC inside a convolutional unit, edadm/contract.py)
        pred, cl = ops.mem_view(pred)
        tgt = ops.mem_like(tgt, cl)
        ctx.save_for_backward(pred, tgt)
        ctx.cl = cl
        return ops.lp_loss_fwd(pred, tgt, C=pred.shape[3] if cl else None).reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, tgt = ctx.saved_tensors
        g = ops.lp_loss_bwd(pred, tgt, g.reshape(1).contiguous().float(), C=pred.shape[3] if ctx.cl else None)
        return ops.mem_restore(g, ctx.cl), None


def lp_loss(pred, tgt, p=2.0, reduction='none'):
    """sum over dim 1, mean over the rest (reduction 'none') or plain mean."""
    if p == 2.0 and reduction == 'none':
        return _LpLoss2.apply(pred, tgt.detach())
    e = (pred - tgt).abs().pow(p)
    return e.sum(1).mean() if reduction == 'none' else e.mean()


# ----------------------------------------------------------------------------- K1
class _FakeQuantTensor(torch.autograd.Function):
    """Per-tensor delta (activations): x and delta receive gradients.  Elementwise with one step size: runs on x's memory order
    (NHWC inside a convolutional unit); an injected uniform tensor (logical layout) is brought to the same order."""

    @staticmethod
    def forward(ctx, x, delta, zp, qmax, prob, seed, u):
        x, cl = ops.mem_view(x)
        if u is not None:
            u = ops.mem_like(u, cl)
        d1, z1 = delta.detach().reshape(1).contiguous(), zp.detach().reshape(1).float().contiguous()
        ctx.save_for_backward(x, d1, z1, u)
        ctx.meta = (qmax, prob, seed, delta.shape, cl)
        return ops.mem_restore(ops.fake_quant_fwd(x, d1, z1, qmax, u=u, prob=prob, seed=seed), cl)

    @staticmethod
    def backward(ctx, gy):
        x, d1, z1, u = ctx.saved_tensors
        qmax, prob, seed, dshape, cl = ctx.meta
        gx, gd = ops.fake_quant_bwd(ops.mem_like(gy, cl), x, d1, z1, qmax, u=u, prob=prob, seed=seed,
                                    need_gx=ctx.needs_input_grad[0])
        if gx is not None:
            gx = ops.mem_restore(gx, cl)
        return gx, gd.reshape(dshape) if ctx.needs_input_grad[1] else None, None, None, None, None, None


class _FakeQuantChannel(torch.autograd.Function):
    """Per-channel delta (weights): constants in the reconstruction loop; straight-through gradient."""

    @staticmethod
    def forward(ctx, x, delta, zp, qmax):
        x = x.contiguous()
        inner = x[0].numel()
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x, delta.detach(), zp.detach())
            ctx.qmax = qmax
        return ops.fake_quant_fwd(x, delta.reshape(-1).contiguous(), zp.reshape(-1).float().contiguous(), qmax,
                                  inner=inner)

    @staticmethod
    def backward(ctx, gy):
        # straight-through inside the clamp range, zero outside (the reference's torch.clamp, quant_layer.py:268-270);
        # off the hot loop: inside reconstruction every weight quantizer is an AdaRoundQuantizer, which detaches
        x, delta, zp = ctx.saved_tensors
        shp = [x.shape[0]] + [1] * (x.dim() - 1)
        code = torch.round(x / delta.reshape(shp)) + zp.reshape(shp)
        return gy * ((code >= 0) & (code <= ctx.qmax)).to(gy.dtype), None, None, None


def _first_argmin(scores, dim=0):
    """Index of the FIRST minimum along `dim` (the reference keeps the earliest candidate on ties:
    strict `<` in the sequential search, argmin on CPU)."""
    n = scores.shape[dim]
    shape = [1] * scores.dim()
    shape[dim] = n
    idx = torch.arange(n, device=scores.device).view(shape).expand_as(scores)
    mn = scores.min(dim=dim, keepdim=True)[0]
    return torch.where(scores == mn, idx, torch.full_like(idx, n)).min(dim=dim)[0]


class UniformAffineQuantizer(nn.Module):
    """Uniform affine quantizer with MSE (p=2.4) clip search, EMA range tracking for activations
    (`leaf_param`), and the dropout-like `prob` mixing during reconstruction."""

    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False,
                 scale_method: str = 'max', leaf_param: bool = False, always_zero: bool = False, prob: float = 1.0):
        super().__init__()
        self.sym = symmetric
        self.bitwidth_refactor(n_bits)
        self.delta = None
        self.zero_point = None
        self.inited = False
        self.leaf_param = leaf_param
        self.channel_wise = channel_wise
        self.scale_method = scale_method
        self.running_stat = False
        self.always_zero = always_zero
        if self.leaf_param:
            self.x_min, self.x_max = None, None
        self.running_min = None
        self.running_max = None
        self.one_side_dist = None
        self.num = 100
        self.eps = torch.tensor(1e-8, dtype=torch.float32)
        self.prob = prob
        self.is_training = False
        self.injected_uniform = None     # parity tests: uniforms that replace the in-kernel RNG

    def set_inited(self, inited: bool = True):
        self.inited = inited

    def bitwidth_refactor(self, refactored_bit: int):
        self.n_bits = refactored_bit
        self.n_levels = 2 ** self.n_bits

    # -- scale search: per-row (min, max) -> candidate grid -> |x - q(x)|^2.4 scores (one HIP pass
    #    over x, K3) -> first minimum -> EMA of the range (activations) -> (delta, zero_point).
    #    All float bookkeeping runs in edadm_mse_candidates / edadm_mse_select with IEEE division in
    #    the reference's operation order, so the scales come out bit-identical.
    def _aminmax(self, x):
        if self.channel_wise:
            mn, mx = torch.aminmax(torch.flatten(x.detach(), 1), dim=1)
            return mn.contiguous(), mx.contiguous()
        mm = ops.minmax(x.detach().reshape(-1).contiguous())
        return mm[0:1].contiguous(), mm[1:2].contiguous()

    def _scores(self, x, scale, zp):
        """scale/zp [nc][rows] -> scores [nc][rows] (rows = 1 for per-tensor quantizers)."""
        qmax = self.n_levels - 1
        nc = scale.shape[0]
        if not self.channel_wise:
            xs = x.detach().reshape(-1).contiguous()
            out = [ops.mse_scores_tensor(xs, scale[i:i + 128].reshape(-1).contiguous(),
                                         zp[i:i + 128].reshape(-1).contiguous(), qmax) for i in range(0, nc, 128)]
            return torch.cat(out).reshape(nc, 1)
        x2 = x.detach().reshape(x.shape[0], -1).contiguous()
        out = [ops.mse_scores_channel(x2, scale[i:i + 4096].contiguous(), zp[i:i + 4096].contiguous(), qmax)
               for i in range(0, nc, 4096)]
        return torch.cat(out)

    def get_x_min_x_max(self, x):
        raise NotImplementedError("folded into init_quantization_scale_1 (HIP K3 path)")

    def init_quantization_scale_1(self, x, channel_wise=False):
        if self.scale_method != "mse":
            raise NotImplementedError
        with torch.no_grad():
            if self.one_side_dist is None:
                self.one_side_dist = "pos" if x.min() >= 0.0 else "neg" if x.max() <= 0.0 else "no"
            mode = 1 if (self.one_side_dist != "no" or self.sym) else 2
            one = {"pos": 1, "neg": -1, "no": 0}[self.one_side_dist]
            clamp = mode == 2 and self.channel_wise
            xmin, xmax = self._aminmax(x)
            scale, zp = ops.mse_candidates(xmin, xmax, mode, one, self.n_bits, self.num, clamp)
            scores = self._scores(x, scale, zp)
            first = self.running_min is None
            if self.leaf_param and first:
                self.running_min = torch.empty_like(xmin)
                self.running_max = torch.empty_like(xmax)
            delta, zero_point = ops.mse_select(scores, xmin, xmax, mode, one, self.n_bits, self.num, clamp,
                                               self.running_min if self.leaf_param else None,
                                               self.running_max if self.leaf_param else None, first)
        if channel_wise:
            shp = [1] * x.dim()
            shp[0] = x.shape[0]
            return delta.reshape(shp), zero_point.reshape(shp)
        return delta.reshape(()), zero_point.reshape(())

    def init_quantization_scale_2(self, x, channel_wise=False):
        """scale_method='max' (quant_layer.py:278-330), the constructor default: the range rule on the (per-channel) extrema.
        The reference evaluates it in Python floats (`.item()`, i.e. double) channel by channel; here the extrema come from one
        device pass and the same double arithmetic runs vectorised on the host -- bit-identical step sizes.  Its symmetric
        branch is delta = absmax / n_levels with zero_point 0 (negative values clamp to code 0): kept as is."""
        import numpy as np
        if 'max' not in self.scale_method:
            raise NotImplementedError
        with torch.no_grad():
            xmin, xmax = self._aminmax(x)
            if self.leaf_param and not channel_wise:
                self.x_min, self.x_max = xmin.reshape(()), xmax.reshape(())
            mn, mx = xmin.double().cpu().numpy(), xmax.double().cpu().numpy()
        lo, hi = np.minimum(mn, 0.0), np.maximum(mx, 0.0)
        if 'scale' in self.scale_method:
            lo, hi = lo * (self.n_bits + 2) / 8, hi * (self.n_bits + 2) / 8
        if self.sym:
            delta = np.maximum(np.abs(lo), hi) / self.n_levels
        else:
            delta = (mx - mn) / (self.n_levels - 1)
        if (delta < 1e-8).any():
            warnings.warn('Quantization range close to zero')
            delta = np.where(delta < 1e-8, 1e-8, delta)
        zp = np.zeros_like(delta) if (self.sym or self.always_zero) else np.round(-lo / delta)     # round half to even, as Python's
        delta = torch.as_tensor(delta.astype(np.float32), device=x.device)
        zp = torch.as_tensor(zp.astype(np.float32) + 0.0, device=x.device)
        if channel_wise:
            shp = [1] * x.dim()
            shp[0] = x.shape[0]
            return delta.reshape(shp), zp.reshape(shp)
        return delta.reshape(()), zp.reshape(())

    def forward(self, x):
        if self.inited is False:
            if self.scale_method == 'mse':
                delta, self.zero_point = self.init_quantization_scale_1(x, self.channel_wise)
            elif self.scale_method == 'max':
                delta, self.zero_point = self.init_quantization_scale_2(x, self.channel_wise)
            else:
                raise NotImplementedError
            self.delta = nn.Parameter(delta) if self.leaf_param else delta
        qmax = self.n_levels - 1
        if self.channel_wise:
            return _FakeQuantChannel.apply(x, self.delta, self.zero_point, qmax)
        prob, seed, u = 1.0, 0, None
        if self.is_training and self.prob < 1.0:
            prob = self.prob
            if self.injected_uniform is not None:
                u = self.injected_uniform(x).contiguous()
            else:
                seed = _mask_rng.getrandbits(62)
        return _FakeQuantTensor.apply(x, self.delta, self.zero_point, qmax, prob, seed, u)

    def extra_repr(self):
        return 'bit={n_bits}, scale_method={scale_method}, symmetric={sym}, channel_wise={channel_wise},' \
               ' leaf_param={leaf_param}'.format(**self.__dict__)


def _contract(fwd_func, x, weight, bias, kw):
    """The contraction of quant_layer.py:434 on the fp32-MFMA kernels of edadm.contract (im2col + NT
    GEMM, autograd backward through the same kernels).  There is no other backend: a host tensor raises in
    edadm.ops (no CPU path)."""
    from edadm import contract
    if fwd_func is F.linear:
        return contract.linear(x, weight, bias)
    if fwd_func is F.conv1d and weight.shape[2] == 1:
        return contract.conv1d_k1(x, weight, bias)
    if (fwd_func is F.conv2d and kw.get("groups", 1) == 1 and tuple(kw.get("dilation", (1, 1))) == (1, 1)
            and kw["stride"][0] == kw["stride"][1] and kw["padding"][0] == kw["padding"][1]):
        return contract.conv2d(x, weight, bias, kw["stride"][0], kw["padding"][0])
    raise NotImplementedError("contraction %s %s %s" % (fwd_func.__name__, tuple(weight.shape), kw))


class QuantModule(nn.Module):
    """Conv2d / Conv1d / Linear with weight and activation quantizers, optional channel split of
    the input (the UNet skip concatenation gets one quantizer per half, quant_layer.py:406-427)."""

    def __init__(self, org_module, weight_quant_params: dict = {}, act_quant_params: dict = {},
                 disable_act_quant: bool = False, act_quant_mode: str = 'qdiff'):
        super().__init__()
        self.weight_quant_params = weight_quant_params
        self.act_quant_params = act_quant_params
        if isinstance(org_module, (nn.Conv2d, nn.Conv1d)):
            self.fwd_kwargs = dict(stride=org_module.stride, padding=org_module.padding,
                                   dilation=org_module.dilation, groups=org_module.groups)
            self.fwd_func = F.conv2d if isinstance(org_module, nn.Conv2d) else F.conv1d
        else:
            self.fwd_kwargs = dict()
            self.fwd_func = F.linear
        self.weight = org_module.weight
        self.org_weight = org_module.weight.data.clone()
        if org_module.bias is not None:
            self.bias = org_module.bias
            self.org_bias = org_module.bias.data.clone()
        else:
            self.bias = None
            self.org_bias = None
        self.use_weight_quant = False
        self.use_act_quant = False
        self.act_quant_mode = act_quant_mode
        self.disable_act_quant = disable_act_quant
        self.weight_quantizer = UniformAffineQuantizer(**self.weight_quant_params)
        if self.act_quant_mode == 'qdiff':
            self.act_quantizer = UniformAffineQuantizer(**self.act_quant_params)
        self.split = 0
        self.activation_function = StraightThrough()
        self.ignore_reconstruction = False
        self.extra_repr = org_module.extra_repr

    def _apply(self, fn, *a, **k):
        # org_weight / org_bias are plain tensors in the reference; keep them on the module's device
        super()._apply(fn, *a, **k)
        self.org_weight = fn(self.org_weight)
        if self.org_bias is not None:
            self.org_bias = fn(self.org_bias)
        return self

    def forward(self, input, split: int = 0):
        if split != 0 and self.split != 0:
            assert split == self.split
        elif split != 0:
            logger.info(f"split at {split}!")
            self.split = split
            self.set_split()
        if not self.disable_act_quant and self.use_act_quant:
            if self.split != 0:
                input = torch.cat([self.act_quantizer(input[:, :self.split]),
                                   self.act_quantizer_0(input[:, self.split:])], dim=1)
            else:
                input = self.act_quantizer(input)
        if self.use_weight_quant:
            if self.split != 0:
                weight = torch.cat([self.weight_quantizer(self.weight[:, :self.split, ...]),
                                    self.weight_quantizer_0(self.weight[:, self.split:, ...])], dim=1)
            else:
                weight = self.weight_quantizer(self.weight)
            bias = self.bias
        else:
            weight, bias = self.org_weight, self.org_bias
        # the fp32-grade contraction of the calibration graph (DESIGN.md "H1 contraction"); the sampling path never
        # comes here (edadm/engine.py)
        out = _contract(self.fwd_func, input, weight, bias, self.fwd_kwargs)
        return self.activation_function(out)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant

    def set_split(self):
        self.weight_quantizer_0 = UniformAffineQuantizer(**self.weight_quant_params)
        if self.act_quant_mode == 'qdiff':
            self.act_quantizer_0 = UniformAffineQuantizer(**self.act_quant_params)
