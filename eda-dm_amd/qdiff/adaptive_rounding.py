"""AdaRound quantizer — API of the reference's qdiff/adaptive_rounding.py:9-78 on the HIP kernels
edadm_adaround_init_alpha / _fwd / _bwd (K2).  Works directly on dim-1 slices of a weight (the
split skip convolutions, quant_layer.py:424-427) through leading-dimension arguments, no copies."""
import logging

import torch
from torch import nn

from edadm import ops
from qdiff.quant_layer import UniformAffineQuantizer, round_ste

logger = logging.getLogger(__name__)


class _SoftRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, alpha, delta, zp, qmax):
        out = torch.empty(w.shape, dtype=torch.float32, device=w.device)
        ops.adaround_fwd(w, alpha, out, delta, zp, qmax, True)
        ctx.save_for_backward(w, alpha, delta, zp)
        ctx.qmax = qmax
        return out

    @staticmethod
    def backward(ctx, gy):
        w, alpha, delta, zp = ctx.saved_tensors
        return None, ops.adaround_bwd(gy.contiguous(), w, alpha, delta, zp, ctx.qmax), None, None, None


class AdaRoundQuantizer(nn.Module):
    """Learned rounding: floor(w/delta) + rectified-sigmoid(alpha) while training (`soft_targets`),
    floor(w/delta) + (alpha >= 0) afterwards."""

    def __init__(self, uaq: UniformAffineQuantizer, weight_tensor: torch.Tensor, round_mode='learned_round_sigmoid'):
        super().__init__()
        self.n_bits = uaq.n_bits
        self.sym = uaq.sym
        self.delta = uaq.delta
        self.zero_point = uaq.zero_point
        self.n_levels = uaq.n_levels
        self.round_mode = round_mode
        self.alpha = None
        self.soft_targets = False
        self.gamma, self.zeta = -0.1, 1.1
        self.beta = 2 / 3
        self._d = self.delta.detach().reshape(-1).float().contiguous()
        self._z = self.zero_point.detach().reshape(-1).float().contiguous()
        self.init_alpha(x=weight_tensor)

    def forward(self, x):
        if self.round_mode != 'learned_hard_sigmoid':
            if self.round_mode in ('nearest', 'nearest_ste', 'stochastic'):
                raise NotImplementedError('only learned_hard_sigmoid is built (the reference constructs no other)')
            raise ValueError('Wrong rounding mode')
        xv = x.detach()
        if self.soft_targets:
            return _SoftRound.apply(xv, self.alpha, self._d, self._z, self.n_levels - 1)
        out = torch.empty(xv.shape, dtype=torch.float32, device=xv.device)
        ops.adaround_fwd(xv, self.alpha.detach(), out, self._d, self._z, self.n_levels - 1, False)
        return out

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def init_alpha(self, x: torch.Tensor):
        if self.round_mode == 'learned_hard_sigmoid':
            self.alpha = nn.Parameter(ops.adaround_init_alpha(x.detach(), self._d))
        else:
            raise NotImplementedError

    def extra_repr(self):
        return 'bit={n_bits}, symmetric={sym}, round_mode={round_mode}'.format(**self.__dict__)
