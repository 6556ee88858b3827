"""Layer-wise reconstruction driver (`--layer_recon`) — qdiff/recon_layer_Qmodel.py:13-120 of the reference.

The walk is the block walk's (same 'down' / '1' / 'up' name keys), but a block is taken apart: every QuantModule
of a QuantResnetBlock is reconstructed on its own in child order (:95-105), a QuantAttnBlock goes q, k, v, then its four
attention step sizes alone (AttnBlock_layer_reconstruction), then proj_out (:107-112).  Block types other than those
two are skipped by `recon_block` (:86-93) -- the reference only ships this mode for the DDPM (CIFAR) UNet.
Not mirrored: `recon_up_model` appends to `self.layer_loss`, an attribute the reference never creates (:71-72); the
branch is unreachable for the DDPM UNet (no QuantModule is a direct child of `up`)."""
import logging

import torch.nn as nn

from qdiff.data_utils import clear_fp_trace
from qdiff.quant_layer import QuantModule
from qdiff.quant_block import BaseQuantBlock, QuantAttnBlock, QuantResnetBlock
from qdiff.layer_recon import layer_reconstruction
from qdiff.attn_layer_recon import AttnBlock_layer_reconstruction

logger = logging.getLogger(__name__)


class recon_layer_Qmodel():
    def __init__(self, args, qnn, cali_data, kwargs):
        self.args, self.model, self.cali_data, self.kwargs = args, qnn, cali_data, kwargs
        self.down_name = None

    def _layer(self, name, m):
        if m.ignore_reconstruction is True:
            logger.info('Ignore reconstruction of layer {}'.format(name))
            return
        logger.info('Reconstruction for layer {}'.format(name))
        layer_reconstruction(self.model, m, **self.kwargs)

    def _stage(self, stage, n_pairs, tail):
        for j in range(n_pairs):
            self.recon_block(stage.block[j])
            self.recon_block(stage.attn[j])
        layer_reconstruction(self.model, tail, **self.kwargs)

    def recon_model(self, module: nn.Module):
        for name, m in module.named_children():
            if self.down_name is None and name == 'down':
                self.down_name = 'down'
            if self.down_name == 'down' and name == '1' and not isinstance(m, BaseQuantBlock):
                logger.info('reconstruction for down 1 modulelist')
                self._stage(m, 2, m.downsample.conv)
                self.down_name = 'over'
            elif isinstance(m, QuantModule):
                self._layer(name, m)
            elif isinstance(m, BaseQuantBlock):
                if m.ignore_reconstruction is True:
                    logger.info('Ignore reconstruction of block {}'.format(name))
                    continue
                logger.info('Reconstruction for block {}'.format(name))
                self.recon_block(m)
            elif name == 'up':
                self.recon_up_model(m)
            else:
                self.recon_model(m)

    def recon_up_model(self, module: nn.Module):
        for name, m in reversed(list(module.named_children())):
            if name == '1':
                logger.info('reconstruction for up 1 modulelist')
                self._stage(m, 3, m.upsample.conv)
            elif isinstance(m, QuantModule):
                self._layer(name, m)
            elif isinstance(m, BaseQuantBlock):
                if m.ignore_reconstruction is True:
                    continue
                self.recon_block(m)
            else:
                self.recon_model(m)

    def recon_block(self, block: nn.Module):
        if isinstance(block, QuantResnetBlock):
            self.recon_QuantResnetBlock_block(block)
        elif isinstance(block, QuantAttnBlock):
            self.recon_QuantAttnBlock_block(block)

    def recon_QuantResnetBlock_block(self, module: nn.Module):
        for name, m in module.named_children():
            if isinstance(m, QuantModule):
                self._layer(name, m)
            else:
                self.recon_QuantResnetBlock_block(m)

    def recon_QuantAttnBlock_block(self, module: nn.Module):
        layer_reconstruction(self.model, module.q, **self.kwargs)
        layer_reconstruction(self.model, module.k, **self.kwargs)
        layer_reconstruction(self.model, module.v, **self.kwargs)
        AttnBlock_layer_reconstruction(self.model, module, **self.kwargs)
        layer_reconstruction(self.model, module.proj_out, **self.kwargs)

    def recon(self):
        try:
            self.recon_model(self.model)
        finally:
            clear_fp_trace(self.model)
        self.model.set_quant_state(weight_quant=True, act_quant=True)
        return self.model
