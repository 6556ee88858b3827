"""Walk the model in execution order and reconstruct unit by unit — the reference's
qdiff/recon_block_Qmodel.py:11-94 (traversal keyed on the child names 'down', '1', 'up')."""
import logging

import torch.nn as nn

from qdiff.data_utils import clear_fp_trace

from qdiff.quant_layer import QuantModule
from qdiff.quant_block import BaseQuantBlock, QuantAttentionBlock
from qdiff.block_recon import block_reconstruction
from qdiff.layer_recon import layer_reconstruction
from edadm.nets.ldm_unet import AttentionBlock

logger = logging.getLogger(__name__)


def Change_LDM_model_attnblock(module: nn.Module, act_quant_params: dict = {}):
    for name, child in module.named_children():
        if isinstance(child, AttentionBlock):
            setattr(module, name, QuantAttentionBlock(child, act_quant_params))
        else:
            Change_LDM_model_attnblock(child, act_quant_params)


class recon_block_Qmodel():
    def __init__(self, args, qnn, cali_data, kwargs):
        self.args, self.model, self.cali_data, self.kwargs = args, qnn, cali_data, kwargs
        self.down_name = None

    def _unit(self, name, m):
        if isinstance(m, QuantModule):
            if m.ignore_reconstruction:
                logger.info('Ignore reconstruction of layer {}'.format(name))
            else:
                logger.info('Reconstruction for layer {}'.format(name))
                layer_reconstruction(self.model, m, **self.kwargs)
            return True
        if isinstance(m, BaseQuantBlock):
            if m.ignore_reconstruction:
                logger.info('Ignore reconstruction of block {}'.format(name))
            else:
                logger.info('Reconstruction for block {}'.format(name))
                block_reconstruction(self.model, m, **self.kwargs)
            return True
        return False

    def _stage(self, stage, n_pairs, tail):
        """DDPM level with attention: block[0], attn[0], block[1], attn[1], ... then the resampler conv."""
        for j in range(n_pairs):
            block_reconstruction(self.model, stage.block[j], **self.kwargs)
            block_reconstruction(self.model, stage.attn[j], **self.kwargs)
        layer_reconstruction(self.model, tail, **self.kwargs)

    def recon_model(self, module: nn.Module):
        for name, m in module.named_children():
            if self.down_name is None and name == 'down':
                self.down_name = 'down'
            if self.down_name == 'down' and name == '1' and not isinstance(m, BaseQuantBlock):
                logger.info('reconstruction for down 1 modulelist')
                self._stage(m, 2, m.downsample.conv)
                self.down_name = 'over'
            elif self._unit(name, m):
                continue
            elif name == 'up':
                self.recon_up_model(m)
            else:
                self.recon_model(m)

    def recon_up_model(self, module: nn.Module):
        for name, m in reversed(list(module.named_children())):
            if name == '1':
                logger.info('reconstruction for up 1 modulelist')
                self._stage(m, 3, m.upsample.conv)
            elif self._unit(name, m):
                continue
            else:
                self.recon_model(m)

    def recon(self):
        try:
            self.recon_model(self.model)
        finally:
            clear_fp_trace(self.model)          # look-ahead FP activations of units the walk never reached
        self.model.set_quant_state(weight_quant=True, act_quant=True)
        return self.model
