"""Unit walk for conditional models — qdiff_control/recon_block_Qmodel.py:10-43 of the reference (plain
recursion: QuantModule -> layer reconstruction, BaseQuantBlock -> block reconstruction)."""
import logging

import torch.nn as nn

from qdiff.data_utils import clear_fp_trace

from qdiff.quant_layer import QuantModule
from qdiff.quant_block import BaseQuantBlock
from qdiff_control.block_recon import block_reconstruction
from qdiff_control.layer_recon import layer_reconstruction

logger = logging.getLogger(__name__)


class recon_block_Qmodel():
    def __init__(self, args, qnn, cali_data, kwargs):
        self.args, self.model, self.cali_data, self.kwargs = args, qnn, cali_data, kwargs
        self.down_name = None

    def recon_model(self, module: nn.Module):
        for name, m in module.named_children():
            if isinstance(m, QuantModule):
                if m.ignore_reconstruction:
                    logger.info('Ignore reconstruction of layer {}'.format(name))
                    continue
                logger.info('Reconstruction for layer {}'.format(name))
                layer_reconstruction(self.model, m, **self.kwargs)
            elif isinstance(m, BaseQuantBlock):
                if m.ignore_reconstruction:
                    logger.info('Ignore reconstruction of block {}'.format(name))
                    continue
                logger.info('Reconstruction for block {}'.format(name))
                block_reconstruction(self.model, m, **self.kwargs)
            else:
                self.recon_model(m)

    def recon(self):
        try:
            self.recon_model(self.model)
        finally:
            clear_fp_trace(self.model)          # look-ahead FP activations of units the walk never reached
        self.model.set_quant_state(weight_quant=True, act_quant=True)
        return self.model
