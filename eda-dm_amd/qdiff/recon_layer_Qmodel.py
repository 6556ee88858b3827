"""Layer-wise reconstruction driver (`--layer_recon`) — qdiff/recon_layer_Qmodel.py:13-120 of the
reference: every QuantModule is reconstructed on its own; attention wrappers get only their
q/k/v/w activation step sizes tuned (attn_layer_recon.py)."""
import logging

import torch.nn as nn

from qdiff.quant_layer import QuantModule
from qdiff.quant_block import BaseQuantBlock, QuantAttnBlock
from qdiff.layer_recon import layer_reconstruction
from qdiff.attn_layer_recon import AttnBlock_layer_reconstruction

logger = logging.getLogger(__name__)


class recon_layer_Qmodel():
    def __init__(self, args, qnn, cali_data, kwargs):
        self.args, self.model, self.cali_data, self.kwargs = args, qnn, cali_data, kwargs

    def recon_model(self, module: nn.Module):
        for name, m in module.named_children():
            if isinstance(m, QuantModule):
                if m.ignore_reconstruction:
                    continue
                logger.info('Reconstruction for layer {}'.format(name))
                layer_reconstruction(self.model, m, **self.kwargs)
            elif isinstance(m, QuantAttnBlock):
                self.recon_model(m)                       # its q/k/v/proj_out layers first
                logger.info('Reconstruction for attention quantizers of {}'.format(name))
                AttnBlock_layer_reconstruction(self.model, m, **self.kwargs)
            elif name == 'up':
                for _, um in reversed(list(m.named_children())):
                    self.recon_model(um)
            else:
                self.recon_model(m)

    def recon(self):
        self.recon_model(self.model)
        self.model.set_quant_state(weight_quant=True, act_quant=True)
        return self.model
