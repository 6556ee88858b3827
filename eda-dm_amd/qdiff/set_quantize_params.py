"""Scale-initialisation drivers — qdiff/set_quantize_params.py:9-71 of the reference: flip
`inited` off, push calibration batches through the model so every quantizer runs its MSE search
(HIP K3), flip it back on."""
import logging
from typing import Union

import torch

from qdiff.quant_layer import QuantModule
from qdiff.quant_block import BaseQuantBlock, QuantAttnBlock
from qdiff.quant_model import QuantModel

logger = logging.getLogger(__name__)
_ATTN_Q = ("act_quantizer_k", "act_quantizer_q", "act_quantizer_v", "act_quantizer_w")


def _act_quantizers(root):
    for m in root.modules():
        if isinstance(m, QuantModule):
            yield m.act_quantizer
            if m.split != 0:
                yield m.act_quantizer_0
        if isinstance(m, QuantAttnBlock):
            for n in _ATTN_Q:
                yield getattr(m, n)


def set_act_quantize_params(module: Union[QuantModel, QuantModule, BaseQuantBlock], cali_data, batch_size: int = 256):
    logger.info("set_act_quantize_params")
    if hasattr(module, 'engine'):
        module.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    module.set_quant_state(True, True)
    for q in _act_quantizers(module):
        q.set_inited(False)
    batch_size = min(batch_size, cali_data[0].size(0))
    with torch.no_grad():
        for i in range(int(cali_data[0].size(0) / batch_size)):
            module(*[c[i * batch_size:(i + 1) * batch_size].cuda() for c in cali_data])
    for q in _act_quantizers(module):
        q.set_inited(True)


def set_weight_quantize_params(model, cali_data):
    logger.info("set_weight_quantize_params")
    if hasattr(model, 'engine'):
        model.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    model.set_quant_state(True, False)
    for m in model.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(False)
    with torch.no_grad():
        model(*[c[:32].cuda() for c in cali_data])
    for m in model.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(True)
            if m.split != 0:
                m.weight_quantizer_0.set_inited(True)
