"""Scale initialisation for unconditional LDM models (LSUN) — qdiff/set_quantize_params_LDM.py of
the reference: the calibration batch goes through the DDIM sampler's single "calibration forward"
(`sample(..., quant_unet=True, cali_data=[x, t, index])`, ddim.py:101-106,221-225)."""
import logging

import torch

from qdiff.quant_layer import QuantModule
from qdiff.quant_block import QuantAttnBlock, QuantSMVMatMul, QuantQKMatMul, QuantBasicTransformerBlock

logger = logging.getLogger(__name__)


def attention_quantizers(root):
    for m in root.modules():
        if isinstance(m, QuantAttnBlock):
            yield from (m.act_quantizer_k, m.act_quantizer_q, m.act_quantizer_v, m.act_quantizer_w)
        if isinstance(m, QuantSMVMatMul):
            yield from (m.act_quantizer_v, m.act_quantizer_w)
        if isinstance(m, QuantQKMatMul):
            yield from (m.act_quantizer_k, m.act_quantizer_q)
        if isinstance(m, QuantBasicTransformerBlock):
            for a in (m.attn1, m.attn2):
                yield from (a.act_quantizer_q, a.act_quantizer_k, a.act_quantizer_v, a.act_quantizer_w)


def all_act_quantizers(root):
    for m in root.modules():
        if isinstance(m, QuantModule):
            yield m.act_quantizer
            if m.split != 0:
                yield m.act_quantizer_0
    yield from attention_quantizers(root)


def _sampler(model, args):
    from ldm.models.diffusion.ddim import DDIMSampler
    return DDIMSampler(model)


def _drive(model, args, cali_batch, sampler):
    shape = [args.C, args.H // args.f, args.W // args.f] if hasattr(args, "C") else list(cali_batch[0].shape[1:])
    sampler.sample(S=args.custom_steps, batch_size=cali_batch[0].shape[0], shape=shape, verbose=False,
                   eta=getattr(args, "eta", getattr(args, "ddim_eta", 0.0)), quant_unet=True, cali_data=cali_batch)


def set_act_quantize_params_LDM(model, cali_data, args, batch_size: int = 32):
    logger.info("set_act_quantize_params")
    unet = model.model.diffusion_model
    if hasattr(unet, 'engine'):
        unet.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    unet.set_quant_state(True, True)
    for q in all_act_quantizers(unet):
        q.set_inited(False)
    batch_size = min(batch_size, cali_data[0].size(0))
    sampler = _sampler(model, args)
    with torch.no_grad():
        for i in range(int(cali_data[0].size(0) / batch_size)):
            _drive(model, args, [c[i * batch_size:(i + 1) * batch_size].cuda() for c in cali_data], sampler)
    for q in all_act_quantizers(unet):
        q.set_inited(True)


def set_weight_quantize_params_LDM(model, cali_data, args):
    logger.info("set_weight_quantize_params")
    unet = model.model.diffusion_model
    if hasattr(unet, 'engine'):
        unet.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    unet.set_quant_state(True, False)
    for m in unet.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(False)
    with torch.no_grad():
        _drive(model, args, [c[:8].cuda() for c in cali_data], _sampler(model, args))
    for m in unet.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(True)
            if m.split != 0:
                m.weight_quantizer_0.set_inited(True)
