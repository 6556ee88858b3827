"""Scale initialisation for class-conditional LDM (ImageNet) — qdiff_control/
set_quantize_params_Conditional.py:11-140 of the reference: calibration batches go through
DDIMSampler_control's single calibration forward with classifier-free guidance; weights use the first 2
samples (:111), activations batches of 32."""
import logging

import torch

from qdiff.quant_layer import QuantModule
from qdiff.set_quantize_params_LDM import all_act_quantizers

logger = logging.getLogger(__name__)


def _sampler(model):
    from ldm.models.diffusion.ddim_control import DDIMSampler_control
    return DDIMSampler_control(model)


def _shape(args, cali):
    return list(cali[0].shape[1:])


def set_act_quantize_params_Conditional(module, cali_data, args, batch_size: int = 32):
    logger.info("set_act_quantize_params")
    unet = module.model.diffusion_model
    if hasattr(unet, 'engine'):
        unet.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    unet.set_quant_state(True, True)
    for q in all_act_quantizers(unet):
        q.set_inited(False)
    batch_size = min(batch_size, cali_data[0].size(0))
    sampler = _sampler(module)
    with torch.no_grad():
        for i in range(int(cali_data[0].size(0) / batch_size)):
            sampler.sample(S=args.custom_steps, conditioning=None, batch_size=batch_size, shape=_shape(args, cali_data),
                           verbose=False, unconditional_guidance_scale=args.scale, eta=args.ddim_eta, quant_unet=True,
                           cali_data=[c[i * batch_size:(i + 1) * batch_size].cuda() for c in cali_data])
    for q in all_act_quantizers(unet):
        q.set_inited(True)


def set_weight_quantize_params_Conditional(model, cali_data, args):
    logger.info("set_weight_quantize_params")
    unet = model.model.diffusion_model
    if hasattr(unet, 'engine'):
        unet.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    unet.set_quant_state(True, False)
    for m in unet.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(False)
    with torch.no_grad():
        _sampler(model).sample(S=args.custom_steps, conditioning=None, batch_size=2, shape=_shape(args, cali_data),
                               verbose=False, unconditional_guidance_scale=args.scale, eta=args.ddim_eta,
                               quant_unet=True, cali_data=[c[:2].cuda() for c in cali_data])
    for m in unet.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(True)
            if m.split != 0:
                m.weight_quantizer_0.set_inited(True)
