"""Scale initialisation for Stable Diffusion — qdiff_control/set_quantize_params_Stable.py:12-145 of the reference.

`args.plms` picks the PLMSSampler (the txt2img default, :58,122), otherwise the plain DDIMSampler; either way a calibration
batch goes through the sampler's single `quant_unet` forward.  cali_data = (x, t, index, cond, uncond, t_next) as the TDAC
generator for COCO prompts returns it (scripts/calibration.py:502-638); weights use the first 2 samples (:116).

The prompt conditioning: the reference encodes `args.list_prompts` with the CLIP text encoder (third-party, out of
scope).  With a `cond_stage_model` attached the same calls are made; without one the conditioning rows captured in
the calibration tuple stand in (the PLMS calibration forward reads them from the tuple anyway, plms.py:99-105)."""
import logging

import torch

from qdiff.quant_layer import QuantModule
from qdiff.set_quantize_params_LDM import all_act_quantizers

logger = logging.getLogger(__name__)


def _sampler(model, args):
    if getattr(args, "plms", False):
        from ldm.models.diffusion.plms import PLMSSampler
        return PLMSSampler(model)
    from ldm.models.diffusion.ddim import DDIMSampler
    return DDIMSampler(model)


def _conditioning(model, args, cali_batch, batch_size):
    if getattr(model, "cond_stage_model", None) is not None:
        uc = model.get_learned_conditioning(batch_size * [""]) if args.scale != 1.0 else None
        return model.get_learned_conditioning(args.list_prompts[:batch_size]), uc
    c = cali_batch[3] if len(cali_batch) > 3 else None
    uc = cali_batch[4] if len(cali_batch) > 4 and args.scale != 1.0 else None
    return c, uc


def _drive(model, args, cali_batch, sampler, batch_size):
    c, uc = _conditioning(model, args, cali_batch, batch_size)
    shape = [args.C, args.H // args.f, args.W // args.f]
    return sampler.sample(S=args.custom_steps, conditioning=c, batch_size=batch_size, shape=shape, verbose=False,
                          unconditional_guidance_scale=args.scale, unconditional_conditioning=uc, eta=args.ddim_eta,
                          x_T=None, quant_unet=True, cali_data=cali_batch)


def set_act_quantize_params_Stable(module, cali_data, args, batch_size: int = 2):
    logger.info("set_act_quantize_params")
    unet = module.model.diffusion_model
    if hasattr(unet, 'engine'):
        unet.engine = None          # a frozen executor was compiled from the old scales: freeze() again
    unet.set_quant_state(True, True)
    for q in all_act_quantizers(unet):
        q.set_inited(False)
    batch_size = min(batch_size, cali_data[0].size(0))
    sampler = _sampler(module, args)
    with torch.no_grad():
        for i in range(int(cali_data[0].size(0) / batch_size)):
            _drive(module, args, [c[i * batch_size:(i + 1) * batch_size].cuda() for c in cali_data], sampler, batch_size)
    for q in all_act_quantizers(unet):
        q.set_inited(True)


def set_weight_quantize_params_Stable(model, cali_data, args):
    logger.info("set_weight_quantize_params")
    unet = model.model.diffusion_model
    if hasattr(unet, 'engine'):
        unet.engine = None
    unet.set_quant_state(True, False)
    for m in unet.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(False)
    with torch.no_grad():
        _drive(model, args, [c[:2].cuda() for c in cali_data], _sampler(model, args), 2)
    for m in unet.modules():
        if isinstance(m, QuantModule):
            m.weight_quantizer.set_inited(True)
            if m.split != 0:
                m.weight_quantizer_0.set_inited(True)
