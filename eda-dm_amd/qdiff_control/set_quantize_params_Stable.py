"""Scale initialisation for Stable Diffusion — qdiff_control/set_quantize_params_Stable.py:12-145 of the
reference.  The reference drives the PLMS sampler (`--plms`); this build ships the DDIM samplers only, so
the DDIM form (cali_data = (x, t, index, cond, uncond[, t_next])) is used and `args.plms` raises."""
from qdiff_control.set_quantize_params_Conditional import (set_act_quantize_params_Conditional,
                                                            set_weight_quantize_params_Conditional)


def _check(args):
    if getattr(args, "plms", False):
        raise NotImplementedError("PLMS sampler (ldm/models/diffusion/plms.py) is not built yet; use DDIM")


def set_act_quantize_params_Stable(module, cali_data, args, batch_size: int = 2):
    _check(args)
    return set_act_quantize_params_Conditional(module, cali_data[:5], args, batch_size=batch_size)


def set_weight_quantize_params_Stable(model, cali_data, args):
    _check(args)
    return set_weight_quantize_params_Conditional(model, cali_data[:5], args)
