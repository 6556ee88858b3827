"""Import-path compatibility for ldm/modules/diffusionmodules/util.py (network helpers and the
DDIM schedule helpers)."""
from edadm.nets.ldm_unet import (checkpoint, conv_nd, linear, avg_pool_nd, zero_module, normalization,  # noqa: F401
                                 GroupNorm32, timestep_embedding)
from edadm.schedule import (make_beta_schedule, make_ddim_timesteps, make_ddim_sampling_parameters,  # noqa: F401
                            extract_into_tensor, noise_like)
