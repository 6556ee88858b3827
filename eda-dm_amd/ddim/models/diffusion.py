"""Import-path compatibility: `from ddim.models.diffusion import Model` (reference
ddim/models/diffusion.py) resolves to the MI355X build's definition."""
from edadm.nets.ddpm_unet import (Model, ResnetBlock, AttnBlock, Upsample, Downsample, Normalize,  # noqa: F401
                                  nonlinearity, get_timestep_embedding)
