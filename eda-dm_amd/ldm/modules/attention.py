"""Import-path compatibility for ldm/modules/attention.py."""
from edadm.nets.ldm_unet import (SpatialTransformer, BasicTransformerBlock, CrossAttention, FeedForward,  # noqa: F401
                                 GEGLU, CrossQKMatMul, CrossSMVMatMul, exists, default, Normalize, zero_module)
