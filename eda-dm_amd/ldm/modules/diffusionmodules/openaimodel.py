"""Import-path compatibility for ldm/modules/diffusionmodules/openaimodel.py."""
from edadm.nets.ldm_unet import (UNetModel, ResBlock, AttentionBlock, QKMatMul, SMVMatMul,  # noqa: F401
                                 QKVAttentionLegacy, QKVAttention, TimestepBlock, TimestepEmbedSequential,
                                 Upsample, Downsample, checkpoint)
