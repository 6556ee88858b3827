"""Minimal stand-in for the reference's LatentDiffusion wrapper (ldm/models/diffusion/ddpm.py:895-997,
1419-1445): schedules, `apply_model`, class-embedding conditioning — only what the samplers and the
calibration drivers touch.  The lightning training module, EMA scope and first stage are out of scope
(`decode_first_stage` raises unless a decoder is attached)."""
from contextlib import contextmanager

import numpy as np
import torch
import torch.nn as nn

from .schedule import make_beta_schedule


class DiffusionWrapper(nn.Module):
    def __init__(self, diffusion_model, conditioning_key="crossattn"):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_crossattn=None):
        if self.conditioning_key is None or c_crossattn is None:
            return self.diffusion_model(x, t)
        return self.diffusion_model(x, t, context=torch.cat(c_crossattn, 1))


class ClassEmbedder(nn.Module):
    """ldm/modules/encoders/modules.py:21-33."""

    def __init__(self, embed_dim, n_classes=1000, key='class_label'):
        super().__init__()
        self.key = key
        self.embedding = nn.Embedding(n_classes, embed_dim)

    def forward(self, batch, key=None):
        return self.embedding(batch[key or self.key][:, None])


class LatentDiffusionLite(nn.Module):
    def __init__(self, unet, timesteps=1000, linear_start=1e-4, linear_end=2e-2, conditioning_key=None,
                 cond_stage_model=None, cond_stage_key="class_label", first_stage_model=None,
                 parameterization="eps"):
        super().__init__()
        self.model = DiffusionWrapper(unet, conditioning_key)
        self.cond_stage_model, self.cond_stage_key = cond_stage_model, cond_stage_key
        self.first_stage_model = first_stage_model
        self.parameterization = parameterization
        self.num_timesteps = timesteps
        betas = make_beta_schedule("linear", timesteps, linear_start, linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(np.append(1.0, ac[:-1])))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - ac)))

    @property
    def device(self):
        return self.betas.device

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        if cond is None:
            return self.model(x_noisy, t)
        if isinstance(cond, dict):
            cond = cond.get("c_crossattn", cond)
        if not isinstance(cond, list):
            cond = [cond]
        return self.model(x_noisy, t, c_crossattn=cond)

    def get_learned_conditioning(self, c):
        if self.cond_stage_model is None:
            raise RuntimeError("no cond_stage_model attached (conditioning encoders are out of scope)")
        return self.cond_stage_model(c)

    def decode_first_stage(self, z):
        if self.first_stage_model is None:
            raise RuntimeError("no first-stage decoder attached (SURVEY.md 8f-3: next tier)")
        return self.first_stage_model.decode(z)

    @contextmanager
    def ema_scope(self, context=None):
        yield None
