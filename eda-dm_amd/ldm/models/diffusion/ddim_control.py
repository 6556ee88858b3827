"""DDIMSampler_control — the conditional (classifier-free guidance) sampler of the reference's
ldm/models/diffusion/ddim_control.py:12-291: same as DDIMSampler, and its calibration forward takes
cali_data = (x, t, index, cond, uncond) (:102-116)."""
from ldm.models.diffusion.ddim import DDIMSampler


class DDIMSampler_control(DDIMSampler):
    cfg_capable = True

    def _calibration_forward(self, cali_data, scale, conditioning=None):
        x, t, index, c, uc = cali_data[:5]
        return self.p_sample_ddim(x, c, t, index=index, unconditional_guidance_scale=scale,
                                  unconditional_conditioning=uc, quant_unet=True)
