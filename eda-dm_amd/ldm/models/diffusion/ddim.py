"""DDIMSampler — interface of the reference's ldm/models/diffusion/ddim.py:12-279 including the
`quant_unet / cali_data / hooks` extensions (single "calibration forward" with a per-sample timestep
index :101-106,221-225; feature capture for TDAC :166-168).  Stepping maths on the HIP K9 kernel."""
import numpy as np
import torch

from edadm import ops
from edadm.schedule import make_ddim_sampling_parameters, make_ddim_timesteps, ddim_coef_table


class DDIMSampler(object):
    cfg_capable = False

    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps
        self.register_buffer('betas', self.model.betas.float())
        self.register_buffer('alphas_cumprod', ac.float())
        self.register_buffer('alphas_cumprod_prev', self.model.alphas_cumprod_prev.float())
        sig, al, alp = make_ddim_sampling_parameters(ac.cpu().numpy(), self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, al, alp
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - al)
        self._coef = torch.tensor(ddim_coef_table(al, alp, sig), device=self.model.device)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, quant_unet=False, cali_data=None, hooks=None, **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=False)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if quant_unet:
            return self._calibration_forward(cali_data, unconditional_guidance_scale, conditioning)
        samples, intermediates, feature_map = self.ddim_sampling(
            conditioning, size, x_T=x_T, temperature=temperature,
            unconditional_guidance_scale=unconditional_guidance_scale,
            unconditional_conditioning=unconditional_conditioning, hooks=hooks, callback=callback,
            img_callback=img_callback)
        if len(feature_map) == 0:
            return samples, intermediates
        return samples, intermediates, feature_map

    def _calibration_forward(self, cali_data, scale, conditioning=None):
        x, t, index = cali_data[0], cali_data[1], cali_data[2]
        return self.p_sample_ddim(x, conditioning, t, index=index, quant_unet=True)          # ddim.py:100-105

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, temperature=1., unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, hooks=None, callback=None, img_callback=None, **kwargs):
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        timesteps = self.ddim_timesteps
        intermediates = {'x_inter': [img], 'pred_x0': [img], 'ts': [], 'cond': [], 'uncond': []}
        feature_map = []
        total = timesteps.shape[0]
        for i, step in enumerate(np.flip(timesteps)):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning)
            if hooks is not None:
                feature_map.append(hooks[0].feature[0])
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            intermediates['x_inter'].append(img)
            intermediates['pred_x0'].append(pred_x0)
            intermediates['ts'].append(ts)
            if index == 0 and cond is not None:
                intermediates['cond'].append(cond)
                if unconditional_conditioning is not None:
                    intermediates['uncond'].append(unconditional_conditioning)
        return img, intermediates, feature_map

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, quant_unet=False):
        b = x.shape[0]
        e_u = None
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            e_t = self.model.apply_model(x, t, c)
        else:
            out = self.model.apply_model(torch.cat([x] * 2), torch.cat([t] * 2),
                                         torch.cat([unconditional_conditioning, c]))
            e_u, e_t = out[:b], out[b:]
        if quant_unet:
            coef = self._coef[torch.as_tensor(index, device=self._coef.device).long().reshape(-1)].contiguous()
        else:
            coef = self._coef[index:index + 1].expand(b, 5).contiguous()
        noise = None
        if float(np.max(self.ddim_sigmas)) > 0:
            noise = torch.randn_like(x) * temperature
        return ops.ddim_step(x.contiguous(), e_t.contiguous(), None if e_u is None else e_u.contiguous(),
                             unconditional_guidance_scale, coef, noise=noise, want_x0=True)
