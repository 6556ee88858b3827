"""PLMSSampler — interface of the reference's ldm/models/diffusion/plms.py:12-279 including its `hooks` feature
capture (TDAC, :186-187), the `quant_unet` calibration forward (:249-250: one guided UNet evaluation, returned
as is) and the `ts_next` / `old_eps` intermediates the Stable-Diffusion calibration consumes.  The guidance
combine, the Adams-Bashforth combinations and the x_{t-1} update run in one HIP kernel (K9b, edadm_plms_step)."""
import numpy as np
import torch

from edadm import ops
from edadm.schedule import make_ddim_sampling_parameters, make_ddim_timesteps, ddim_coef_table


class PLMSSampler(object):
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
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
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
        if quant_unet:                                  # (x, t, index, cond, uncond, t_next): one guided evaluation
            x, t = cali_data[0], cali_data[1]
            c = cali_data[3] if len(cali_data) > 3 else conditioning
            uc = cali_data[4] if len(cali_data) > 4 else unconditional_conditioning
            return self._model_output(x, t, c, uc, unconditional_guidance_scale)
        samples, intermediates, feature_map = self.plms_sampling(
            conditioning, (batch_size, C, H, W), x_T=x_T, callback=callback, img_callback=img_callback,
            unconditional_guidance_scale=unconditional_guidance_scale,
            unconditional_conditioning=unconditional_conditioning, hooks=hooks)
        if len(feature_map) == 0:
            return samples, intermediates
        return samples, intermediates, feature_map

    def _eps(self, x, t, c, uc, scale):
        """(e_cond, e_uncond or None): the guidance combine itself happens inside the step kernel."""
        b = x.shape[0]
        if uc is None or scale == 1.:
            return self.model.apply_model(x, t, c).contiguous(), None
        out = self.model.apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uc, c])).contiguous()
        return out[b:], out[:b]

    def _model_output(self, x, t, c, uc, scale):
        e_c, e_u = self._eps(x, t, c, uc, scale)
        return e_c if e_u is None else e_u + scale * (e_c - e_u)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, hooks=None, **kwargs):
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        time_range = np.flip(self.ddim_timesteps)
        total = time_range.shape[0]
        intermediates = {'x_inter': [img], 'pred_x0': [img], 'ts': [], 'cond': [], 'uncond': [], 'old_eps': [], 'ts_next': []}
        feature_map, old_eps = [], []
        for i, step in enumerate(time_range):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, total - 1)]), device=device, dtype=torch.long)
            intermediates['old_eps'].append(list(old_eps))
            img, pred_x0, e_t = self.p_sample_plms(img, cond, ts, index=index,
                                                   unconditional_guidance_scale=unconditional_guidance_scale,
                                                   unconditional_conditioning=unconditional_conditioning,
                                                   old_eps=old_eps, t_next=ts_next)
            if hooks is not None:
                feature_map.append(hooks[0].feature[0])
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            intermediates['x_inter'].append(img)
            intermediates['pred_x0'].append(pred_x0)
            intermediates['ts'].append(ts)
            intermediates['ts_next'].append(ts_next)
            if index == 0 and cond is not None:
                intermediates['cond'].append(cond)
                if unconditional_conditioning is not None:
                    intermediates['uncond'].append(unconditional_conditioning)
        return img, intermediates, feature_map

    @torch.no_grad()
    def p_sample_plms(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, old_eps=None, t_next=None,
                      quant_unet=False):
        b = x.shape[0]
        scale, uc = unconditional_guidance_scale, unconditional_conditioning
        if quant_unet:
            return self._model_output(x, t, c, uc, scale)
        coef = self._coef[index:index + 1].expand(b, 5).contiguous()
        x = x.contiguous()
        e_c, e_u = self._eps(x, t, c, uc, scale)
        olds = list(reversed(old_eps))                       # newest first
        if len(olds) == 0:
            x_tmp, e_t = ops.plms_step(x, e_c, e_u, scale, [], 0, coef)
            e_c2, e_u2 = self._eps(x_tmp, t_next, c, uc, scale)
            x_prev, _, pred_x0 = ops.plms_step(x, e_c2, e_u2, scale, [e_t], -1, coef, want_x0=True)
        else:
            x_prev, e_t, pred_x0 = ops.plms_step(x, e_c, e_u, scale, olds[:3], min(len(olds), 3), coef, want_x0=True)
        return x_prev, pred_x0, e_t
