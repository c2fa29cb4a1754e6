"""Host-side mirror of the reference's ``ldm/models/diffusion/ddim.py`` (``DDIMSampler``):
uniform DDIM schedule, classifier-free guidance with the (conditional, unconditional) batch
order and linear guidance annealing (ddim.py:27-67, 133-221, 223-302).

Differences in execution only: the guidance combine and the DDIM update of one step are ONE
fused element-wise kernel (``af_cfg_ddim_step``) on the fp32 latents, the per-step scalar
tables stay on the host (no ``torch.full`` launches), and ``register_buffer`` does not force
tensors onto "cuda" by name (ddim.py:21-25 makes the reference sampler unusable elsewhere).
"""
import numpy as np
import torch

from .... import ops
from ...modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps


class DDIMSampler:
    def __init__(self, model, schedule="linear"):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        if ddim_eta != 0.0:
            raise NotImplementedError("eta > 0 (stochastic DDIM) is not used by the reference path")
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod.detach().float().cpu().numpy()
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps
        sig, a, ap = make_ddim_sampling_parameters(alphas_cumprod, self.ddim_timesteps, ddim_eta, verbose=verbose)
        self.register_buffer("ddim_sigmas", sig)
        self.register_buffer("ddim_alphas", a)
        self.register_buffer("ddim_alphas_prev", ap)
        self.register_buffer("ddim_sqrt_one_minus_alphas", np.sqrt(1.0 - a))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, img_callback=None, eta=0.0, mask=None, x0=None,
               verbose=True, x_T=None, log_every_t=100, guidance_scale=1.0, unconditional_conditioning=None, **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.ddim_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback,
                                  mask=mask, x0=x0, x_T=x_T, log_every_t=log_every_t, guidance_scale=guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @staticmethod
    def guide_scales(total_steps, guidance_scale):
        """The scale used at each of the steps (ddim.py:166-181, 216-219)."""
        if isinstance(guidance_scale, (list, tuple)):
            max_g, min_g = guidance_scale
        else:
            min_g = max_g = max(2.0, guidance_scale)
        max_anneal = total_steps - 1
        delta = (max_g - min_g) / max_anneal
        scales, g = [], max_g
        for i in range(total_steps):
            scales.append(g)
            g = g - delta if i <= max_anneal else 1
        return scales

    @torch.no_grad()
    def ddim_sampling(self, cond_context, shape, x_T=None, callback=None, timesteps=None, mask=None, x0=None,
                      img_callback=None, log_every_t=100, guidance_scale=1.0, unconditional_conditioning=None, **kwargs):
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        img = img.to(torch.float32).contiguous()
        if timesteps is None:
            timesteps = self.ddim_timesteps
        else:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        scales = self.guide_scales(total_steps, guidance_scale)
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img_orig = self.model.q_sample(x0, ts)
                img = (img_orig * mask + (1.0 - mask) * img).contiguous()
            img, pred_x0 = self.p_sample_ddim(img, cond_context, ts, index=index, guidance_scale=scales[i],
                                              unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, guidance_scale=1.0, unconditional_conditioning=None, **kwargs):
        """One DDIM step (ddim.py:223-302, eta = 0)."""
        has_uncond = not (unconditional_conditioning is None or guidance_scale == 1.0)
        if not has_uncond:
            e2 = self.model.apply_model(x, t, c)
        else:
            x_in = torch.cat([x] * 2)
            t_in = torch.cat([t] * 2)
            if isinstance(c, tuple):
                c_c, prompt_in_c, extra_info = c
                c_u, prompt_in_u, _ = unconditional_conditioning
                c2 = (torch.cat([c_c, c_u]), sum([prompt_in_c, prompt_in_u], []), extra_info)  # (cond, uncond) order
            else:
                c2 = torch.cat([c, unconditional_conditioning])
            e2 = self.model.apply_model(x_in, t_in, c2)
        a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
        x_prev, pred_x0 = ops.cfg_ddim_step(e2.to(torch.float32).contiguous(), x.to(torch.float32).contiguous(),
                                            guidance_scale, a_t, a_prev, has_uncond)
        return x_prev, pred_x0
