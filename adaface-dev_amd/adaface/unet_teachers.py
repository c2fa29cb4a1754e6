"""Frozen teacher U-Net for the Stage-1 distillation (reference ``adaface/unet_teachers.py:9-187,216-226``).

The reference wraps a diffusers ``UNet2DConditionModel`` (Arc2Face) under fp16 autocast; here the teacher is a
second instance of this package's ``UNetModel`` (same SD-1.5 arithmetic; diffusers and LDM key layouts are
equivalent by construction), so the teacher's forwards run on the same HIP kernels.
``forward`` keeps the reference signature and the multi-step self-denoising recipe:
q_sample -> U-Net (CFG if enabled) -> x0 -> sample an earlier t in [t*0.5^p, t*0.7^p], p = (n-1)^-0.3."""
import numpy as np
import torch
import torch.nn as nn


class UNetTeacher(nn.Module):
    def __init__(self, unet=None, cfg_scale_range=(1.3, 2), p_uses_cfg=0.0, name="unet_teacher"):
        super().__init__()
        self.name = name
        self.unet = unet
        self.p_uses_cfg = p_uses_cfg
        self.cfg_scale_range = cfg_scale_range
        self.cfg_scale = 1
        self.uses_cfg = False
        self.graphs = None            # graphs.GraphedSegment: set by the trainer (use_graphs) to replay the multi-step forward as one hipGraph
        if self.unet is not None:
            for p in self.unet.parameters():
                p.requires_grad_(False)

    def extract_pos_context(self, teacher_context, BS):
        """First half = positive context when (pos, neg) are stacked (unet_teachers.py:189-205)."""
        return teacher_context[:BS] if teacher_context.shape[0] == 2 * BS else teacher_context

    def _eps(self, x, t, ctx):
        return self.unet(x, t, ctx, extra_info=None)

    @torch.no_grad()
    def forward(self, ddpm_model, x_start, noise, t, teacher_context, negative_context=None, num_denoising_steps=1,
                force_uses_cfg=False, same_t_noise_across_instances=False, global_t_lb=0, global_t_ub=1000,
                presampled=None):
        """Returns (noise_preds, x_starts, noises, ts) like the reference.  `presampled`: optional list of
        (relative_ts, noise) per extra step -- lets tests feed identical randomness to the oracle."""
        assert num_denoising_steps <= 10
        if force_uses_cfg:
            self.uses_cfg = True
        elif self.p_uses_cfg > 0:
            self.uses_cfg = np.random.rand() < self.p_uses_cfg
        else:
            self.uses_cfg = False
        if self.uses_cfg:
            self.cfg_scale = np.random.uniform(*self.cfg_scale_range)
            if negative_context is not None:
                negative_context = negative_context[:1].repeat(x_start.shape[0], 1, 1)
        else:
            self.cfg_scale = 1
            if negative_context is None:
                teacher_context = self.extract_pos_context(teacher_context, x_start.shape[0])
        if same_t_noise_across_instances:
            t = t[0].repeat(x_start.shape[0])
            noise = noise[:1].repeat(x_start.shape[0], 1, 1, 1)
        same = bool(same_t_noise_across_instances)
        if self.graphs is not None and presampled is None and x_start.is_cuda:
            # fixed shapes, no host decision inside: one hipGraph per signature (graphs.py).  The random draws of the extra steps are
            # captured as graph-safe generator increments, so every replay draws fresh numbers; the guidance scale (a fresh host draw
            # per call) enters as a 0-dim device tensor, so a replay combines with THIS call's scale.
            guided = bool(self.uses_cfg and self.cfg_scale > 1)
            cfg_t = torch.full((), float(self.cfg_scale), device=x_start.device, dtype=torch.float32) if guided else None
            key = (tuple(x_start.shape), tuple(teacher_context.shape), None if negative_context is None else tuple(negative_context.shape),
                   int(num_denoising_steps), same, global_t_lb, global_t_ub, guided)
            return self.graphs.run(key, lambda xs, nz, tt, tc, nc, cf: self._multistep(ddpm_model, xs, nz, tt, tc, nc, num_denoising_steps, same,
                                                                                       global_t_lb, global_t_ub, None, cfg=cf),
                                   [x_start, noise, t, teacher_context, negative_context, cfg_t])
        return self._multistep(ddpm_model, x_start, noise, t, teacher_context, negative_context, num_denoising_steps, same, global_t_lb,
                               global_t_ub, presampled)

    def _multistep(self, ddpm_model, x_start, noise, t, teacher_context, negative_context, num_denoising_steps, same_t_noise_across_instances,
                   global_t_lb, global_t_ub, presampled, cfg=None):
        """The denoising loop of forward() (unet_teachers.py:115-185); same_t_noise was already applied to t / noise by the caller, the
        flag re-applies it to the draws of the extra steps."""
        x_starts, noises, ts, noise_preds = [x_start], [noise], [t], []
        cfg = self.cfg_scale if cfg is None else cfg                  # float, or a 0-dim device tensor under graph capture
        for i in range(num_denoising_steps):
            x_start, t, noise = x_starts[i], ts[i], noises[i]
            x_noisy = ddpm_model.q_sample(x_start, t, noise)
            guided = self.uses_cfg and self.cfg_scale > 1
            doubled = guided and negative_context is None
            # a separate negative context of the positive one's shape rides in the SAME U-Net call as a doubled batch (the reference makes
            # two calls, unet_teachers.py:150-158; per sample the arithmetic is identical, the launches are half as many and twice as wide)
            stacked = guided and negative_context is not None and negative_context.shape == teacher_context.shape
            if stacked:
                noise_pred = self._eps(x_noisy.repeat(2, 1, 1, 1), t.repeat(2), torch.cat([teacher_context, negative_context], dim=0))
            else:
                x2, t2 = (x_noisy.repeat(2, 1, 1, 1), t.repeat(2)) if doubled else (x_noisy, t)
                noise_pred = self._eps(x2, t2, teacher_context)
            if guided:
                if doubled or stacked:
                    pos, neg = torch.chunk(noise_pred, 2, dim=0)
                else:
                    pos, neg = noise_pred, self._eps(x_noisy, t, negative_context)
                noise_pred = pos * cfg - neg * (cfg - 1)
            noise_preds.append(noise_pred)
            pred_x0 = ddpm_model.predict_start_from_noise(x_noisy, t, noise_pred)
            x_starts.append(pred_x0)
            if i < num_denoising_steps - 1:
                if presampled is not None:
                    relative_ts, noise = presampled[i]
                else:
                    relative_ts = torch.rand_like(t.float())
                    noise = torch.randn_like(pred_x0)
                p = np.power(num_denoising_steps - 1, -0.3)
                t_lb = torch.clamp(t * np.power(0.5, p), min=global_t_lb)
                t_ub = torch.clamp(t * np.power(0.7, p), max=global_t_ub)
                earlier = ((t_ub - t_lb) * relative_ts + t_lb).long()
                if same_t_noise_across_instances:
                    earlier = earlier[0].repeat(x_start.shape[0])
                    noise = noise[:1].repeat(x_start.shape[0], 1, 1, 1)
                ts.append(earlier)
                noises.append(noise)
        return noise_preds, x_starts, noises, ts


class Arc2FaceTeacher(UNetTeacher):
    """Arc2Face teacher: CFG is pinned off (cfg_scale = 1, unet_teachers.py:216-226)."""

    def __init__(self, unet=None, **kwargs):
        super().__init__(unet=unet, name="arc2face", **kwargs)
        self.cfg_scale = 1
        self.p_uses_cfg = 0.0
