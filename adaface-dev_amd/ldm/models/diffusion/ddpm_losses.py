"""The loss assemblies of the compositional-distillation and normal-recon iterations of ``LatentDiffusion`` (reference
``ldm/models/diffusion/ddpm.py``): ``calc_comp_feat_distill_loss`` :3190-3600, ``calc_comp_face_align_and_mb_suppress_losses`` :3602-3732,
``redenoise_subj_single`` :2093-2266, ``calc_arcface_align_loss`` :2511-2534, ``recon_multistep_denoise`` :1753-1917,
``calc_normal_recon_loss`` :2593-2883, ``calc_arcface_adv_grad`` :2536-2582 (the adversarial face edit), with the reference's signatures, gating and loss weights.  A mixin of
``ddpm.LatentDiffusion`` (kept in its own file: host orchestration only -- every tensor op below runs on device tensors, the U-Net /
VAE / ResNetFace passes they trigger are the HIP paths).

Pinned by fixtures written by the REFERENCE methods driven on a constructor-free shell (tests/golden/gen_golden.py::gen_comp_feat_distill,
gen_normal_recon) with the face detector, the ArcFace embedding step and the logging hooks replaced by the same stand-ins on both
sides: the sum, every monitor entry and the gradients w.r.t. the captured tensors.

Deviations, stated: (1) the RetinaFace detector network is an external package -- ``self.arcface.retinaface`` is a
``modules/arcface_wrapper.FaceCropper`` around a caller-supplied detector; (2) image logging (``cache_and_log_generations``) is a
no-op hook.

The ArcFace alignment / face-suppression terms carry their gradient like the reference's: ``decode_first_stage_with_grad`` is the VAE
decoder's autograd node (``diffusionmodules/model.VAEDecodeFn``), the crops / grey / resize are torch ops, the embedding is
``evaluation/arcface_resnet.FaceEncodeFn`` -- both frozen networks with input-gradient kernels only."""
import copy
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F

from ... import comp_losses as CL
from ...util import collate_dicts, split_dict


class RollingStats:
    """Windowed running sums / means of one or several values (reference ldm/util.py:198-237)."""

    def __init__(self, num_values=1, window_size=200, stat_type="mean"):
        self.window_size, self.num_values, self.stat_type = window_size, num_values, stat_type
        self.buffers = [deque(maxlen=window_size) for _ in range(num_values)]
        self.sums, self.means = [0] * num_values, [0] * num_values
        if num_values == 1:
            self.mean = 0

    def update(self, values):
        values = [values] if self.num_values == 1 else values
        if len(self.buffers[0]) == self.window_size:
            for i, b in enumerate(self.buffers):
                self.sums[i] -= b[0]
        for i, v in enumerate(values):
            self.buffers[i].append(v)
            self.sums[i] += v
            self.means[i] = self.sums[i] / len(self.buffers[i])
        if self.num_values == 1:
            self.sum, self.mean = self.sums[0], self.means[0]
            return self.mean if self.stat_type == "mean" else self.sum
        return self.means if self.stat_type == "mean" else self.sums


def box_mask(boxes, B, H, W, device, dtype=torch.float32):
    """[B, 1, H, W] with ones inside the (x1, y1, x2, y2) boxes.  The boxes are host data (FaceCropper): the mask is filled on the host
    and copied over without waiting for the device; the host copy rides along as ``.af_host`` for the decisions taken on it."""
    from ...modules.arcface_wrapper import to_device_async
    m = torch.zeros(B, 1, H, W, dtype=dtype)
    for i in range(len(boxes)):
        x1, y1, x2, y2 = (int(v) for v in boxes[i])
        m[i, :, y1:y2, x1:x2] = 1
    d = to_device_async(m, device)
    d.af_host = (m, d._version)
    return d


def host_of(mask):
    """The host copy of a ``box_mask`` (the mask itself when it has none, or when the device mask has been written in place since)."""
    memo = getattr(mask, "af_host", None)
    return memo[0] if memo is not None and memo[1] == mask._version else mask


def resolve_monitors(mon_loss_dict, extra=()):
    """The loss assemblies leave 0-dim DEVICE tensors in ``mon_loss_dict`` while they run; this turns them all into Python floats with
    ONE device read (the reference reads each with ``.item()`` where it is produced: ~20 waits per call).  ``extra``: more tensors to
    read in the same go -> their values."""
    for k, v in mon_loss_dict.items():
        if torch.is_tensor(v) and not v.is_cuda:
            mon_loss_dict[k] = v.detach().float().mean().item()                            # host tensors: nothing to wait for
    keys = [k for k, v in mon_loss_dict.items() if torch.is_tensor(v)]
    ts = [mon_loss_dict[k].detach().float().mean() for k in keys] + [t.detach().float().mean() for t in extra]
    if not ts:
        return []
    vals = torch.stack(ts).tolist()
    for k, v in zip(keys, vals):
        mon_loss_dict[k] = v
    return vals[len(keys):]


def chunk_list(lst, num_chunks):
    n = int(np.ceil(len(lst) / num_chunks))
    return [lst[i:i + n] for i in range(0, len(lst), n)]


COMP_MONITOR_NAMES = CL.PRESERVE_MONITORS + ("loss_sc_to_ssfg_sparse_attns_distill", "loss_sc_to_mc_sparse_attns_distill", "ssfg_sameloc_win_rate",
                                             "ssfg_flow_win_rate", "mc_flow_win_rate", "mc_sameloc_win_rate", "ssfg_avg_sparse_distill_weight",
                                             "mc_avg_sparse_distill_weight", "sc_bg_percent", "discarded_loss_ratio")


class CompReconLossesMixin:
    # reference constructor defaults (ddpm.py:84-127)
    recon_subj_mb_suppress_loss_weight = 0.2
    comp_sc_subj_mb_suppress_loss_weight = 0.2
    sc_fg_face_suppress_mask_shrink_ratio = 0.3
    comp_sc_fg_mask_percent_range = (0.0225, 0.36)
    recon_face_align_loss_thres = 0.8
    comp_sc_face_align_loss_thres = 0.7
    num_recon_denoising_steps = 2
    num_comp_distill_denoising_steps = 4
    redenoise_subj_comp_crop_mix_weights = (0.5, 0.25, 0.25)
    comp_ss_face_confidence_thres = 0.99
    comp_ss_face_lap_vars_tolerance = 0.3
    recon_bg_pixel_weight = 0.025
    arcface_align_loss_weight = 1e-2
    p_do_adv_attack_when_recon_on_images = 0
    recon_adv_mod_mag_range = (0.001, 0.003)
    adaface_adv_iters_count = 0
    adaface_adv_success_iters_count = 0
    # instance attributes set by LatentDiffusion.__init__ (modules must not be shadowed by class attributes): ``arcface`` -- a
    # modules/arcface_wrapper.ArcFaceWrapper, ``flow_model`` -- None (ddpm.py:652-662: only with use_face_flow_for_sc_matching_loss)
    comp_iters_count = 0
    comp_iters_bg_has_face_count = 0

    def _stat(self, name, **kw):
        """The reference's RollingStats members (ddpm.py:213-220), created on first use."""
        st = self.__dict__.get("_rolling_stats")
        if st is None:
            st = self.__dict__["_rolling_stats"] = {}
        if name not in st:
            st[name] = RollingStats(**kw)
        return st[name]

    comp_sc_face_detected_frac = property(lambda self: self._stat("comp_sc_face_detected_frac"))
    comp_mc_face_detected_frac = property(lambda self: self._stat("comp_mc_face_detected_frac"))
    comp_sc_face_suppressed_frac = property(lambda self: self._stat("comp_sc_face_suppressed_frac"))
    comp_sc_face_align_loss_kept_frac = property(lambda self: self._stat("comp_sc_face_align_loss_kept_frac"))
    comp_ss_redenoise_success_frac = property(lambda self: self._stat("comp_ss_redenoise_success_frac"))
    normal_recon_face_align_loss_kept_frac = property(lambda self: self._stat("normal_recon_face_align_loss_kept_frac"))
    normal_recon_face_images_on_image_stats = property(
        lambda self: self._stat("normal_recon_face_images_on_image_stats", num_values=2, window_size=600, stat_type="sum"))
    normal_recon_face_images_on_noise_stats = property(
        lambda self: self._stat("normal_recon_face_images_on_noise_stats", num_values=2, window_size=200, stat_type="sum"))

    def cache_and_log_generations(self, samples, img_colors, img_type, prompts=None, do_normalize=True):
        """Image logging hook of the reference (ddpm.py:3775-3800): nothing to do here."""

    def decode_first_stage_with_grad(self, z):
        """``decode_first_stage`` attached to the graph (ddpm.py:899-908): the decoder's input-gradient node when z requires grad."""
        if self.first_stage_model is None:
            raise RuntimeError("no first-stage decoder: call instantiate_first_stage() and load its weights")
        return self.first_stage_model.decode(z / self.scale_factor)

    # ------------------------------------------------------------------ face alignment (ddpm.py:2511-2534)
    def calc_arcface_align_loss(self, x_start, x_recon, fg_faces_grad_mask_ratios=(1, 0.3), ref_cache=None):
        """``ref_cache``: a dict the caller keeps while it aligns several x0 predictions to the SAME ``x_start`` (the compositional
        iteration does, once per denoising step, ddpm.py:3241-3247): the decoded reference and its face embedding -- deterministic
        functions of ``x_start`` -- are then computed once instead of once per call."""
        if ref_cache is not None and ref_cache.get("x_start") is x_start:
            x_start_pixels, ref = ref_cache["pixels"], ref_cache["ref"]
        else:
            x_start_pixels = self.decode_first_stage(x_start)
            ref = self.arcface.embed_reference(x_start_pixels)
            if ref_cache is not None:
                ref_cache.update(x_start=x_start, pixels=x_start_pixels, ref=ref)
        subj_recon_pixels = self.decode_first_stage_with_grad(x_recon)
        l_align, l_fg, l_bg, boxes, conf, found = self.arcface.calc_arcface_align_loss(x_start_pixels, subj_recon_pixels,
                                                                                       fg_faces_grad_mask_ratios=fg_faces_grad_mask_ratios, ref=ref)
        boxes = CL.map_bboxes_coords(boxes, x_start_pixels.shape[-1], x_start.shape[-1])
        return l_align.to(x_start.dtype), l_fg, l_bg.to(x_start.dtype), boxes, conf, found

    def calc_comp_face_align_and_mb_suppress_losses(self, mon_loss_dict, session_prefix, x_start0_ss, x_recons, ca_layers_activations_list,
                                                    all_subj_indices_1b, fg_faces_grad_mask_ratios, BLOCK_SIZE,
                                                    comp_sc_face_align_loss_kept_frac, comp_sc_face_align_loss_thres):
        """From the clearest (last) denoising step backwards: ArcFace alignment of the subject-comp x0 prediction to the input face
        (at most three steps are optimised; a step above the threshold only feeds the statistics), the SC face mask from the first
        step that shows a face, and from then on the subject-attention background suppression of every earlier step."""
        dev, dt = x_start0_ss.device, x_start0_ss.dtype
        zero = lambda: torch.zeros((), device=dev, dtype=dt)
        l_mb, l_align, l_align_stat, l_fg, l_bg = zero(), zero(), zero(), zero(), zero()
        n_align = n_stat = n_mb = n_fg = n_bg = 0
        sc_fg_mask = sc_boxes = None
        first_step = -1
        ref_cache = {}
        if self.arcface_align_loss_weight > 0:
            for step in range(len(x_recons) - 1, -1, -1):
                sc_recon = x_recons[step].chunk(4)[1]
                if n_align < 3:
                    la, lf, lb, boxes, _, _ = self.calc_arcface_align_loss(x_start0_ss, sc_recon, fg_faces_grad_mask_ratios, ref_cache=ref_cache)
                    # the branches below are on the VALUES of the three losses: ONE read of all three (one wait for the device) instead of
                    # one per comparison; without a face they are exact zeros and nothing needs reading
                    la_v = lf_v = lb_v = 0.0
                    if boxes is not None:
                        la_v, lf_v, lb_v = torch.stack([la.detach().float(), lf.detach().float(), lb.detach().float()]).tolist()
                    if la_v > 0:
                        keep = comp_sc_face_align_loss_thres <= 0 or la_v <= float(np.float32(comp_sc_face_align_loss_thres))
                        if keep:
                            l_align = l_align + la
                            n_align += 1
                        l_align_stat = l_align_stat + la
                        n_stat += 1
                        comp_sc_face_align_loss_kept_frac.update(1 if keep else 0)
                        if first_step == -1:
                            first_step = step
                        if sc_fg_mask is None:
                            sc_boxes = boxes
                            sc_fg_mask = box_mask(sc_boxes, sc_recon.shape[0], sc_recon.shape[2], sc_recon.shape[3], dev, sc_recon.dtype)
                        if lf_v > 0:
                            l_fg = l_fg + lf
                            n_fg += 1
                        if lb_v > 0:
                            l_bg = l_bg + lb
                            n_bg += 1
                if sc_fg_mask is not None:
                    sc_attn = {li: a.chunk(4)[1] for li, a in ca_layers_activations_list[step]["attn"].items()}
                    l_mb = l_mb + CL.calc_subj_masked_bg_suppress_loss(sc_attn, all_subj_indices_1b, BLOCK_SIZE, sc_fg_mask)
                    n_mb += 1
            for name, tot, n in (("arcface_align_comp_opt", l_align, n_align), ("arcface_align_comp", l_align_stat, n_stat),
                                 ("comp_sc_subj_mb_suppress", l_mb, n_mb), ("comp_fg_faces_suppress", l_fg, n_fg),
                                 ("comp_bg_faces_suppress", l_bg, n_bg)):
                if n > 0:
                    mon_loss_dict[f"{session_prefix}/{name}"] = (tot / n).mean().detach()
            self._host_pos = {"l_align": n_align > 0, "l_fg": n_fg > 0, "l_bg": n_bg > 0}     # sums of positive terms: their signs, known without a read
            l_align = l_align / n_align if n_align else l_align
            l_mb = l_mb / n_mb if n_mb else l_mb
            l_fg = l_fg / n_fg if n_fg else l_fg
            l_bg = l_bg / n_bg if n_bg else l_bg
        return l_align, l_fg, l_bg, l_mb, sc_fg_mask, sc_boxes, first_step

    # ------------------------------------------------------------------ second pass of the subject-single instance (ddpm.py:2093-2266)
    def redenoise_subj_single(self, x_starts, noises, ts, ca_layers_activations_list, ss_context, uncond_emb, all_subj_indices_1b,
                              ss_fg_face_crops_collate, ss_fg_face_bboxes, sc_fg_face_bboxes, use_attn_lora, use_ffn_lora,
                              sc_crop_mix_weights=(0.5, 0.25, 0.25), comp_ss_face_confidence_thres=0.99, lap_vars_tolerance=0.5):
        """The subject-single instance is denoised again from latents / noises whose face area is a mix of (the subject-comp face crop
        resized into the subject-single box, fresh noise, the original), so that its face matches the pose of the subject-comp one; the
        steps whose new face is detected confidently and is not much blurrier replace the first pass's SS activations and box IN PLACE
        in ``ca_layers_activations_list``.  Returns (per-step SS boxes, fraction of steps replaced)."""
        S = len(noises)
        latent_w = x_starts[0].shape[-1]
        xs_ss, xs_sc = [x.chunk(4)[0] for x in x_starts], [x.chunk(4)[1] for x in x_starts]
        nz_ss, nz_sc = [n.chunk(4)[0] for n in noises], [n.chunk(4)[1] for n in noises]
        xs_mix, nz_mix = copy.copy(xs_ss), copy.copy(nz_ss)
        w_sc, w_rand, w_ss = sc_crop_mix_weights
        for i in range(len(sc_fg_face_bboxes)):
            x1, y1, x2, y2 = sc_fg_face_bboxes[i]
            X1, Y1, X2, Y2 = ss_fg_face_bboxes[i]
            for step in range(S):
                for mix, src_sc, src_ss in ((nz_mix, nz_sc, nz_ss), (xs_mix, xs_sc, xs_ss)):      # noise first, then x_start: the reference's draw order
                    crop = F.interpolate(src_sc[step][i:i + 1, :, y1:y2, x1:x2], (Y2 - Y1, X2 - X1), mode="bilinear", align_corners=False)
                    mix[step][i, :, Y1:Y2, X1:X2] = crop * w_sc + torch.randn_like(crop) * w_rand + src_ss[step][i, :, Y1:Y2, X1:X2] * w_ss
        ts_ss = [t.chunk(4)[0] for t in ts]
        _, _, x_recons_ss, _, _, acts_ss = self.comp_distill_multistep_denoise(
            xs_mix, nz_mix, ts_ss, ss_context, uncond_emb=uncond_emb, all_subj_indices_1b=all_subj_indices_1b, normalize_cross_attn=False,
            mix_sc_mc_attn=False, cfg_scale=2.5, num_denoising_steps=self.num_comp_distill_denoising_steps, old_x_starts_mix_ratio=0.3,
            use_attn_lora=use_attn_lora, use_ffn_lora=use_ffn_lora, ffn_lora_adapter_name="comp_distill", BLKS=1, batch_part_has_grad="none")
        pixels = self.decode_first_stage(torch.cat(x_recons_ss, dim=0))
        crops2, _, boxes2, conf2, found2 = self.arcface.retinaface.crop_faces(pixels, out_size=(128, 128), T=20)
        found_steps = found2.chunk(S, dim=0)
        boxes_list, replaced = [ss_fg_face_bboxes] * S, 0
        if (1 - found_steps[-1]).sum() == 0:
            lap1 = [v.mean() for v in CL.var_of_laplacian(ss_fg_face_crops_collate).chunk(S, dim=0)]
            lap2 = [v.mean() for v in CL.var_of_laplacian(crops2).chunk(S, dim=0)]
            conf_steps = [c.mean() for c in conf2.chunk(S, dim=0)]
            boxes2_steps = chunk_list(CL.map_bboxes_coords(boxes2, pixels.shape[-1], latent_w), S)
            sharp = (torch.stack(lap2) >= torch.stack(lap1) * lap_vars_tolerance).tolist()          # one device read for all steps
            for step in range(S):
                good = bool(conf_steps[step] >= comp_ss_face_confidence_thres) and sharp[step]         # the confidences are host data
                if good:
                    replaced += 1
                    _, ca_sc, ca_sr, ca_mc = split_dict(ca_layers_activations_list[step], 4)
                    ca_layers_activations_list[step] = collate_dicts([acts_ss[step], ca_sc, ca_sr, ca_mc])
                    boxes_list[step] = boxes2_steps[step]
        return boxes_list, replaced / S

    # ------------------------------------------------------------------ the Stage-2 loss (ddpm.py:3190-3600)
    def calc_comp_feat_distill_loss(self, mon_loss_dict, session_prefix, x_start0_ss, x_starts, x_recons, x_recons_pixel_allsteps, noise_preds,
                                    noises, ts, ca_layers_activations_list, all_subj_indices_1b, ss_context, uncond_emb, prompt_emb_mask_4b,
                                    prompt_pad_mask_4b, BLOCK_SIZE, sc_fg_face_suppress_mask_shrink_ratio, use_attn_lora, use_ffn_lora):
        dev, dt = x_start0_ss.device, x_start0_ss.dtype
        latent_shape, S = x_start0_ss.shape, len(x_recons)
        P = session_prefix
        zero = lambda: torch.zeros((), device=dev, dtype=dt)
        loss = zero()
        l_fg_suppress, l_align = zero(), zero()
        sc_fg_mask = mc_fg_mask = ss_boxes = sc_boxes = ss_crops_collate = None
        all_ss_contain_faces, first_step = False, -1
        pos = {"l_align": False, "l_fg": False, "l_bg": False}       # signs of the face losses, known on the host
        rep_dist_fg_bounds = (0.1, 0.20, 0.25)

        if self.arcface_align_loss_weight > 0:
            # faces of the subject-single instances of every step (the last step decides), then of the class-comp instance
            # (x_recons_pixel_allsteps[i] holds the decoded blocks of step i, subject-single first: all four, or that block alone)
            ss_pixels = torch.cat([x_recons_pixel_allsteps[i][:BLOCK_SIZE] for i in range(len(x_recons_pixel_allsteps))], dim=0)
            ss_crops_collate, _, boxes_c, conf_c, found_c = self.arcface.retinaface.crop_faces(ss_pixels, out_size=(128, 128), T=20)
            ss_boxes, ss_conf, ss_found = boxes_c.chunk(S)[-1], conf_c.chunk(S)[-1], found_c.chunk(S)[-1]
            if (1 - ss_found).sum() == 0 and ss_conf.min() >= self.comp_ss_face_confidence_thres:
                all_ss_contain_faces = True
                ss_boxes = CL.map_bboxes_coords(ss_boxes, ss_pixels.shape[-1], latent_shape[-1])
                l_align, l_fg_suppress, l_bg_suppress, l_mb, sc_fg_mask, sc_boxes, first_step = \
                    self.calc_comp_face_align_and_mb_suppress_losses(
                        mon_loss_dict, P, x_start0_ss, x_recons, ca_layers_activations_list, all_subj_indices_1b,
                        fg_faces_grad_mask_ratios=(0.9, sc_fg_face_suppress_mask_shrink_ratio), BLOCK_SIZE=BLOCK_SIZE,
                        comp_sc_face_align_loss_kept_frac=self.comp_sc_face_align_loss_kept_frac,
                        comp_sc_face_align_loss_thres=self.comp_sc_face_align_loss_thres)
                pos = self._host_pos
                loss = loss + l_bg_suppress * 400 * self.arcface_align_loss_weight + l_mb * self.comp_sc_subj_mb_suppress_loss_weight
                if pos["l_bg"]:
                    self.comp_iters_bg_has_face_count += 1
                    mon_loss_dict[f"{P}/comp_iters_bg_has_face_frac"] = self.comp_iters_bg_has_face_count / self.comp_iters_count
            mc_pixels = self.decode_first_stage(x_recons[-1].chunk(4)[3])
            _, _, mc_boxes, _, mc_found = self.arcface.retinaface.crop_faces(mc_pixels, out_size=(128, 128), T=20)
            if (1 - mc_found).sum() == 0:
                mc_boxes = CL.map_bboxes_coords(mc_boxes, mc_pixels.shape[-1], latent_shape[-1])
                mc_fg_mask = box_mask(mc_boxes, BLOCK_SIZE, latent_shape[-2], latent_shape[-1], dev)

        for name in COMP_MONITOR_NAMES:
            mon_loss_dict[f"{P}/{name}"] = 0
        sc_pct = mc_pct = 0
        if sc_fg_mask is not None:
            sc_pct = host_of(sc_fg_mask).float().mean().item()
            mon_loss_dict[f"{P}/sc_fg_mask_percent"] = sc_pct
        if mc_fg_mask is not None:
            mc_pct = host_of(mc_fg_mask).float().mean().item()
            mon_loss_dict[f"{P}/mc_fg_mask_percent"] = mc_pct
        mon_loss_dict[f"{P}/comp_mc_face_detected_frac"] = self.comp_mc_face_detected_frac.update(1 if mc_fg_mask is not None else 0)
        mon_loss_dict[f"{P}/comp_sc_face_align_loss_kept_frac"] = self.comp_sc_face_align_loss_kept_frac.mean

        lo, hi = self.comp_sc_fg_mask_percent_range
        if sc_pct == 0:
            kind = "sc-noface"
        elif mc_pct == 0 and sc_pct >= 0.16 * hi:
            kind = "mc-no-sc-large"
        elif mc_pct > 0 and ((host_of(sc_fg_mask) * host_of(mc_fg_mask)).sum() / host_of(sc_fg_mask).sum()) < 0.16:
            kind = "little-no-overlap"
        elif sc_pct <= lo:
            kind = "too-small"
        elif sc_pct >= hi or (mc_pct > 0 and sc_pct >= 6.25 * mc_pct):
            kind = "too-large"
        else:
            kind = "good"
        self.sc_face_proportion_type = kind

        if pos["l_align"]:                                         # l_align > 0, known from the values read step by step
            frac = self.comp_sc_face_detected_frac.update(1)
            scale = (3 if kind in ("too-small", "good") else 1.5) * min(4, 1 / (frac ** 2 + 0.01))
            l_align_scaled = l_align * scale
            loss = loss + l_align_scaled * self.arcface_align_loss_weight
        else:
            frac = self.comp_sc_face_detected_frac.update(0)
            l_align_scaled = zero()
        mon_loss_dict[f"{P}/comp_sc_face_detected_frac"] = frac

        if kind != "sc-noface":
            ss_boxes_list, repl_frac = self.redenoise_subj_single(
                x_starts, noises, ts, ca_layers_activations_list, ss_context, uncond_emb, all_subj_indices_1b, ss_crops_collate, ss_boxes, sc_boxes,
                use_attn_lora=use_attn_lora, use_ffn_lora=use_ffn_lora, sc_crop_mix_weights=self.redenoise_subj_comp_crop_mix_weights,
                comp_ss_face_confidence_thres=self.comp_ss_face_confidence_thres, lap_vars_tolerance=self.comp_ss_face_lap_vars_tolerance)
            self.comp_ss_redenoise_success_frac.update(repl_frac)
        else:
            ss_boxes_list = [ss_boxes] * len(ca_layers_activations_list)
        mon_loss_dict[f"{P}/comp_ss_redenoise_success_frac"] = self.comp_ss_redenoise_success_frac.mean

        suppress = kind in ("mc-no-sc-large", "little-no-overlap", "too-large")
        if suppress:
            scale = {"mc-no-sc-large": 5, "little-no-overlap": 10, "too-large": 10}[kind]
            if pos["l_align"] and pos["l_fg"]:                     # l_align_scaled > 0 and l_fg_suppress > 0
                ratio = l_align_scaled.detach() / l_fg_suppress.detach()
                mon_loss_dict[f"{P}/align_suppress_loss_ratio"] = ratio
                scale = torch.clamp(ratio * 0.1, scale / 2, scale)              # CL.clamp's value, without reading ratio
            loss = loss + l_fg_suppress * scale * self.arcface_align_loss_weight
        mon_loss_dict[f"{P}/comp_sc_face_suppressed_frac"] = self.comp_sc_face_suppressed_frac.update(1 if suppress else 0)
        bg_match_shrink = sc_fg_face_suppress_mask_shrink_ratio if suppress else 1

        reps, preserve, cross_t, pred_l2s = [], [], [], []
        rep_steps = [CL.calc_sc_rep_attn_distill_loss(acts, all_subj_indices_1b, prompt_emb_mask_4b, prompt_pad_mask_4b, sc_pct, FG_THRES=rep_dist_fg_bounds[0])
                     for acts in ca_layers_activations_list]
        live = [i for i, ls in enumerate(rep_steps) if torch.is_tensor(ls[0])]                  # (a skipped step returns plain zeros)
        rep_is_zero = [True] * len(rep_steps)
        if live:
            for i, z in zip(live, (torch.stack([rep_steps[i][0].detach() for i in live]) == 0).tolist()):      # one read for all steps
                rep_is_zero[i] = z
        for step, acts in enumerate(ca_layers_activations_list):
            pred_l2s.append((noise_preds[step] ** 2).mean())
            reps.append([zero() for _ in range(5)] if rep_is_zero[step] else list(rep_steps[step]))
            if not all_ss_contain_faces or first_step == -1:
                continue
            if step < len(ca_layers_activations_list) - 1 and step >= first_step - 1:
                cross_t.append(CL.calc_subj_attn_cross_t_diff_loss(acts, ca_layers_activations_list[step + 1], all_subj_indices_1b))
            if step < first_step or kind == "sc-noface":
                continue
            preserve.append(CL.calc_comp_subj_bg_preserve_loss(
                mon_loss_dict, P, dev, getattr(self, "flow_model", None), acts, ss_boxes_list[step], sc_boxes, sc_face_shrink_ratio_for_bg_matching_mask=bg_match_shrink,
                recon_scaled_loss_threses={"mc": 0.4, "ssfg": 0.4}, recon_max_scale_of_threses=5, do_sc_fg_faces_suppress=suppress))

        n_preserve = len(preserve) + 1e-6
        rep = [torch.stack([r[i] for r in reps]).mean() for i in range(5)]
        if preserve:
            l_preserve = torch.stack(preserve).mean()
            mon_loss_dict[f"{P}/comp_fg_bg_preserve"] = l_preserve.mean().detach()
            loss = loss + l_preserve
        if cross_t:
            mon_loss_dict[f"{P}/subj_attn_cross_t_diff"] = torch.stack(cross_t).mean().detach()       # monitored, weight 0
        if not all(rep_is_zero):                                  # rep[0] > 0: the mean of the steps' attention terms, each >= 0
            for name, v in zip(("subj_attn", "subj_k", "nonsubj_k", "subj_v", "nonsubj_v"), rep):
                mon_loss_dict[f"{P}/comp_rep_distill_{name}"] = v.detach()
            l_rep = CL.comp_rep_distill_total(rep, sc_pct, rep_dist_fg_bounds)
            mon_loss_dict[f"{P}/comp_rep_distill_total"] = l_rep.mean().detach()
            loss = loss + l_rep
        mon_loss_dict[f"{P}/pred_l2"] = torch.stack(pred_l2s).mean().detach()
        # every monitor left as a device tensor above, and the total, read in ONE go
        (v,) = resolve_monitors(mon_loss_dict, extra=[loss])
        if v > 0:
            mon_loss_dict[f"{P}/comp_feat_distill_total"] = v
        for name in COMP_MONITOR_NAMES:                           # per-step sums -> means, 'loss_' dropped from the key, zeros removed
            key = f"{P}/{name}"
            if key in mon_loss_dict:
                if mon_loss_dict[key] > 0:
                    mon_loss_dict[key.replace("loss_", "")] = mon_loss_dict.pop(key) / n_preserve
                else:
                    del mon_loss_dict[key]
        return loss

    # ------------------------------------------------------------------ adversarial face edit (ddpm.py:2536-2582)
    def calc_arcface_adv_grad(self, x_start):
        """Gradient, w.r.t. the latents, of the mean squared (30 %-dropped) face embedding of their decoded faces -- through the VAE
        decoder's and ResNetFace-18's input-gradient nodes -- kept inside the detected face boxes (latent coordinates); None when any
        instance shows no face."""
        x = x_start.detach().requires_grad_(True)
        with torch.enable_grad():
            image = self.decode_first_stage_with_grad(x)
            emb_centre, _, _, boxes, _, found = self.arcface.embed_image_tensor(image, T=20, enable_grad=True, fg_faces_grad_mask_ratios=(0.9, 0.9))
            if (1 - found).sum() > 0:
                print(f"Failed to detect faces in {int((1 - found).sum())} image, unable to compute adv_grad.")
                return None
            loss = (F.dropout(emb_centre, p=0.3, training=True) ** 2).mean()
            (adv_grad,) = torch.autograd.grad(loss, x)
        boxes = CL.map_bboxes_coords(boxes, image.shape[-1], x_start.shape[-1])
        face_mask = torch.zeros_like(adv_grad)
        for i in range(x_start.shape[0]):
            x1, y1, x2, y2 = boxes[i]
            face_mask[i, :, y1:y2, x1:x2] = 1
        return adv_grad * face_mask

    # ------------------------------------------------------------------ do_normal_recon iteration (ddpm.py:1753-1917, 2593-2883)
    def recon_multistep_denoise(self, mon_loss_dict, session_prefix, x_start0, noise, t, subj_context, cls_context, uncond_emb, img_mask, fg_mask,
                                cfg_scale, num_denoising_steps, num_priming_steps, normal_recon_on_pure_noise, enable_unet_attn_lora,
                                enable_unet_ffn_lora, ffn_lora_adapter_name, do_adv_attack, DO_ADV_BS):
        """``num_denoising_steps`` passes at successively earlier timesteps.  On images every step restarts from the input latents; on
        pure noise each step continues from the previous x0 prediction, the first ``num_priming_steps`` without gradient and
        alternating class / subject prompt.  Every step also gets a no-grad pass under the class prompt (the background target)."""
        assert num_denoising_steps <= 10
        x_starts, noises, ts = [x_start0], [noise], [t]
        noise_preds, x_recons, acts_list = [], [], []
        noise_preds_cls, x_recons_cls = ([], []) if cls_context is not None else (None, None)
        for i in range(num_denoising_steps):
            x_start, t, noise = x_starts[i], ts[i], noises[i]
            priming = i < num_priming_steps
            context = cls_context if (priming and cls_context is not None and i % 2 == 0) else subj_context
            uc = {} if self.cache_uncond_in_step else None      # this step's null-prompt prediction, shared by its guided passes
            # Round 5: on a NON-priming step the class-prompt pass (gradient-free, no capture) goes FIRST when it can take the step's null-prompt
            # rows along in its own U-Net call (guided_denoise, batch_cond_with_uncond): the main pass -- with gradients, so it cannot -- then finds
            # the null prediction in `uc`.  Both are pure functions of (x_start, noise, t, prompt): the order changes no value and no draw.
            cls_first = None
            if (cls_context is not None and not priming and uc is not None and self.batch_cond_with_uncond and cfg_scale > 1 and img_mask is None
                    and not enable_unet_attn_lora and not enable_unet_ffn_lora):
                cls_first = self.guided_denoise(
                    x_start, noise, t, cls_context, uncond_emb, img_mask, subj_indices=None, normalize_cross_attn=False, mix_sc_mc_attn=False,
                    batch_part_has_grad="none", do_pixel_recon=True, cfg_scale=cfg_scale, capture_ca_activations=False,
                    res_hidden_states_gradscale=0, use_attn_lora=enable_unet_attn_lora, use_ffn_lora=enable_unet_ffn_lora,
                    ffn_lora_adapter_name=ffn_lora_adapter_name, uncond_cache=uc)
            noise_pred, x_recon, acts = self.guided_denoise(
                x_start, noise, t, context, uncond_emb, img_mask, subj_indices=None, normalize_cross_attn=False, mix_sc_mc_attn=False,
                batch_part_has_grad="none" if priming else "all", do_pixel_recon=True, cfg_scale=cfg_scale, capture_ca_activations=not priming,
                res_hidden_states_gradscale=self.res_hidden_states_gradscale, use_attn_lora=enable_unet_attn_lora,
                use_ffn_lora=enable_unet_ffn_lora and not priming, ffn_lora_adapter_name=ffn_lora_adapter_name, uncond_cache=uc)
            noise_preds.append(noise_pred)
            acts_list.append(acts)
            x_recons.append(x_recon)
            x_starts.append(x_recon if (normal_recon_on_pure_noise or priming) else x_start0)
            if cls_context is not None and self.skip_unread_cls_priming and priming:
                # the class-prompt prediction of a PRIMING step is read by nobody (the losses start at step num_priming_steps, and the chain
                # continues from the main pass): the reference computes it all the same -- on even priming steps it even is the main pass
                # over again, argument for argument; here the slot keeps the main pass's tensors where they are that, None otherwise (the lists
                # stay aligned with the step index; calc_normal_recon_loss asserts that every entry it reads exists)
                same = context is cls_context and not enable_unet_ffn_lora
                noise_preds_cls.append(noise_pred if same else None)
                x_recons_cls.append(x_recon if same else None)
            elif cls_first is not None:
                noise_preds_cls.append(cls_first[0])
                x_recons_cls.append(cls_first[1])
            elif cls_context is not None:
                eps_cls, x_cls, _ = self.guided_denoise(
                    x_start, noise, t, cls_context, uncond_emb, img_mask, subj_indices=None, normalize_cross_attn=False, mix_sc_mc_attn=False,
                    batch_part_has_grad="none", do_pixel_recon=True, cfg_scale=cfg_scale, capture_ca_activations=False,
                    res_hidden_states_gradscale=0, use_attn_lora=enable_unet_attn_lora, use_ffn_lora=enable_unet_ffn_lora,
                    ffn_lora_adapter_name=ffn_lora_adapter_name, uncond_cache=uc)
                noise_preds_cls.append(eps_cls)
                x_recons_cls.append(x_cls)
            if i < num_denoising_steps - 1:
                p = np.power(num_denoising_steps - 1, -0.3)
                t_lb, t_ub = t * np.power(0.5, p), t * np.power(0.7, p)
                ts.append(((t_ub - t_lb) * torch.rand_like(t.float()) + t_lb).long())
                noise = torch.randn_like(x_start)
                if do_adv_attack:                       # the adversarial face edit of the NEXT step's noise (ddpm.py:1879-1913)
                    adv_grad = self.calc_arcface_adv_grad(x_start[:DO_ADV_BS])
                    self.adaface_adv_iters_count += 1
                    if adv_grad is not None:
                        adv_max = adv_grad.abs().max().detach().item()
                        adv_fg_mean = adv_grad[fg_mask[:DO_ADV_BS].repeat(1, 4, 1, 1).bool()].abs().mean().detach().item()
                        mod_mag = CL.torch_uniform(*self.recon_adv_mod_mag_range).item()
                        scale = mod_mag / (np.sqrt(adv_max * adv_fg_mean) + 1e-6)
                        mon_loss_dict.update({f"{session_prefix}/adv_grad_max": adv_max, f"{session_prefix}/adv_grad_fg_mean": adv_fg_mean,
                                              f"{session_prefix}/adv_grad_scale": scale})
                        noise[:DO_ADV_BS] -= adv_grad * min(scale, 10)
                        self.adaface_adv_success_iters_count += 1
                noises.append(noise)
        return noise_preds, noise_preds_cls, x_starts, x_recons, x_recons_cls, noises, ts, acts_list

    def calc_normal_recon_loss(self, mon_loss_dict, session_prefix, num_denoising_steps, num_recon_priming_steps, x_start, noise, subj_context,
                               cls_context, img_mask, fg_mask, all_subj_indices, recon_bg_pixel_weight, normal_recon_on_pure_noise,
                               enable_unet_attn_lora, enable_unet_ffn_lora, ffn_lora_adapter_name, do_adv_attack, DO_ADV_BS):
        """eps-reconstruction of the input images under the subject prompt (fg weight 1, bg ``recon_bg_pixel_weight``; restricted to the
        detected face box when one is found, x0.1 when none is), the background pulled to the class-prompt prediction, the subject
        attention kept off the background, plus the ArcFace alignment of the x0 prediction.
        Like the reference, the per-step losses only exist when ``arcface_align_loss_weight > 0`` (:2702)."""
        from ...modules.arcface_wrapper import to_device_async
        P, dev = session_prefix, x_start.device
        loss = torch.zeros((), device=dev)
        BS = x_start.shape[0]
        ref_cache = {}                               # the decoded inputs and their face embeddings: the same for every step
        if normal_recon_on_pure_noise:
            t = torch.randint(int(self.num_timesteps * 0.7), int(self.num_timesteps * 0.9), (BS,), device=dev).long()
            x_start0 = torch.randn_like(x_start)
            num_denoising_steps += num_recon_priming_steps
        else:
            t = torch.randint(int(self.num_timesteps * 0.5), int(self.num_timesteps * 0.8), (BS,), device=dev).long()
            x_start0 = x_start
        if num_denoising_steps > 1 or normal_recon_on_pure_noise:
            uncond_emb, cfg_scale = self.uncond_context[0].repeat(BS, 1, 1), 2
            if normal_recon_on_pure_noise:
                img_mask, fg_mask = None, torch.ones_like(fg_mask)
        else:
            uncond_emb, cfg_scale = None, -1
        # (the reference decodes the inputs and every step's x0 predictions here for its image log, :2633-2694: pure logging, skipped)
        noise_preds, noise_preds_cls, x_starts, x_recons, x_recons_cls, noises, ts, acts_list = self.recon_multistep_denoise(
            mon_loss_dict, P, x_start0, noise, t, subj_context, cls_context, uncond_emb, img_mask, fg_mask, cfg_scale, num_denoising_steps,
            num_recon_priming_steps, normal_recon_on_pure_noise, enable_unet_attn_lora, enable_unet_ffn_lora, ffn_lora_adapter_name,
            do_adv_attack, DO_ADV_BS)
        l_recon, l_cls, scales, l_mb, l_align, l_align_stat, l_bgf, pred_l2s = [], [], [], [], [], [], [], []
        face_stats = self.normal_recon_face_images_on_noise_stats if normal_recon_on_pure_noise else self.normal_recon_face_images_on_image_stats
        for i in range(num_recon_priming_steps, num_denoising_steps):
            noise, noise_pred, x_recon, acts = noises[i], noise_preds[i], x_recons[i], acts_list[i]
            noise_pred_cls = noise_preds_cls[i] if cls_context is not None else None
            assert cls_context is None or noise_pred_cls is not None, "a non-priming step must carry its class-prompt prediction (skip_unread_cls_priming only drops priming steps')"
            pred_l2s.append((noise_pred ** 2).mean())
            if self.arcface_align_loss_weight > 0:
                la, _, lb, boxes, _, found = self.calc_arcface_align_loss(x_start, x_recon, fg_faces_grad_mask_ratios=(1, 0.3), ref_cache=ref_cache)
                face_stats.update([found.sum().item(), found.shape[0]])                       # (host data)
                la_v = lb_v = 0.0
                if boxes is not None:                # one read of the two values the branches below are on (exact zeros without a face)
                    la_v, lb_v = torch.stack([la.detach().float(), lb.detach().float()]).tolist()
                if la_v > 0:
                    keep = self.recon_face_align_loss_thres <= 0 or la_v < float(np.float32(self.recon_face_align_loss_thres))
                    if keep:
                        l_align.append(la)
                    self.normal_recon_face_align_loss_kept_frac.update(1 if keep else 0)
                    l_align_stat.append(la)
                    # the reference's arithmetic, kept literally (ddpm.py:2738-2739): the mask is a LONG tensor (retinaface_pytorch.py:236), so the
                    # 0.1 its comment intends for instances without a face truncates to 0 -- those instances get weight 0, not 0.1
                    inst_w = found.clone()
                    inst_w[found == 0] = 0.1
                    inst_w = to_device_async(inst_w, dev)
                    scale = 1.0
                    fg_mask2 = fg_mask * box_mask(boxes, BS, x_start.shape[-2], x_start.shape[-1], dev)
                else:
                    scale, inst_w, fg_mask2 = 0.1, torch.ones(found.shape, dtype=torch.float32, device=dev), fg_mask
                a, b, c = CL.calc_recon_and_suppress_losses(noise, noise_pred, noise_pred_cls, inst_w, acts, all_subj_indices, None, fg_mask2,
                                                            recon_bg_pixel_weight, BS, normal_recon_on_pure_noise)
                l_recon.append(a)
                l_cls.append(b)
                scales.append(scale)
                l_mb.append(c)
                if lb_v > 0:
                    l_bgf.append(lb)
        kind = "noise" if normal_recon_on_pure_noise else "image"
        on_noise, on_image = self.normal_recon_face_images_on_noise_stats, self.normal_recon_face_images_on_image_stats
        mon_loss_dict[f"{P}/recon_face_images_on_noise_frac"] = on_noise.sums[0] / (on_noise.sums[1] + 1e-2)
        mon_loss_dict[f"{P}/recon_face_images_on_image_frac"] = on_image.sums[0] / (on_image.sums[1] + 1e-2)
        mon_loss_dict[f"{P}/recon_face_align_loss_kept_frac"] = self.normal_recon_face_align_loss_kept_frac.mean
        align_scale = 1
        if l_align:                                  # (every entry was read > 0 above, so is their mean: no 'if la > 0' read)
            la = torch.stack(l_align).mean()
            mon_loss_dict[f"{P}/arcface_align_recon_on_{kind}_opt"] = la.mean().detach()
            align_scale = 4 if normal_recon_on_pure_noise else 1
            loss = loss + la * self.arcface_align_loss_weight * align_scale
        if l_align_stat:
            mon_loss_dict[f"{P}/arcface_align_recon_on_{kind}"] = torch.stack(l_align_stat).mean().detach()
        if l_bgf:
            lb = torch.stack(l_bgf).mean()
            mon_loss_dict[f"{P}/recon_bg_faces_suppress"] = lb.mean().detach()
            loss = loss + lb * 2 * align_scale
        mon_loss_dict[f"{P}/pred_l2"] = torch.stack(pred_l2s).mean().detach()
        l_mb = torch.stack(l_mb).mean()
        mon_loss_dict[f"{P}/recon_subj_mb_suppress"] = l_mb.mean().detach()               # kept below only if > 0
        scales = to_device_async(torch.tensor(scales, dtype=torch.float32), dev)
        if not normal_recon_on_pure_noise:
            l_recon = torch.stack(l_recon)
            mon_loss_dict[f"{P}/loss_recon"] = l_recon.mean().detach()
            loss = loss + (l_recon * scales).mean() + l_mb * self.recon_subj_mb_suppress_loss_weight
        if cls_context is not None:
            l_cls = torch.stack(l_cls)
            loss = loss + (l_cls * scales).mean()
            mon_loss_dict[f"{P}/loss_recon_cls"] = l_cls.mean().detach()
        mon_loss_dict[f"{P}/normal_recon_total"] = loss.mean().detach()
        resolve_monitors(mon_loss_dict)                                                    # all of the above in one device read
        if not mon_loss_dict[f"{P}/recon_subj_mb_suppress"] > 0:
            del mon_loss_dict[f"{P}/recon_subj_mb_suppress"]
        return loss
