"""One compositional-distillation iteration's loss assembly and one normal-recon iteration's, written ONCE against the
``LatentDiffusion`` method surface and run on both sides: tests/golden/gen_golden.py drives the REFERENCE class (a constructor-free
shell of ``ldm.models.diffusion.ddpm.LatentDiffusion``) through it to write the fixtures, the tests drive this package's mirror.  The
U-Net wrapper, the VAE decoder, the face detector and the face-embedding network are the same small stand-ins on both sides
(tests/standin.py); everything between them -- gating, loss functions, weights, the re-denoising pass -- is the code under test."""
import contextlib
import io

import numpy as np
import torch

from standin import StandInCaptureWrapper, StandInEps, standin_decode

B4, H, T, D, S = 4, 16, 12, 16, 3


def common_attrs(ld, device):
    """Attributes both shells need (values of the reference's constructor defaults, ddpm.py:84-127)."""
    from adaface_dev_amd import rng
    ld.model = StandInCaptureWrapper(StandInEps(D, seed=61), ctx_dim=D).to(device)
    ld.uncond_context = (rng.synth_input("s2.uncond", (1, T, D), seed=81).to(device), [""], {})
    ld.res_hidden_states_gradscale = 0.5
    ld.arcface_align_loss_weight = 1e-2
    ld.comp_ss_face_confidence_thres, ld.comp_ss_face_lap_vars_tolerance = 0.99, 0.3
    ld.comp_sc_fg_mask_percent_range, ld.comp_sc_face_align_loss_thres = [0.0225, 0.36], 0.7
    ld.comp_sc_subj_mb_suppress_loss_weight, ld.recon_subj_mb_suppress_loss_weight = 0.2, 0.2
    ld.redenoise_subj_comp_crop_mix_weights = (0.5, 0.25, 0.25)
    ld.num_comp_distill_denoising_steps = S
    ld.comp_iters_count, ld.comp_iters_bg_has_face_count = 1, 0
    ld.recon_face_align_loss_thres = 0.8
    ld.flow_model = None
    ld.decode_first_stage = standin_decode
    ld.decode_first_stage_with_grad = standin_decode
    ld.cache_and_log_generations = lambda *a, **k: None
    return ld


def comp_inputs(device):
    from adaface_dev_amd import rng
    x0 = rng.synth_input("s2.x0", (1, 4, H, H), seed=81).to(device)
    xs = [torch.cat([rng.synth_input(f"s2.xs{i}", (1, 4, H, H), seed=81).to(device)] * 1 + [rng.synth_input(f"s2.xc{i}", (1, 4, H, H), seed=81).to(device)] * 3)
          for i in range(1)]
    noises = [rng.synth_input(f"s2.n{i}", (1, 4, H, H), seed=81).repeat(B4, 1, 1, 1).to(device) for i in range(S)]
    ts = [torch.tensor([t]).repeat(B4).to(device) for t in (560, 400, 290)]
    emb = rng.synth_input("s2.emb", (B4, T, D), seed=81).to(device).requires_grad_(True)
    subj = (torch.zeros(3, dtype=torch.long, device=device), torch.tensor([3, 4, 5], device=device))
    emb_mask = torch.zeros(B4, T, 1, device=device)
    emb_mask[:, 1:7] = 1
    emb_mask[1, 7:9] = 1
    pad_mask = torch.zeros(B4, T, 1, device=device)
    pad_mask[:, 10:] = 1
    return x0, xs, noises, ts, emb, subj, emb_mask, pad_mask


def run_comp_feat_distill(ld, device, mix_sc_mc_attn=False):
    """-> dict(loss, demb, mon..., kind) of one assembly call on seeded inputs."""
    x0, xs, noises, ts, emb, subj, emb_mask, pad_mask = comp_inputs(device)
    prompts = ["ss", "sc", "sr", "mc"]
    ctx = (emb, prompts, {})
    torch.manual_seed(1357)
    with contextlib.redirect_stdout(io.StringIO()):
        noise_preds, x_starts, x_recons, noises, ts, acts = ld.comp_distill_multistep_denoise(
            xs, noises, ts, ctx, uncond_emb=ld.uncond_context[0].repeat(B4, 1, 1), all_subj_indices_1b=subj, normalize_cross_attn=not mix_sc_mc_attn,
            mix_sc_mc_attn=mix_sc_mc_attn, cfg_scale=2.5, num_denoising_steps=S, old_x_starts_mix_ratio=0, use_attn_lora=True, use_ffn_lora=True,
            ffn_lora_adapter_name="comp_distill", batch_part_has_grad="subject-compos")
        pixels = [standin_decode(x).detach() for x in x_recons]
        ss_context = (emb.chunk(4)[0], prompts[:1], {})
        mon = {}
        loss = ld.calc_comp_feat_distill_loss(mon, "train", x0, x_starts, x_recons, pixels, noise_preds, noises, ts, acts, subj, ss_context,
                                              ld.uncond_context[0], emb_mask, pad_mask, 1, 0.3, use_attn_lora=True, use_ffn_lora=True)
    res = {"loss": np.asarray(float(loss.detach()))}
    if loss.requires_grad:
        loss.backward()
        res["demb"] = emb.grad.detach().cpu().numpy()
    for k, v in mon.items():
        res["mon." + k.replace("/", "__")] = np.asarray(float(v))
    return res


def recon_inputs(device):
    from adaface_dev_amd import rng
    BS = 2
    x0 = rng.synth_input("s2.rx0", (BS, 4, H, H), seed=82).to(device)
    noise = rng.synth_input("s2.rn", (BS, 4, H, H), seed=82).to(device)
    emb = rng.synth_input("s2.remb", (BS, T, D), seed=82).to(device).requires_grad_(True)
    cls_emb = rng.synth_input("s2.rcls", (BS, T, D), seed=82).to(device)
    fg = torch.zeros(BS, 1, H, H, device=device)
    fg[:, :, 3:12, 4:13] = 1
    img_mask = torch.ones(BS, 1, H, H, device=device)
    img_mask[:, :, :, :2] = 0
    subj = (torch.arange(BS, device=device).repeat_interleave(3), torch.tensor([3, 4, 5], device=device).repeat(BS))
    return x0, noise, emb, cls_emb, fg, img_mask, subj


def run_normal_recon(ld, device, on_pure_noise=False, steps=2, do_adv=False):
    x0, noise, emb, cls_emb, fg, img_mask, subj = recon_inputs(device)
    extra = {}
    subj_context, cls_context = (emb, ["a", "b"], extra), (cls_emb, ["a", "b"], extra)
    mon = {}
    torch.manual_seed(2468)
    with contextlib.redirect_stdout(io.StringIO()):
        loss = ld.calc_normal_recon_loss(mon, "train", steps, 4 if on_pure_noise else 0, x0, noise, subj_context, cls_context, img_mask, fg, subj, 0.025,
                                         on_pure_noise, True, False, "recon_loss", do_adv, 2)
    res = {"loss": np.asarray(float(loss.detach()))}
    loss.backward()
    res["demb"] = emb.grad.detach().cpu().numpy()
    for k, v in mon.items():
        res["mon." + k.replace("/", "__")] = np.asarray(float(v))
    return res
