"""The loss ASSEMBLIES of the compositional-distillation and normal-recon iterations (adaface_dev_amd/ldm/models/diffusion/ddpm_losses.py:
``calc_comp_feat_distill_loss``, ``calc_comp_face_align_and_mb_suppress_losses``, ``redenoise_subj_single``, ``calc_arcface_align_loss``,
``calc_normal_recon_loss``, ``recon_multistep_denoise``; modules/arcface_wrapper.py) against ``stage2_assembly.npz``, which the REFERENCE
methods wrote when driven through the very same scenario code (tests/stage2_scenario.py) on a constructor-free shell
(gen_golden.py::gen_stage2_assembly): the loss, EVERY monitor entry (same keys, same values) and d loss / d prompt_emb.

Here on the CPU the one HIP leaf on the path, ``q_sample``, is replaced by its two-line torch formula (the mirror itself has no CPU
path); tests/test_hip_orchestration.py runs the same cases on device tensors with the real one."""
import os

import numpy as np
import pytest
import torch

import stage2_scenario as SC
import standin
from conftest import GOLDEN, rel_l2

DETECTORS = ("standin_detect", "no_faces", "standin_detect_small_second_face", "standin_detect_dark_images_faceless")


def mirror_shell(device, dname, torch_q_sample=False):
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.arcface_wrapper import ArcFaceWrapper, FaceCropper, no_faces
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import extract_into_tensor
    ld = LatentDiffusion.__new__(LatentDiffusion)
    torch.nn.Module.__init__(ld)
    ld.parameterization = "eps"
    ld.register_schedule(beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.012)
    ld = ld.to(device)
    SC.common_attrs(ld, device)
    detect = no_faces if dname == "no_faces" else getattr(standin, dname)
    ld.arcface = ArcFaceWrapper(standin.StandInFaceNet().to(device), FaceCropper(detect), dtype=torch.float32)
    if torch_q_sample:
        ld.q_sample = lambda x, t, noise=None: (extract_into_tensor(ld.sqrt_alphas_cumprod, t, x.shape) * x
                                                + extract_into_tensor(ld.sqrt_one_minus_alphas_cumprod, t, x.shape) * noise)
    return ld


def check_case(g, prefix, res, tol):
    want_keys = sorted(k[len(prefix):] for k in g.files if k.startswith(prefix))
    assert sorted(res) == want_keys, (prefix, sorted(set(res) ^ set(want_keys)))
    for k in want_keys:
        w, v = g[prefix + k], res[k]
        if k == "demb":
            assert rel_l2(v, w) < 20 * tol, (prefix, k, rel_l2(v, w))
        else:
            assert abs(float(v) - float(w)) <= tol * max(abs(float(w)), 1e-2), (prefix, k, float(v), float(w))


def run_comp_cases(device, tol, torch_q_sample):
    g = np.load(os.path.join(GOLDEN, "stage2_assembly.npz"))
    for dname in DETECTORS:
        for mix in (False, True):
            res = SC.run_comp_feat_distill(mirror_shell(device, dname, torch_q_sample), device, mix_sc_mc_attn=mix)
            check_case(g, f"comp.{dname}.mix{int(mix)}.", res, tol)
    assert float(g["comp.standin_detect.mix0.loss"]) > 0 and float(g["comp.no_faces.mix0.loss"]) == 0
    assert "comp.standin_detect.mix0.mon.train__comp_fg_bg_preserve" in g.files       # the feature-matching loss was live in the fixture


def run_recon_cases(device, tol, torch_q_sample):
    g = np.load(os.path.join(GOLDEN, "stage2_assembly.npz"))
    for dname in DETECTORS:
        for pure, steps in ((False, 2), (False, 1), (True, 2)):
            res = SC.run_normal_recon(mirror_shell(device, dname, torch_q_sample), device, on_pure_noise=pure, steps=steps)
            check_case(g, f"recon.{dname}.pure{int(pure)}.steps{steps}.", res, tol)
        # with the adversarial face edit of the second step's noise (the gradient of the decoded faces' embedding w.r.t. the latents)
        ld = mirror_shell(device, dname, torch_q_sample)
        res = SC.run_normal_recon(ld, device, on_pure_noise=False, steps=2, do_adv=True)
        res["adv_iters"], res["adv_success"] = np.asarray(float(ld.adaface_adv_iters_count)), np.asarray(float(ld.adaface_adv_success_iters_count))
        check_case(g, f"recon_adv.{dname}.", res, tol)
        if dname == "standin_detect":
            assert "mon.train__adv_grad_scale" in res and float(res["adv_iters"]) == float(res["adv_success"]) == 1


def test_calc_comp_feat_distill_loss_mirror_vs_reference_cpu():
    run_comp_cases("cpu", 2e-5, torch_q_sample=True)


def test_calc_normal_recon_loss_mirror_vs_reference_cpu():
    run_recon_cases("cpu", 2e-5, torch_q_sample=True)


def test_decode_with_grad_needs_the_first_stage_and_embedding_modules_may_opt_out():
    """decode_first_stage_with_grad is the decoder's autograd node: without a first-stage model it raises instead of detaching; an
    embedding module that declares ``inference_only`` is refused for a tensor that needs its input gradient (never silently detached)."""
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.arcface_wrapper import ArcFaceWrapper
    ld = LatentDiffusion.__new__(LatentDiffusion)
    torch.nn.Module.__init__(ld)
    ld.first_stage_model = None
    z = torch.zeros(1, 4, 8, 8, requires_grad=True)
    with pytest.raises(RuntimeError):
        ld.decode_first_stage_with_grad(z)

    class Frozen(torch.nn.Module):
        inference_only = True

        def forward(self, g):
            return g.flatten(1)[:, :512] * 2.0

    w = ArcFaceWrapper(Frozen(), dtype=torch.float32)
    grey = torch.zeros(1, 1, 128, 128, requires_grad=True)
    with pytest.raises(NotImplementedError):
        w._embed(grey, True)
    assert not w._embed(grey, False).requires_grad
