"""The HOST-side mirrors of the reference's orchestration (``adaface/unet_teachers.py::UNetTeacher.forward``,
``LatentDiffusion.guided_denoise`` / ``sliced_apply_model`` / ``prepare_unet_teacher_context`` / ``calc_unet_distill_loss``) against
fixtures written by the REFERENCE methods themselves (tests/golden/gen_golden.py imports them from /root/reference), around the same
stand-in eps-model (tests/standin.py) placed where the reference puts its U-Net.  ``-m gpu``: the mirrors run q_sample through the HIP
extension and have no CPU path; the stand-in is plain torch on the GPU.  fp32 throughout => tight bounds."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from standin import StandInEps, StandInWrapper
from test_host_orchestration import GUIDED_CASES, TEACHER_CASES, check_guided, distill_inputs, guided_inputs, teacher_inputs

pytestmark = pytest.mark.gpu
TOL = 5e-6
CFG = dict(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=2, attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4],
           num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=16, legacy=False)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


class _AsUNet(torch.nn.Module):
    """The teacher mirror calls its U-Net as unet(x, t, context, extra_info=None)."""

    def __init__(self, eps_model):
        super().__init__()
        self.eps_model = eps_model

    def forward(self, x, t, ctx, extra_info=None):
        return self.eps_model(x, t, ctx)


def _ld(dev, wrapper=None):
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    ld = LatentDiffusion(CFG).to(dev)
    if wrapper is not None:
        ld.model = wrapper.to(dev)
    return ld


@pytest.mark.parametrize("case", TEACHER_CASES, ids=[c["name"] for c in TEACHER_CASES])
def test_unet_teacher_mirror_vs_reference(dev, case):
    from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
    g, k = np.load(os.path.join(GOLDEN, "teacher.npz")), case["name"]
    inp = {n: v.to(dev) for n, v in teacher_inputs().items()}
    teacher = UNetTeacher(_AsUNet(StandInEps(16, seed=61)).to(dev), cfg_scale_range=[1.3, 2], p_uses_cfg=case["p"])
    ctx = torch.cat([inp["pos"], inp["neg"]]) if case["ctx"] == "doubled" else inp["pos"]
    pres = [(torch.from_numpy(g[f"{k}.rel{i}"]).to(dev), torch.from_numpy(g[f"{k}.drawn_noise{i}"]).to(dev)) for i in range(case["steps"] - 1)]
    np.random.seed(77)                                       # the reference's numpy draws: the p_uses_cfg coin, then the cfg scale
    preds, xs, ns, ts = teacher(_ld(dev), inp["x0"], inp["noise"], inp["t"], ctx, negative_context=inp["neg"] if case["neg"] else None,
                                num_denoising_steps=case["steps"], force_uses_cfg=case["force"],
                                same_t_noise_across_instances=case["same"], presampled=pres)
    assert bool(teacher.uses_cfg) == bool(g[f"{k}.uses_cfg"]) and abs(float(teacher.cfg_scale) - float(g[f"{k}.cfg_scale"])) < 1e-12
    for i in range(case["steps"]):
        assert np.array_equal(ts[i].cpu().numpy(), g[f"{k}.t{i}"]), (k, i)
        assert rel_l2(ns[i].cpu().numpy(), g[f"{k}.noise{i}"]) < 1e-7
        assert rel_l2(preds[i].cpu().numpy(), g[f"{k}.eps{i}"]) < TOL, (k, i)
        assert rel_l2(xs[i + 1].cpu().numpy(), g[f"{k}.x{i + 1}"]) < 2e-5, (k, i)


def test_unet_teacher_mirror_draws_like_the_reference(dev):
    """Without presampled draws the mirror consumes the global torch RNG in the reference's order (rand_like(t) then randn_like(x0) per
    extra step): on the CPU generator the fixture's timesteps are reproduced exactly when the tensors live on the CPU side of the draw."""
    from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
    g, k = np.load(os.path.join(GOLDEN, "teacher.npz")), "nocfg_4step"
    torch.manual_seed(1234)
    draws = [(torch.rand(3), torch.randn(3, 4, 8, 8)) for _ in range(3)]
    for i, (r, n) in enumerate(draws):
        assert np.array_equal(r.numpy(), g[f"{k}.rel{i}"]) and np.array_equal(n.numpy(), g[f"{k}.drawn_noise{i}"])


@pytest.mark.parametrize("case", GUIDED_CASES, ids=[c["name"] for c in GUIDED_CASES])
def test_guided_denoise_mirror_vs_reference(dev, case):
    g, c = np.load(os.path.join(GOLDEN, "guided_denoise.npz")), case
    i = guided_inputs(c, dev)
    wrapper = StandInWrapper(StandInEps(16, seed=61))
    ld = _ld(dev, wrapper)
    ld.uncond_context = (i["un"], [""], {})
    cond = (i["emb"], [f"p{j}" for j in range(i["B"])], {})
    torch.manual_seed(4321)                                  # the FFN-LoRA coin of subject-compos: torch.rand(1) on the CPU generator
    eps, recon, acts = ld.guided_denoise(i["x0"], i["noise"], i["t"], cond, uncond_emb=i["uncond"], img_mask=i["mask"],
                                         normalize_cross_attn=c.get("norm", False), mix_sc_mc_attn=c.get("mix", False),
                                         batch_part_has_grad=c["mode"], do_pixel_recon=c["recon"], cfg_scale=c["cfg"],
                                         capture_ca_activations=c["capture"], res_hidden_states_gradscale=c.get("gradscale", 1),
                                         use_attn_lora=c.get("attn_lora", False), use_ffn_lora=c.get("ffn", False),
                                         ffn_lora_adapter_name="comp_distill" if c.get("ffn") else None)
    check_guided(c, g, eps, recon, acts, i["emb"], wrapper.calls, tol=TOL)


@pytest.mark.parametrize("steps,pcfg", [(1, 0), (3, 0), (2, 1)])
def test_calc_unet_distill_loss_mirror_vs_reference(dev, steps, pcfg):
    """prepare_unet_teacher_context + UNetTeacher + guided_denoise per step + fg-masked MSE / sqrt(steps), exactly the reference's
    on-image branch (ddpm.py:2984-3184): loss and d loss / d prompt_emb, per-step (the reference's loop) and batched student passes."""
    from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
    g, k = np.load(os.path.join(GOLDEN, "distill_loss.npz")), f"steps{steps}_pcfg{pcfg}"
    for batched in (False, True):
        i = distill_inputs(dev)
        student = StandInWrapper(StandInEps(16, seed=61))
        student.ffn_lora = object()                          # "adapters exist": the mirror turns use_ffn_lora on like the reference always does
        ld = _ld(dev, student)
        ld.batch_student_steps = batched
        ld.uncond_context = (i["un"], [""], {})
        ld.unet_teacher = UNetTeacher(_AsUNet(StandInEps(16, seed=65)).to(dev), cfg_scale_range=[1.3, 2], p_uses_cfg=float(pcfg), name="arc2face")
        tctx = ld.prepare_unet_teacher_context(None, ld.uncond_context, i["B"], i["id2img"], None, i["prefix"], ["arc2face"], None, float(pcfg), False)
        pres = [(torch.from_numpy(g[f"{k}.rel{j}"]).to(dev), torch.from_numpy(g[f"{k}.drawn_noise{j}"]).to(dev)) for j in range(steps - 1)]
        np.random.seed(99)
        loss = ld.calc_unet_distill_loss(i["x0"], i["noise"], (i["emb"], ["p"] * i["B"], {}), tctx, None, i["fg"], steps,
                                         t=torch.from_numpy(g[f"{k}.t"]).to(dev), presampled=pres)
        loss.backward()
        assert abs(float(ld.unet_teacher.cfg_scale) - float(g[f"{k}.cfg_scale"])) < 1e-12
        assert abs(float(loss.detach()) - float(g[f"{k}.loss"])) < 1e-5 * abs(float(g[f"{k}.loss"])), (batched, float(loss.detach()), float(g[f"{k}.loss"]))
        assert rel_l2(i["emb"].grad.cpu().numpy(), g[f"{k}.demb"]) < 2e-5, batched


def test_comp_distill_multistep_denoise_mirror_vs_reference(dev):
    """``LatentDiffusion.comp_distill_multistep_denoise`` (ddpm.py:1997-2086): three subject-compos denoising steps with capture on the
    four-block batch -- per-step eps / x0 / next x_start / timestep / noise / captured attention, d/d(prompt_emb) and the per-call flag
    log; first with the timesteps and noises drawn inside (the reference's draws are replayed from its seed on the CPU generator and
    handed over), then re-using the first pass's trajectory (old-x_start mixing) with SC / MC attention mixing (all LoRAs off)."""
    import json
    from adaface_dev_amd import rng
    g = np.load(os.path.join(GOLDEN, "comp_multistep.npz"))
    B, h, T, D = 4, 8, 6, 16
    wrapper = StandInWrapper(StandInEps(D, seed=61))
    ld = _ld(dev, wrapper)
    un = rng.synth_input("cm.uncond", (B, T, D), seed=67).to(dev)
    ld.uncond_context = (un[:1], [""], {})
    ld.res_hidden_states_gradscale = 0.5
    x0 = rng.synth_input("cm.x0", (1, 4, h, h), seed=67).repeat(B, 1, 1, 1).to(dev)
    noise = rng.synth_input("cm.noise", (1, 4, h, h), seed=67).repeat(B, 1, 1, 1).to(dev)
    emb = rng.synth_input("cm.emb", (B, T, D), seed=67).to(dev).requires_grad_(True)
    t = torch.tensor([900]).repeat(B).to(dev)
    subj = (torch.tensor([0, 0], device=dev), torch.tensor([2, 3], device=dev))
    keep = None
    # the third case repeats the second with SS + SR batched into one no-grad pass (ddpm.batch_no_grad_instances, the default): same tensors,
    # one U-Net call fewer per step; the first two hold the reference's exact call structure
    for tag, kw, reuse, batched in (("draw", dict(normalize_cross_attn=True, mix_sc_mc_attn=False, use_attn_lora=True, use_ffn_lora=True), False, False),
                                    ("mix_reuse", dict(normalize_cross_attn=False, mix_sc_mc_attn=True, use_attn_lora=True, use_ffn_lora=True), True, False),
                                    ("mix_reuse", dict(normalize_cross_attn=False, mix_sc_mc_attn=True, use_attn_lora=True, use_ffn_lora=True), True, True)):
        ld.batch_no_grad_instances = batched
        wrapper.calls.clear()
        if not reuse:
            # the draws happen on the GPU generator here; feed the reference's (recorded) noises / timesteps instead, which the method accepts
            x_starts = [x0]
            noises = [torch.from_numpy(g[f"{tag}.noise{i}"]).to(dev) for i in range(3)]
            ts = [torch.from_numpy(g[f"{tag}.t{i}"]).to(dev) for i in range(3)]
        else:
            x_starts, noises, ts = [x.clone() for x in keep[0]], list(keep[1]), list(keep[2])
        torch.manual_seed(97)                              # the FFN-LoRA coin of every subject-compos pass: CPU generator, as in the reference
        if not reuse:
            # the reference consumed extra CPU draws (randn / rand for the next step) between the coins; replay them to stay aligned
            real_gd = ld.guided_denoise
            step = {"i": 0}

            def gd(*a, **k):
                out = real_gd(*a, **k)
                if step["i"] < 2:
                    torch.randn(1, 4, h, h)
                    torch.rand(1)
                step["i"] += 1
                return out
            ld.guided_denoise = gd
        preds, xs, recons, ns, tss, acts = ld.comp_distill_multistep_denoise(x_starts, noises, ts, (emb, [f"p{i}" for i in range(B)], {}), un,
                                                                            all_subj_indices_1b=subj, cfg_scale=2.5, num_denoising_steps=3,
                                                                            ffn_lora_adapter_name="comp_distill", **kw)
        if not reuse:
            ld.guided_denoise = real_gd
        if not batched:
            keep = ([x.clone() for x in xs], list(ns), list(tss)) if not reuse else keep
        for i in range(3):
            assert np.array_equal(tss[i].cpu().numpy(), g[f"{tag}.t{i}"]), (tag, i)
            assert rel_l2(preds[i].detach().cpu().numpy(), g[f"{tag}.eps{i}"]) < 2e-5, (tag, i)
            assert rel_l2(recons[i].detach().cpu().numpy(), g[f"{tag}.recon{i}"]) < 1e-4, (tag, i)
            assert rel_l2(xs[i].cpu().numpy(), g[f"{tag}.x{i}"]) < 1e-4, (tag, i)
            assert rel_l2(acts[i]["attn"].detach().cpu().numpy(), g[f"{tag}.attn{i}"]) < 2e-5, (tag, i)
        emb.grad = None
        sum(p.sum() for p in preds).backward()
        assert rel_l2(emb.grad.cpu().numpy(), g[f"{tag}.demb"]) < 5e-5, tag
        want_calls = json.loads(str(g[f"{tag}.calls"]))
        got_calls = [[n, fl, ad, npr, ge] for n, fl, ad, npr, ge in wrapper.calls]
        if not batched:
            assert got_calls == want_calls, tag
        else:
            assert len(got_calls) == len(want_calls) - 3 and sum(c[0] for c in got_calls) == sum(c[0] for c in want_calls), (tag, got_calls)


# ----------------------------------------------------------------------------- Stage-2 / recon loss assemblies on device tensors
def _cpu_draws(monkeypatch):
    """The scenario's in-method draws (torch.randn_like / rand_like on device tensors) follow the CPU generator the fixture was written
    with: draw there, move over."""
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: torch.randn(t.shape, dtype=t.dtype).to(t.device))
    monkeypatch.setattr(torch, "rand_like", lambda t, **k: torch.rand(t.shape, dtype=t.dtype).to(t.device))
    real_randint = torch.randint

    def randint(*a, device=None, **k):
        out = real_randint(*a, **k)
        return out if device is None else out.to(device)
    monkeypatch.setattr(torch, "randint", randint)
    import torch.nn.functional as F
    real_dropout = F.dropout
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x * real_dropout(torch.ones(x.shape), p, training).to(x.device))


def test_calc_comp_feat_distill_loss_mirror_vs_reference_on_device(dev, monkeypatch):
    """tests/test_stage2_assembly.py's cases with every tensor on the GPU and the real (HIP) q_sample: the loss assembly of the
    compositional-distillation iteration against the fixture the reference's calc_comp_feat_distill_loss wrote."""
    import test_stage2_assembly as T
    _cpu_draws(monkeypatch)
    # on the device the feature-matching products run on the MFMA kernel (fp16 operands, autograd_ops.MatmulNTFn): fp16-operand tolerance
    T.run_comp_cases(dev, 3e-3, torch_q_sample=False)


def test_calc_normal_recon_loss_mirror_vs_reference_on_device(dev, monkeypatch):
    import test_stage2_assembly as T
    _cpu_draws(monkeypatch)
    T.run_recon_cases(dev, 1e-4, torch_q_sample=False)


def test_comp_losses_on_device_tensors_vs_reference(dev):
    """comp_losses.py (feature matching with flow_model None, recon / suppress losses) on GPU tensors against the reference's values
    and gradients -- the CPU suite runs the same checks on host tensors."""
    import test_comp_losses as TC
    from gen_golden import PRESERVE_CASES
    g = np.load(os.path.join(GOLDEN, "comp_preserve.npz"))
    for tag, kw, scale in PRESERVE_CASES:
        TC.check_preserve_case(g, tag, kw, scale, device=dev, tol=2e-3)      # the matmuls are MFMA GEMMs on fp16 operands here (MatmulNTFn)
    TC.check_recon_and_suppress(g, device=dev, tol=2e-5)


def test_arcface_align_loss_gradient_into_the_latent_vs_oracle_pipeline(dev):
    """``LatentDiffusion.calc_arcface_align_loss`` (ddpm.py:2511-2535) with its gradient, end to end on the device: the x0 prediction is
    decoded by the VAE decoder's autograd node, cropped at the detector's box, resized, greyed, gradient-masked, embedded by ResNetFace-18's
    node and aligned to the reference face.  Checked against the SAME wrapper code on the CPU with the torch-functional oracles standing
    for the two networks (autograd through them = the reference's gradient): the three losses to 2e-2, d(loss)/d(x_recon) to 8e-2
    rel-L2 / cosine > 0.995 (fp16 sign decisions through both networks, see test_hip_face.py)."""
    import torch.nn.functional as F
    from adaface_dev_amd import rng
    from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.arcface_wrapper import ArcFaceWrapper, FaceCropper
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
    from oracle import face_oracle as FO, vae_oracle as VO
    from trainer_util import fixed_face_detector
    cfg = dict(AutoencoderKLDecoder.SD15_DDCONFIG, ch=32, resolution=128)
    ae = AutoencoderKLDecoder(cfg)
    with torch.no_grad():
        for n, p in ae.named_parameters():
            p.copy_(rng.synth_tensor(n, p.shape, seed=90))
    net = resnet_face18().eval()
    fsd = rng.synth_face_state_dict(net.state_dict(), seed=50)
    net.load_state_dict(fsd)
    vsd = {k: v.detach().float() for k, v in ae.state_dict().items()}

    class OracleFace(torch.nn.Module):
        def forward(self, g):
            return FO.resnet_face18(fsd, g.float())

    class OracleVAE(torch.nn.Module):
        def decode(self, z):
            return VO.decode(vsd, z)

    def shell(first_stage, arcface):
        ld = LatentDiffusion.__new__(LatentDiffusion)
        torch.nn.Module.__init__(ld)
        ld.first_stage_model, ld.arcface, ld.scale_factor = first_stage, arcface, 0.18215
        return ld

    x_start = 0.18215 * rng.synth_input("afl.x0", (2, 4, 32, 32), seed=93)
    x_recon = 0.18215 * rng.synth_input("afl.xr", (2, 4, 32, 32), seed=94)
    res = {}
    for name, d, ld in (("cpu", torch.device("cpu"), shell(OracleVAE(), ArcFaceWrapper(OracleFace(), FaceCropper(fixed_face_detector), dtype=torch.float32))),
                        ("hip", dev, shell(ae.to(dev).eval(), ArcFaceWrapper(net.to(dev), FaceCropper(fixed_face_detector))))):
        xr = x_recon.to(d).detach().requires_grad_(True)
        l_align, l_fg, l_bg, boxes, conf, found = ld.calc_arcface_align_loss(x_start.to(d), xr, fg_faces_grad_mask_ratios=(0.9, 0.3))
        assert int(found.sum()) == 2 and boxes is not None
        (l_align + 10.0 * l_fg).backward()
        res[name] = (float(l_align.detach()), float(l_fg.detach()), xr.grad.float().cpu(), boxes.cpu())
    assert torch.equal(res["cpu"][3], res["hip"][3])
    for i in (0, 1):
        assert abs(res["hip"][i] - res["cpu"][i]) < 2e-2 * abs(res["cpu"][i]) + 1e-5, (i, res["hip"][i], res["cpu"][i])
    g, r = res["hip"][2], res["cpu"][2]
    e = rel_l2(g.numpy(), r.numpy())
    cos = float(F.cosine_similarity(g.flatten(), r.flatten(), dim=0))
    print(f"calc_arcface_align_loss: d/d(x_recon) rel-L2 {e:.3e} cosine {cos:.5f} (|g| {float(r.norm()):.3e})")
    assert float(r.norm()) > 0 and e < 8e-2 and cos > 0.995
