#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE modules.

Run in the build container only (it imports /root/reference, which never travels to
the GPU box):

    python tests/golden/gen_golden.py [--skip-full]

What is committed is data: seeded inputs are regenerated from ``adaface_dev_amd.rng`` on
both sides, so the fixtures hold only expected outputs (plus small probes for the
full-size network).  The reference has no tests / fixtures of its own, so
these files are what pins the oracle (oracle/*.py) and, through it, the HIP path.
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = "/root/reference"


def install_reference_stubs():
    """Two import-time stubs the reference needs here: torchvision.utils and
    omegaconf.listconfig.  transformers must be imported BEFORE the fake torchvision exists."""
    import transformers  # noqa: F401
    import transformers.models.clip.modeling_clip  # noqa: F401

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tvu.draw_bounding_boxes = lambda *a, **k: None
        tv.utils = tvu
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        ocl = types.ModuleType("omegaconf.listconfig")

        class ListConfig(list):
            pass

        ocl.ListConfig = ListConfig
        oc.listconfig = ocl
        sys.modules["omegaconf"] = oc
        sys.modules["omegaconf.listconfig"] = ocl
    if REF not in sys.path:
        sys.path.insert(0, REF)


def probes(t: torch.Tensor):
    """Size-independent summary of a big tensor: mean, abs-mean, and 64 strided samples."""
    f = t.detach().float().reshape(-1)
    n = f.numel()
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761) % n
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()]).astype(np.float32)


def gen_unet_tiny(out):
    from adaface_dev_amd import TINY_UNET_CONFIG, rng
    from ldm.modules.diffusionmodules.openaimodel import UNetModel

    torch.manual_seed(0)
    m = UNetModel(**TINY_UNET_CONFIG).eval()
    rng.load_synth_weights(m, seed=1)
    x = rng.synth_input("tiny.x", (2, 4, 16, 16), seed=1)
    ctx = rng.synth_input("tiny.ctx", (2, 77, 64), seed=1)
    t = torch.tensor([10, 500], dtype=torch.int64)

    blocks = {}
    hooks = []
    names = [(f"in{i}", b) for i, b in enumerate(m.input_blocks)] + [("mid", m.middle_block)] + [
        (f"out{i}", b) for i, b in enumerate(m.output_blocks)
    ]
    for n, b in names:
        hooks.append(b.register_forward_hook(lambda mod, a, o, n=n: blocks.__setitem__(n, o.detach().clone())))

    res = {}
    # (a) plain forward + gradients wrt x and context for a fixed cotangent
    xg = x.clone().requires_grad_(True)
    cg = ctx.clone().requires_grad_(True)
    eps = m(xg, t, cg, extra_info={})
    cot = rng.synth_input("tiny.cot", eps.shape, seed=1)
    (eps * cot).sum().backward()
    res["eps"] = eps.detach().numpy()
    res["grad_x"] = xg.grad.numpy()
    res["grad_ctx"] = cg.grad.numpy()
    for n, o in blocks.items():
        res["block_" + n] = o.numpy()
    for h in hooks:
        h.remove()

    # (b) img_mask (self-attention key mask) + capture of cross-attn activations, layers 22-24
    mask = torch.ones(2, 1, 16, 16)
    mask[0, :, :, :5] = 0
    mask[1, :, 9:, :] = 0
    ei = {"img_mask": mask, "capture_ca_activations": True}
    with torch.no_grad():
        eps_m = m(x, t, ctx, extra_info=ei)
    res["eps_masked"] = eps_m.numpy()
    acts = ei["ca_layers_activations"]
    for key in ("outfeat", "attn", "attnscore", "q", "attn_out"):
        for li, v in acts[key].items():
            res[f"cap_{key}_{li}"] = probes(v)
            res[f"cap_{key}_{li}_shape"] = np.asarray(v.shape, dtype=np.int64)
    # one capture kept in full (layer 24, smallest useful set)
    res["cap_attn_24_full"] = acts["attn"][24].numpy().astype(np.float16)
    res["cap_q_24_full"] = acts["q"][24].numpy()
    res["cap_attn_out_24_full"] = acts["attn_out"][24].numpy()
    np.savez_compressed(os.path.join(out, "unet_tiny.npz"), **res)
    print("unet_tiny: eps absmean", float(np.abs(res["eps"]).mean()), "masked", float(np.abs(res["eps_masked"]).mean()))


def gen_unet_full(out):
    from adaface_dev_amd import SD15_UNET_CONFIG, rng
    from ldm.modules.diffusionmodules.openaimodel import UNetModel

    m = UNetModel(**SD15_UNET_CONFIG).eval()
    nparams = sum(p.numel() for p in m.parameters())
    ntens = len(list(m.state_dict().keys()))
    rng.load_synth_weights(m, seed=0)
    x = rng.synth_input("full.x", (1, 4, 64, 64), seed=0)
    ctx = rng.synth_input("full.ctx", (1, 77, 768), seed=0)
    t = torch.tensor([500], dtype=torch.int64)
    res = {"nparams": np.int64(nparams), "ntensors": np.int64(ntens)}
    hooks = []
    names = [(f"in{i}", b) for i, b in enumerate(m.input_blocks)] + [("mid", m.middle_block)] + [
        (f"out{i}", b) for i, b in enumerate(m.output_blocks)
    ]
    for n, b in names:
        hooks.append(b.register_forward_hook(lambda mod, a, o, n=n: res.__setitem__("probe_" + n, probes(o))))
    t0 = time.time()
    with torch.no_grad():
        eps = m(x, t, ctx, extra_info={})
    dt = time.time() - t0
    res["eps"] = eps.numpy()
    res["ref_cpu_seconds"] = np.float64(dt)
    # gradients of <eps, cot> w.r.t. x and context through the REFERENCE module's autograd at full size (base weights frozen)
    for h in hooks:
        h.remove()
    for p in m.parameters():
        p.requires_grad_(False)
    cot = rng.synth_input("full.cot", (1, 4, 64, 64), seed=0)
    xg, cg = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    t0 = time.time()
    (m(xg, t, cg, extra_info={}) * cot).sum().backward()
    res["grad_x"], res["grad_ctx"] = xg.grad.numpy(), cg.grad.numpy()
    print(f"unet_full backward {time.time() - t0:.1f}s, |grad_ctx| {float(cg.grad.abs().mean()):.3e}")
    np.savez_compressed(os.path.join(out, "unet_full.npz"), **res)
    print(f"unet_full: {nparams} params, {ntens} tensors, fwd {dt:.1f}s, eps absmean {float(eps.abs().mean()):.4f}")


def gen_blocks(out):
    """Reference leaf/block modules at reduced widths, incl. the shapes the HIP kernels special-case."""
    from adaface_dev_amd import rng
    from ldm.modules.attention import BasicTransformerBlock, CrossAttention, SpatialTransformer
    from ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
    from ldm.modules.diffusionmodules.util import normalization, timestep_embedding

    res = {}
    # timestep embedding known answers
    t = torch.tensor([0, 1, 21, 500, 981, 999], dtype=torch.int64)
    res["temb_t"] = t.numpy()
    res["temb_320"] = timestep_embedding(t, 320).numpy()

    # GroupNorm32 + SiLU, eps 1e-5
    gn = normalization(64)
    rng.load_synth_weights(gn, seed=2)
    x = rng.synth_input("blk.gn.x", (2, 64, 8, 8), seed=2, scale=2.0) + 0.5
    res["gn_silu"] = torch.nn.functional.silu(gn(x)).detach().numpy()

    # CrossAttention: self (masked / unmasked) and cross, head dim 40 like SD-1.5 at C=320 (scaled: 8 heads x 8)
    for tag, qd, cd, heads, dh, n, l in (("self", 64, None, 8, 8, 64, 64), ("cross", 64, 48, 8, 8, 64, 77),
                                         ("d40", 80, 96, 2, 40, 48, 77)):
        ca = CrossAttention(query_dim=qd, context_dim=cd, heads=heads, dim_head=dh).eval()
        rng.load_synth_weights(ca, seed=3)
        xq = rng.synth_input(f"blk.ca.{tag}.x", (2, n, qd), seed=3)
        cx = None if cd is None else rng.synth_input(f"blk.ca.{tag}.ctx", (2, l, cd), seed=3)
        with torch.no_grad():
            res[f"ca_{tag}"] = ca(xq, cx).numpy()
            if cd is None:
                mask = torch.ones(2, 1, 8, 8)
                mask[0, :, :3] = 0
                mask[1, :, :, 6:] = 0
                res[f"ca_{tag}_masked"] = ca(xq, None, mask=mask).numpy()
                res[f"ca_{tag}_allmasked"] = ca(xq, None, mask=torch.zeros(2, 1, 8, 8)).numpy()

    # ResBlock with and without 1x1 skip
    emb = rng.synth_input("blk.rb.emb", (2, 128), seed=4)
    for tag, cin, cout in (("same", 64, 64), ("proj", 96, 64)):
        rb = ResBlock(cin, 128, 0.0, out_channels=cout).eval()
        rng.load_synth_weights(rb, seed=4)
        x = rng.synth_input(f"blk.rb.{tag}.x", (2, cin, 8, 8), seed=4)
        with torch.no_grad():
            res[f"rb_{tag}"] = rb(x, emb).numpy()

    # Down / Up sample
    dn = Downsample(64, True, out_channels=64).eval()
    up = Upsample(64, True, out_channels=64).eval()
    rng.load_synth_weights(dn, seed=5)
    rng.load_synth_weights(up, seed=5)
    x = rng.synth_input("blk.ud.x", (2, 64, 8, 8), seed=5)
    with torch.no_grad():
        res["down"] = dn(x).numpy()
        res["up"] = up(x).numpy()

    # BasicTransformerBlock / SpatialTransformer
    st = SpatialTransformer(64, 8, 8, depth=1, context_dim=48).eval()
    rng.load_synth_weights(st, seed=6)
    x = rng.synth_input("blk.st.x", (2, 64, 8, 8), seed=6)
    cx = rng.synth_input("blk.st.ctx", (2, 77, 48), seed=6)
    mask = torch.ones(2, 1, 16, 16)
    mask[0, :, :6] = 0
    with torch.no_grad():
        res["st"] = st(x, cx).numpy()
        res["st_masked"] = st(x, cx, mask=mask).numpy()
    np.savez_compressed(os.path.join(out, "blocks.npz"), **res)
    print("blocks:", sorted(res.keys()))


def gen_schedule(out):
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import (make_beta_schedule, make_ddim_sampling_parameters,
                                                   make_ddim_timesteps)

    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)
    ts = make_ddim_timesteps("uniform", 50, 1000, verbose=False)
    ac32 = torch.tensor(ac, dtype=torch.float32)
    sig, a, ap = make_ddim_sampling_parameters(ac32, ts, 0.0, verbose=False)
    np.savez_compressed(
        os.path.join(out, "schedule.npz"),
        betas=betas, alphas_cumprod=ac, ddim_timesteps=ts,
        ddim_alphas=np.asarray(a, dtype=np.float32), ddim_alphas_prev=np.asarray(ap, dtype=np.float32),
        ddim_sigmas=np.asarray(sig, dtype=np.float32),
        # in-code known answers, ldm/models/diffusion/ddim.py:265-270 (fp16-printed)
        kat_alphas_first=np.asarray([0.9985, 0.9805, 0.9609, 0.9399, 0.9170], dtype=np.float32),
        kat_alphas_last=np.asarray([0.0140, 0.0113, 0.0091, 0.0073, 0.0058], dtype=np.float32),
    )

    # One reference DDIM trajectory with a stand-in epsilon model (tests the sampler arithmetic,
    # CFG ordering (cond, uncond) and guidance annealing -- ddim.py:133-302).
    class FakeLDM:
        num_timesteps = 1000
        device = torch.device("cpu")

        def __init__(self):
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = ac32
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)

        def apply_model(self, x, t, c):
            ctx = c[0] if isinstance(c, tuple) else c
            return torch.tanh(x) * 0.7 + 0.05 * ctx.mean(dim=(1, 2)).reshape(-1, 1, 1, 1) + 1e-4 * t.reshape(-1, 1, 1, 1).float()

    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    sampler = DDIMSampler(FakeLDM())
    from adaface_dev_amd import rng
    xT = rng.synth_input("ddim.xT", (2, 4, 8, 8), seed=7)
    c = rng.synth_input("ddim.c", (2, 77, 16), seed=7)
    uc = rng.synth_input("ddim.uc", (2, 77, 16), seed=7)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        x0, inter = sampler.sample(S=50, batch_size=2, shape=(4, 8, 8), conditioning=(c, ["a", "b"], {}),
                                   verbose=False, x_T=xT, guidance_scale=(4.0, 1.0),
                                   unconditional_conditioning=(uc, ["", ""], {}), log_every_t=10)
    np.savez_compressed(os.path.join(out, "ddim_step.npz"), x_final=x0.numpy(),
                        x_inter=np.stack([t.numpy() for t in inter["x_inter"]]))
    print("schedule: ddim_timesteps[:3]", ts[:3], "alphas[:3]", np.asarray(a)[:3])


def gen_train(out):
    """Training-step pieces that ARE importable from the reference: calc_recon_loss (ldm/util.py:1678),
    CAdamW (ldm/c_adamw.py) and the LR schedule (ldm/modules/lr_scheduler.py)."""
    import torch.nn.functional as F
    from adaface_dev_amd import rng
    from ldm.c_adamw import AdamW as CAdamW   # the reference names the class AdamW (c_adamw.py:13)
    from ldm.modules.lr_scheduler import LambdaWarmUpCosineScheduler
    from ldm.util import calc_recon_loss

    res = {}
    pred = rng.synth_input("train.pred", (3, 4, 16, 16), seed=8)
    gt = rng.synth_input("train.gt", (3, 4, 16, 16), seed=8)
    fg = (rng.synth_input("train.fg", (3, 1, 16, 16), seed=8) > 0.2).float()
    im = torch.ones(3, 1, 16, 16)
    im[1, :, :, :4] = 0
    res["recon_plain"] = calc_recon_loss(F.mse_loss, pred, gt, None, None)[0].numpy()
    res["recon_fg0"] = calc_recon_loss(F.mse_loss, pred, gt, im, fg, fg_pixel_weight=1, bg_pixel_weight=0)[0].numpy()
    res["recon_fg_half"] = calc_recon_loss(F.mse_loss, pred, gt, im, fg, fg_pixel_weight=1, bg_pixel_weight=0.5)[0].numpy()
    res["recon_inst"] = calc_recon_loss(F.mse_loss, pred, gt, im, fg, instance_weights=torch.tensor([1.0, 0.0, 2.0]),
                                        fg_pixel_weight=1, bg_pixel_weight=0.1)[0].numpy()

    # CAdamW trace: two tensors, weight decay on, 4 steps (betas of the yaml: 0.9, 0.995; eps 1e-6)
    ps = [torch.nn.Parameter(rng.synth_input("train.p0", (37, 5), seed=8)), torch.nn.Parameter(rng.synth_input("train.p1", (130,), seed=8))]
    opt = CAdamW([{"params": [ps[0]], "weight_decay": 0.02}, {"params": [ps[1]], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.995), eps=1e-6)
    for step in range(4):
        for i, p in enumerate(ps):
            p.grad = rng.synth_input(f"train.g{i}.{step}", p.shape, seed=8)
        opt.step()
        res[f"cadamw_p0_step{step}"] = ps[0].detach().numpy().copy()
        res[f"cadamw_p1_step{step}"] = ps[1].detach().numpy().copy()

    sch = LambdaWarmUpCosineScheduler(warm_up_steps=500, lr_min=0.1, lr_max=1.0, lr_start=0.01, max_decay_steps=60000)
    ns = np.asarray([0, 1, 250, 499, 500, 501, 1000, 30000, 59999, 60000, 90000], dtype=np.int64)
    res["lr_n"] = ns
    res["lr_mult"] = np.asarray([sch(int(n)) for n in ns], dtype=np.float64)
    np.savez_compressed(os.path.join(out, "train.npz"), **res)
    print("train:", {k: (float(v) if v.ndim == 0 else v.shape) for k, v in res.items() if k.startswith("recon")})


CLIP_SMALL = dict(hidden=128, heads=2, layers=3, inter=512, vocab=1000, max_pos=77)


def gen_clip(out):
    """(a) the reference's own CLIPAttentionMKV (m = 1, 2); (b) transformers' CLIPTextModel (third-party layer arithmetic,
    version recorded) on a reduced config.  Extra import-time stubs: diffusers / ConsistentID names that
    adaface/arc2face_models.py and adaface/util.py import at module top but the attention class never touches."""
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    from adaface_dev_amd import rng

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(name, m)

    D = lambda n: type(n, (), {})
    stub("diffusers", StableDiffusionPipeline=D("a"), UNet2DConditionModel=D("b"), DDIMScheduler=D("c"))
    stub("diffusers.models")
    stub("diffusers.models.unets")
    stub("diffusers.models.unets.unet_2d_condition", UNet2DConditionOutput=D("d"))
    stub("ConsistentID")
    stub("ConsistentID.lib")
    stub("ConsistentID.lib.pipeline_ConsistentID", ConsistentIDPipeline=D("e"))
    import warnings
    warnings.simplefilter("ignore")
    from adaface.arc2face_models import CLIPAttentionMKV
    from transformers.modeling_attn_mask_utils import AttentionMaskConverter

    c = CLIP_SMALL
    cfg = CLIPTextConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"],
                         num_attention_heads=c["heads"], max_position_embeddings=c["max_pos"], hidden_act="quick_gelu")
    res = {"transformers_version": np.asarray(transformers.__version__)}
    for m in (1, 2):
        att = CLIPAttentionMKV(cfg, multiplier=m).eval()
        with torch.no_grad():
            for n, p in att.named_parameters():
                p.copy_(rng.synth_tensor(f"mkv{m}." + n, p.shape, seed=20))
        for T in (22, 77):
            h = rng.synth_input(f"clip.h{T}", (2, T, c["hidden"]), seed=20)
            cm = AttentionMaskConverter._make_causal_mask((2, T), torch.float32, device=h.device)
            with torch.no_grad():
                res[f"mkv_m{m}_T{T}"] = att(h, None, cm)[0].numpy()
    model = CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            key = n if n.startswith("text_model.") else "text_model." + n      # transformers-4 key layout
            p.copy_(rng.synth_tensor(key, p.shape, seed=21))
    ids = torch.randint(0, c["vocab"], (2, 77), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        o = model(input_ids=ids, output_hidden_states=True)
    res["text_ids"] = ids.numpy()
    res["text_last"] = o.last_hidden_state.numpy()
    res["text_hidden_m3"] = o.hidden_states[-3].numpy()
    res["text_hidden_0"] = o.hidden_states[0].numpy()
    np.savez_compressed(os.path.join(out, "clip.npz"), **res)
    print("clip:", {k: v.shape for k, v in res.items()}, len(o.hidden_states))


def gen_arcface(out):
    """ResNetFace-18 IR-SE (evaluation/arcface_resnet.py:337-339) in eval mode on seeded weights AND BatchNorm statistics:
    the 512-d embedding and the four stage outputs' probes for 3 grey 128x128 crops."""
    from evaluation.arcface_resnet import resnet_face18
    from adaface_dev_amd import rng
    m = resnet_face18(use_se=True).eval()
    m.load_state_dict(rng.synth_face_state_dict(m.state_dict(), seed=50))
    x = rng.synth_input("face.x", (3, 1, 128, 128), seed=50)
    feats = []
    hooks = [getattr(m, f"layer{i}").register_forward_hook(lambda mod, inp, o: feats.append(o.detach())) for i in range(1, 5)]
    with torch.no_grad():
        y = m(x)
    for h in hooks:
        h.remove()
    d = {"emb": y.numpy()}
    for i, f in enumerate(feats):
        d[f"layer{i + 1}_probes"] = probes(f)
    d["layer4"] = feats[3].numpy()
    np.savez_compressed(os.path.join(out, "arcface.npz"), **d)
    print("arcface.npz", y.shape, float(y.abs().mean()))


VAE_SMALL = dict(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=128, z_channels=4)


E_PD = 32


def gen_embedding_manager(out):
    """The REFERENCE EmbeddingManager (ldm/modules/embedding_manager.py) and its ldm.util helpers on the prompt cases of
    tests/em_fixture_util.py.  The manager imports ``adaface.face_id_to_ada_prompt`` (diffusers / insightface: absent), of which it
    only uses the encoder factory; the factory is replaced by one returning the fixture's fake encoder -- the code under test
    (token search, slot placement, class-string scan, masks, RNG consumption of the training perturbation) is the reference's."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import em_fixture_util as U
    from adaface_dev_amd.adaface.adaface_wrapper import WordTokenizer
    install_reference_stubs()
    fake_mod = types.ModuleType("adaface.face_id_to_ada_prompt")
    holder = {}
    fake_mod.create_id2ada_prompt_encoder = lambda *a, **k: holder["enc"]
    pkg = types.ModuleType("adaface")
    pkg.__path__ = []
    sys.modules.setdefault("adaface", pkg)
    sys.modules["adaface.face_id_to_ada_prompt"] = fake_mod
    import contextlib
    import io
    from ldm.modules.embedding_manager import EmbeddingManager
    import ldm.util as RU
    d = {}
    table = U.token_table()
    for name, (iter_type, subj_names, prompts, id_bs, K, real_bs, training) in U.CASES.items():
        tok = WordTokenizer()
        holder["enc"] = U.FakeID2AdaPromptEncoder(num_id_vecs=K)
        with contextlib.redirect_stdout(io.StringIO()):
            em = EmbeddingManager(U.text_embedder(tok, table), ["z"], subj_name_to_cls_delta_string={"alice": "young woman", "bob": "man"},
                                  out_emb_dim=U.E, cls_delta_string="person", adaface_encoder_types=["arc2face"],
                                  training_perturb_std_range=(0.05, 0.1) if name == "perturbed" else None,
                                  training_perturb_prob={"unet_distill_iter": 0.6} if name == "perturbed" else None)
            em.train(training)
            em.set_curr_batch_subject_names(subj_names)
            em.set_image_prompts_and_iter_type(None if id_bs is None else U.id_embs(id_bs, K, 5), None, iter_type, real_bs)
        ids = tok(prompts, max_length=77)["input_ids"]
        torch.manual_seed(123)
        patched = em(ids, table[ids])
        d[name + ".ids"] = ids.numpy()
        d[name + ".patched"] = patched.numpy()
        d[name + ".emb_mask"] = em.prompt_emb_mask.numpy()
        d[name + ".pad_mask"] = em.prompt_pad_mask.numpy()
        d[name + ".rng_after"] = torch.rand(4).numpy()                       # the perturbation consumed the global RNG identically
        for k, v in U.flatten_indices(em.placeholder2indices).items():
            d[f"{name}.p2i.{k}"] = np.zeros((0,)) if v is None else v.numpy()
        d[name + ".p2i_keys"] = np.array(sorted(em.placeholder2indices.keys()), dtype="U8")
        cls = em.cls_delta_string_indices
        d[name + ".cls"] = np.array([[b, s, m] for b, s, m, _ in cls], dtype=np.int64).reshape(-1, 3)
        d[name + ".cls_names"] = np.array([n for *_, n in cls], dtype="U16")
        d[name + ".merged"] = RU.merge_cls_token_embeddings(patched, cls).numpy()
        d[name + ".span"] = np.array(em.CLS_DELTA_STRING_MAX_SEARCH_SPAN)
        d[name + ".cls_w"] = em.cls_delta_token_weights.numpy()
        print("embedding_manager", name, ids.shape, "cls", cls, "p2i", {k: (None if v is None else v[0].shape) for k, v in em.placeholder2indices.items()})
    # helpers on inputs the manager cases do not reach
    g = torch.Generator().manual_seed(3)
    b = torch.randint(0, 6, (40,), generator=g)
    n = torch.randint(0, 77, (40,), generator=g)
    fb, fn = RU.extract_first_index_in_each_instance((b, n))
    d["first.b"], d["first.n"], d["first.fb"], d["first.fn"] = b.numpy(), n.numpy(), fb.numpy(), fn.numpy()
    emb = torch.randn(3, 20, 8, generator=g)
    multi = [(1, 9, 3, "x"), (1, 3, 2, "y"), (2, 5, 1, "x"), (0, 2, 4, "y")]
    d["merge.in"], d["merge.out"] = emb.numpy(), RU.merge_cls_token_embeddings(emb, multi).numpy()
    d["merge.idx"] = np.array([m[:3] for m in multi], dtype=np.int64)
    # prompt-delta regularisation (ldm/util.py:296-470, 1426-1480): values and gradients through the reference functions
    pe = torch.randn(4, 77, E_PD, generator=g).requires_grad_(True)
    ids4 = torch.full((4, 77), 49407)
    ids4[:, 0] = 49406
    for bi, nreal in enumerate((9, 14, 9, 14)):
        ids4[bi, 1:1 + nreal] = 1000 + torch.arange(nreal)
    mask4 = ((ids4 != 49406) & (ids4 != 49407)).unsqueeze(2)
    loss = RU.calc_prompt_emb_delta_loss(pe, mask4.clone())
    loss.backward()
    d["pd.emb"], d["pd.mask"], d["pd.loss"], d["pd.grad"] = pe.detach().numpy(), mask4.numpy(), loss.detach().numpy(), pe.grad.numpy()
    a3, b3 = torch.randn(2, 5, 6, 8, generator=g), torch.randn(2, 1, 6, 8, generator=g)
    o2, w2 = RU.ortho_subtract(a3, b3, b_discount=0.7, on_last_n_dims=2, return_align_coeffs=True)
    d["os.a"], d["os.b"], d["os.out"], d["os.w"] = a3.numpy(), b3.numpy(), o2.numpy(), w2.numpy()
    dl, rf = torch.randn(3, 4, 10, 16, generator=g), torch.randn(3, 4, 10, 16, generator=g)
    d["rc.delta"], d["rc.ref"] = dl.numpy(), rf.numpy()
    d["rc.none"] = RU.calc_ref_cosine_loss(dl, rf, None, exponent=3, do_demeans=[True, False], first_n_dims_into_instances=3,
                                           ref_grad_scale=0, aim_to_align=False, reduction="none").numpy()
    np.savez_compressed(os.path.join(out, "embedding_manager.npz"), **d)


def vae_test_masks(r):
    """Deterministic fg / aug masks for the masked-encoder vectors (also used by the tests)."""
    import torch
    fg = torch.zeros(2, 1, r, r)
    fg[0, :, r // 4: 3 * r // 4, r // 4: 3 * r // 4] = 1
    fg[1, :, :, : r // 2] = 1
    aug = torch.ones(2, 1, r, r)
    aug[1, :, : r // 8] = 0
    return fg, aug


def gen_vae(out):
    """The REFERENCE's VAE Decoder (ldm/modules/diffusionmodules/model.py:502-608) at reduced width (ch 32 -> 128 channels at the
    latent level, 16x16 latent -> 128x128 image, 256-token mid attention): full output + the mid / first-up-level features, and the
    full-size SD-1.5 decoder (ch 128, 64x64 latent -> 512x512) as probes + a 3x64x64 crop."""
    from ldm.modules.diffusionmodules.model import Decoder
    from adaface_dev_amd import rng
    d = {}
    for tag, cfg, zhw in (("small", VAE_SMALL, 16), ("full", dict(VAE_SMALL, ch=128, resolution=256), 64)):
        m = Decoder(**cfg).eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(rng.synth_tensor("decoder." + n, p.shape, seed=90))
        z = rng.synth_input(f"vae.z.{tag}", (2 if tag == "small" else 1, 4, zhw, zhw), seed=90)
        feats = {}
        hooks = [m.mid.attn_1.register_forward_hook(lambda mod, i, o: feats.__setitem__("attn", o.detach())),
                 m.up[3].register_forward_hook(lambda mod, i, o: None)]
        with torch.no_grad():
            y = m(z)
        for h in hooks:
            h.remove()
        if tag == "small":
            d["small_out"] = y.numpy()
            d["small_attn"] = feats["attn"].numpy()
        else:
            d["full_probes"] = probes(y)
            d["full_crop"] = y[0, :, 200:264, 300:364].numpy()
            d["full_attn_probes"] = probes(feats["attn"])
        print("vae", tag, tuple(y.shape), float(y.abs().mean()))
    # Encoder (model.py:408-500), reduced width: 2 x 3 x 128 x 128 image -> moments [2, 8, 16, 16]; asymmetric-padding Downsample
    from ldm.modules.diffusionmodules.model import Encoder
    e = Encoder(**dict(VAE_SMALL, double_z=True)).eval()
    with torch.no_grad():
        for n, p in e.named_parameters():
            p.copy_(rng.synth_tensor("encoder." + n, p.shape, seed=90))
        img = rng.synth_input("vae.img.small", (2, 3, 128, 128), seed=90)
        d["enc_small_out"] = e(img).numpy()
        # masked mid-block attention (model.py:191-232): fg / aug masks built by tests/trainer_util-free arithmetic below
        fg, aug = vae_test_masks(128)
        d["enc_small_masked_out"] = e(img, {"fg_mask": fg, "aug_mask": aug}).numpy()
        d["enc_small_fgonly_out"] = e(img, {"fg_mask": fg, "aug_mask": None}).numpy()
    print("vae encoder", d["enc_small_out"].shape, float(np.abs(d["enc_small_out"]).mean()),
          "masked delta", float(np.abs(d["enc_small_masked_out"] - d["enc_small_out"]).mean()))
    np.savez_compressed(os.path.join(out, "vae.npz"), **d)



# ----------------------------------------------------------------------------- host orchestration, driven through the REAL reference
def _ref_ddpm_shell(**attrs):
    """A reference ``LatentDiffusion`` object WITHOUT running its constructor (which needs Lightning, diffusers pipelines and model
    files): nn.Module state + the real ``register_schedule`` tables (SD-1.5 linear schedule, v1-finetune yaml) + whatever attributes
    the driven method reads.  Every method executed on it is the reference's own code."""
    import ldm.models.diffusion.ddpm as ref_ddpm
    obj = ref_ddpm.LatentDiffusion.__new__(ref_ddpm.LatentDiffusion)
    torch.nn.Module.__init__(obj)
    obj.parameterization, obj.v_posterior = "eps", 0.0
    obj.register_schedule(beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.012)
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


class _AsDiffusersUNet(torch.nn.Module):
    """Call protocol of the teacher's U-Net (unet_teachers.py:137-138): unet(sample=, timestep=, encoder_hidden_states=, return_dict=False)[0]."""

    def __init__(self, eps_model):
        super().__init__()
        self.eps_model = eps_model

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=False):
        return (self.eps_model(sample, timestep, encoder_hidden_states),)


TEACHER_CASES = (
    dict(name="nocfg_1step_doubled_ctx", steps=1, force=False, p=0.0, ctx="doubled", neg=False, same=False),
    dict(name="nocfg_4step", steps=4, force=False, p=0.0, ctx="pos", neg=False, same=False),
    dict(name="cfg_3step_doubled_ctx", steps=3, force=True, p=0.0, ctx="doubled", neg=False, same=False),
    dict(name="cfg_3step_separate_neg", steps=3, force=True, p=0.0, ctx="pos", neg=True, same=False),
    dict(name="cfg_4step_same_t_noise", steps=4, force=True, p=0.0, ctx="doubled", neg=False, same=True),
    dict(name="pcfg_2step_coin", steps=2, force=False, p=0.6, ctx="doubled", neg=False, same=False),
)


def gen_teacher(out):
    """REFERENCE ``UNetTeacher.forward`` (adaface/unet_teachers.py:64-187) around the stand-in eps-model (tests/standin.py), with the
    reference's own ``q_sample`` / ``predict_start_from_noise``: with and without classifier-free guidance (doubled context, separate
    negative context, the p_uses_cfg coin), same_t_noise_across_instances, 1 - 4 denoising steps, seeded torch / numpy RNG.  Stored:
    every eps, x0, noise and timestep, the drawn cfg scale, and the RNG draws (relative_ts) recovered by replaying the seed."""
    import contextlib
    import io
    import adaface.unet_teachers as ref_teachers
    from adaface_dev_amd import rng
    from standin import StandInEps
    B, h, T, D = 3, 8, 6, 16
    eps_model = StandInEps(D, seed=61)
    ddpm = _ref_ddpm_shell()
    res = {}
    for c in TEACHER_CASES:
        x0 = rng.synth_input("teacher.x0", (B, 4, h, h), seed=62)
        noise = rng.synth_input("teacher.noise", (B, 4, h, h), seed=62)
        pos = rng.synth_input("teacher.pos", (B, T, D), seed=62)
        neg = rng.synth_input("teacher.neg", (B, T, D), seed=62)
        t = torch.tensor([880, 745, 801])
        with contextlib.redirect_stdout(io.StringIO()):
            teacher = ref_teachers.UNetTeacher(p_uses_cfg=c["p"], cfg_scale_range=[1.3, 2])
        teacher.name, teacher.unet = "standin", _AsDiffusersUNet(eps_model)
        ctx = torch.cat([pos, neg]) if c["ctx"] == "doubled" else pos
        torch.manual_seed(1234)
        np.random.seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            preds, xs, ns, ts = teacher(ddpm, x0, noise, t, ctx, negative_context=neg if c["neg"] else None, num_denoising_steps=c["steps"],
                                        force_uses_cfg=c["force"], same_t_noise_across_instances=c["same"])
        k = c["name"]
        res[f"{k}.uses_cfg"] = np.asarray(bool(teacher.uses_cfg))
        res[f"{k}.cfg_scale"] = np.asarray(float(teacher.cfg_scale))
        for i in range(c["steps"]):
            res[f"{k}.eps{i}"] = preds[i].numpy()
            res[f"{k}.x{i + 1}"] = xs[i + 1].numpy()
            res[f"{k}.t{i}"] = ts[i].numpy()
            res[f"{k}.noise{i}"] = ns[i].numpy()
        # the draws the reference made from the global torch RNG, in its order: rand_like(t.float()) then randn_like(pred_x0) per extra step
        torch.manual_seed(1234)
        for i in range(c["steps"] - 1):
            res[f"{k}.rel{i}"] = torch.rand(B).numpy()
            res[f"{k}.drawn_noise{i}"] = torch.randn(B, 4, h, h).numpy()
    np.savez_compressed(os.path.join(out, "teacher.npz"), **res)
    print("teacher:", {c["name"]: (bool(res[c["name"] + ".uses_cfg"]), round(float(res[c["name"] + ".cfg_scale"]), 4)) for c in TEACHER_CASES})


def gen_sdpa(out):
    """REFERENCE ``scaled_dot_product_attention`` and ``ScaleGrad`` (adaface/diffusers_attn_lora_capture.py:23-42, 79-139): plain,
    boolean / additive mask, ``mix_attn_mats_in_batch`` (SC scores averaged with detached MC scores), ``normalize_cross_attn``
    (subject-token scores centred over the pixels and scaled by the learnable factor through the x10 gradient scaler); outputs, scores,
    probabilities and the gradients of sum(out * g) w.r.t. q, k, v and the scale factor."""
    import adaface.diffusers_attn_lora_capture as ref_cap
    from adaface_dev_amd import rng
    B, H, L, S, d = 4, 2, 12, 7, 8
    res = {}

    def run(tag, **kw):
        q = rng.synth_input("sdpa.q", (B, H, L, d), seed=63).requires_grad_(True)
        k = rng.synth_input("sdpa.k", (B, H, S, d), seed=63).requires_grad_(True)
        v = rng.synth_input("sdpa.v", (B, H, S, d), seed=63).requires_grad_(True)
        sf = torch.tensor(0.8, requires_grad=True)
        g = rng.synth_input("sdpa.g", (B, H, L, d), seed=63)
        o, score, w = ref_cap.scaled_dot_product_attention(q, k, v, sf, **kw)
        (o * g).sum().backward()
        res[f"{tag}.out"], res[f"{tag}.score"], res[f"{tag}.prob"] = o.detach().numpy(), score.detach().numpy(), w.detach().numpy()
        res[f"{tag}.dq"], res[f"{tag}.dk"], res[f"{tag}.dv"] = q.grad.numpy(), k.grad.numpy(), v.grad.numpy()
        res[f"{tag}.dscale"] = np.asarray(0.0 if sf.grad is None else float(sf.grad))

    run("plain")
    keep = rng.synth_input("sdpa.mask", (B, 1, L, S), seed=63) > -0.6
    keep[..., 0] = True
    run("boolmask", attn_mask=keep)
    run("addmask", attn_mask=rng.synth_input("sdpa.bias", (B, 1, L, S), seed=63))
    run("mix", mix_attn_mats_in_batch=True)
    subj_b = torch.tensor([0, 0, 0, 1, 1, 1, 3, 3, 3])
    subj_n = torch.tensor([2, 3, 4, 2, 3, 4, 1, 2, 3])
    run("normalize", subj_indices=(subj_b, subj_n), normalize_cross_attn=True)
    run("scale", scale=0.2)
    res["normalize.subj_b"], res["normalize.subj_n"] = subj_b.numpy(), subj_n.numpy()

    # ScaleGrad / gen_gradient_scaler: identity forward, gradient x alpha (alpha 0 -> detach, alpha 1 -> Identity)
    x = rng.synth_input("sg.x", (5, 3), seed=63).requires_grad_(True)
    y = ref_cap.ScaleGrad.apply(x, torch.tensor(0.5), torch.tensor(False))
    (y * y).sum().backward()
    res["scalegrad.y"], res["scalegrad.dx"] = y.detach().numpy(), x.grad.numpy()
    for alpha in (10, 1, 0):
        x = rng.synth_input("sg.x", (5, 3), seed=63).requires_grad_(True)
        y = ref_cap.gen_gradient_scaler(alpha)(x)
        z = (y * y).sum() + x.sum()
        z.backward()
        res[f"gradscaler{alpha}.dx"] = x.grad.numpy()
    np.savez_compressed(os.path.join(out, "sdpa.npz"), **res)
    print("sdpa:", sorted(k for k in res if k.endswith(".dscale")), float(res["normalize.dscale"]))


GUIDED_CASES = (
    dict(name="all_cfg3_recon", mode="all", cfg=3.0, recon=True, capture=False, uncond="given"),
    dict(name="none_cfg1", mode="none", cfg=-1, recon=False, capture=True, uncond=None),
    dict(name="all_cfg2_default_uncond_mask", mode="all", cfg=2.0, recon=True, capture=True, uncond=None, mask=True, gradscale=0.5),
    dict(name="compos_mix", mode="subject-compos", cfg=-1, recon=True, capture=True, uncond=None, mix=True, norm=True, attn_lora=True, ffn=True),
    dict(name="compos_nomix", mode="subject-compos", cfg=-1, recon=False, capture=True, uncond=None, mix=False, norm=True, attn_lora=True, ffn=True),
)


def gen_guided_denoise(out):
    """REFERENCE ``LatentDiffusion.guided_denoise`` / ``apply_model`` / ``sliced_apply_model`` (ddpm.py:1560-1750) on a constructor-free
    shell, with the stand-in wrapper as ``self.model``: gradient modes all / none / subject-compos (the four single-instance passes or
    the joint SC+MC pass with mixed attention), classifier-free guidance with a given or the default unconditional context, x0
    reconstruction, activation capture and collation; eps, x_recon, captured activations, d(sum eps)/d(prompt_emb), and the per-call
    (batch size, flags) log of the wrapper."""
    import json
    from adaface_dev_amd import rng
    from standin import StandInEps, StandInWrapper
    B, h, T, D = 4, 8, 6, 16
    res = {}
    for c in GUIDED_CASES:
        wrapper = StandInWrapper(StandInEps(D, seed=61))
        un = rng.synth_input("gd.uncond_default", (1, T, D), seed=64)
        ld = _ref_ddpm_shell(model=wrapper, uncond_context=(un, [""], {}))
        x0 = rng.synth_input("gd.x0", (B, 4, h, h), seed=64)
        noise = rng.synth_input("gd.noise", (B, 4, h, h), seed=64)
        emb = rng.synth_input("gd.emb", (B, T, D), seed=64).requires_grad_(True)
        t = torch.tensor([500, 20, 981, 333])
        mask = (rng.synth_input("gd.mask", (B, 1, h, h), seed=64) > 0).float() if c.get("mask") else None
        uncond = rng.synth_input("gd.uncond", (B, T, D), seed=64) if c["uncond"] == "given" else None
        cond = (emb, [f"p{i}" for i in range(B)], {})
        torch.manual_seed(4321)            # subject-compos draws torch.rand(1) < 0.5 for the FFN LoRA
        eps, recon, acts = ld.guided_denoise(x0, noise, t, cond, uncond_emb=uncond, img_mask=mask, subj_indices=None,
                                             normalize_cross_attn=c.get("norm", False), mix_sc_mc_attn=c.get("mix", False),
                                             batch_part_has_grad=c["mode"], do_pixel_recon=c["recon"], cfg_scale=c["cfg"],
                                             capture_ca_activations=c["capture"], res_hidden_states_gradscale=c.get("gradscale", 1),
                                             use_attn_lora=c.get("attn_lora", False), use_ffn_lora=c.get("ffn", False),
                                             ffn_lora_adapter_name="comp_distill" if c.get("ffn") else None)
        k = c["name"]
        res[f"{k}.eps"] = eps.detach().numpy()
        res[f"{k}.requires_grad"] = np.asarray(bool(eps.requires_grad))
        if eps.requires_grad:
            eps.sum().backward()
            res[f"{k}.demb"] = emb.grad.numpy()
        if recon is not None:
            res[f"{k}.recon"] = recon.detach().numpy()
        if acts is not None:
            res[f"{k}.act_attn"] = acts["attn"].detach().numpy()
            res[f"{k}.act_attn_requires_grad"] = np.asarray(bool(acts["attn"].requires_grad))
            for li, v in acts["outfeat"].items():
                res[f"{k}.act_outfeat{li}"] = v.detach().numpy()
            res[f"{k}.act_names"] = np.asarray(json.dumps(acts["names"]))
        res[f"{k}.calls"] = np.asarray(json.dumps([[n, fl, ad, npr, ge] for n, fl, ad, npr, ge in wrapper.calls]))
    np.savez_compressed(os.path.join(out, "guided_denoise.npz"), **res)
    print("guided_denoise:", {c["name"]: res[c["name"] + ".eps"].shape for c in GUIDED_CASES})


def gen_distill_loss(out):
    """REFERENCE ``LatentDiffusion.calc_unet_distill_loss`` (ddpm.py:2984-3184; the on-image branch Stage 1 runs) with the REFERENCE
    ``prepare_unet_teacher_context`` (:2885-2980), ``UNetTeacher.forward``, ``guided_denoise`` and ``calc_recon_loss``, stand-in eps-models
    for teacher and student, logging / VAE decode no-ops: the loss, d loss / d prompt_emb, the timesteps drawn inside, for 1 and 3
    denoising steps, without and with teacher CFG (p_unet_teacher_uses_cfg 1: doubled [positive; negative] teacher context and the
    student's guided_denoise at the teacher's cfg scale)."""
    import contextlib
    import io
    import adaface.unet_teachers as ref_teachers
    from adaface_dev_amd import rng
    from standin import StandInEps, StandInWrapper
    B, h, T, D = 2, 8, 24, 16
    res = {}
    for steps, pcfg in ((1, 0.0), (3, 0.0), (2, 1.0)):
        student = StandInWrapper(StandInEps(D, seed=61))
        with contextlib.redirect_stdout(io.StringIO()):
            teacher = ref_teachers.UNetTeacher(p_uses_cfg=pcfg, cfg_scale_range=[1.3, 2])
        teacher.name, teacher.unet = "arc2face", _AsDiffusersUNet(StandInEps(D, seed=65))
        un = rng.synth_input("dl.uncond", (1, T, D), seed=66)
        id2img = rng.synth_input("dl.id2img", (B, 16, D), seed=66)
        flags = {"id2img_prompt_embs": id2img, "id2img_neg_prompt_embs": None, "encoders_num_id_vecs": None, "unet_distill_uses_comp_prompt": False}
        ld = _ref_ddpm_shell(model=student, uncond_context=(un, [""], {}), unet_teacher=teacher, iter_flags=flags,
                             img_prompt_prefix_embs=rng.synth_input("dl.prefix", (1, 4, D), seed=66), unet_teacher_types=["arc2face"],
                             p_unet_teacher_uses_cfg=pcfg, res_hidden_states_gradscale=0.5, unet_distill_on_noise_iters_count=0,
                             trainer=types.SimpleNamespace(global_rank=0))
        ld.decode_first_stage = lambda z: torch.zeros(z.shape[0], 3, 8, 8)
        ld.cache_and_log_generations = lambda *a, **k: None
        x0 = rng.synth_input("dl.x0", (B, 4, h, h), seed=66)
        noise = rng.synth_input("dl.noise", (B, 4, h, h), seed=66)
        emb = rng.synth_input("dl.emb", (B, T, D), seed=66).requires_grad_(True)
        fg = (rng.synth_input("dl.fg", (B, 1, h, h), seed=66) > -0.3).float()
        mon = {}
        torch.manual_seed(2468)
        np.random.seed(99)
        with contextlib.redirect_stdout(io.StringIO()):
            loss = ld.calc_unet_distill_loss(mon, "train", x0, noise, (emb, ["p"] * B, {}), None, fg, steps, False)
        loss.backward()
        k = f"steps{steps}_pcfg{int(pcfg)}"
        res[f"{k}.loss"], res[f"{k}.demb"] = loss.detach().numpy(), emb.grad.numpy()
        res[f"{k}.cfg_scale"] = np.asarray(float(teacher.cfg_scale))
        res[f"{k}.mon"] = np.asarray(float(mon["train/unet_distill_on_image"]))
        # replay of the global torch RNG in the reference's order: t ~ randint(700, 900), then per extra teacher step rand(B), randn
        torch.manual_seed(2468)
        res[f"{k}.t"] = torch.randint(700, 900, (B,)).numpy()
        for i in range(steps - 1):
            res[f"{k}.rel{i}"] = torch.rand(B).numpy()
            res[f"{k}.drawn_noise{i}"] = torch.randn(B, 4, h, h).numpy()
    np.savez_compressed(os.path.join(out, "distill_loss.npz"), **res)
    print("distill_loss:", {k: float(v) for k, v in res.items() if k.endswith(".loss")})



def comp_loss_inputs(device="cpu"):
    """Seeded stand-ins for captured activations of a four-block Stage-2 batch (BLOCK_SIZE 1: [SS, SC, SC-rep, MC]), layers 22-24:
    attention probabilities [4, 8, 64, 20] (rows softmaxed), k / v [4, 32, 20], out features [4, 32, 8, 8]; 4 subject tokens."""
    from adaface_dev_amd import rng
    B4, H, N, L, C = 4, 8, 64, 20, 32
    acts = {"attn": {}, "k": {}, "v": {}, "outfeat": {}}
    for li in (22, 23, 24):
        acts["attn"][li] = torch.softmax(rng.synth_input(f"cl.attn{li}", (B4, H, N, L), seed=71) * 1.5, dim=-1).to(device).requires_grad_(True)
        acts["k"][li] = rng.synth_input(f"cl.k{li}", (B4, C, L), seed=71).to(device).requires_grad_(True)
        acts["v"][li] = rng.synth_input(f"cl.v{li}", (B4, C, L), seed=71).to(device).requires_grad_(True)
        acts["outfeat"][li] = rng.synth_input(f"cl.of{li}", (B4, C, 8, 8), seed=71).to(device)
    future = {"attn": {li: torch.softmax(rng.synth_input(f"cl.fattn{li}", (B4, H, N, L), seed=71), dim=-1).to(device) for li in (22, 23, 24)}}
    subj_1b = (torch.zeros(4, dtype=torch.long, device=device), torch.tensor([4, 5, 6, 7], device=device))
    subj_2b = (torch.tensor([0, 0, 0, 0, 1, 1, 1, 1], device=device), torch.tensor([4, 5, 6, 7, 4, 5, 6, 7], device=device))
    emb_mask = torch.zeros(B4, L, 1, device=device)
    emb_mask[:, 1:12] = 1
    emb_mask[1, 12:15] = 1
    pad_mask = torch.zeros(B4, L, 1, device=device)
    pad_mask[:, 16:] = 1
    fg_mask = torch.zeros(B4, 1, 16, 16, device=device)
    fg_mask[:, :, 3:11, 4:13] = 1
    return acts, future, subj_1b, subj_2b, emb_mask, pad_mask, fg_mask


def gen_comp_losses(out):
    """REFERENCE ldm/util.py loss functions on the captured-activation stand-ins: values and gradients w.r.t. attn / k / v."""
    import ldm.util as RU
    res = {}

    def grads(acts, tag):
        for key in ("attn", "k", "v"):
            for li in (23, 24):
                g = acts[key][li].grad
                res[f"{tag}.d{key}{li}"] = np.zeros(1, np.float32) if g is None else g.numpy()

    for pct in (0.05, 0.15, 0.22, 0.3):
        acts, future, s1, s2, em, pm, fg = comp_loss_inputs()
        ls = RU.calc_sc_rep_attn_distill_loss(acts, s1, em, pm, pct, FG_THRES=0.1)
        res[f"rep{pct}.values"] = np.asarray([float(v) for v in ls], dtype=np.float64)
        if pct >= 0.1:
            sum(ls).backward()
            grads(acts, f"rep{pct}")
    acts, future, s1, s2, em, pm, fg = comp_loss_inputs()
    l = RU.calc_subj_attn_cross_t_diff_loss(acts, future, s1)
    l.backward()
    res["crosst.value"] = np.asarray(float(l))
    grads(acts, "crosst")
    acts, future, s1, s2, em, pm, fg = comp_loss_inputs()
    l = RU.calc_attn_norm_loss(acts["outfeat"], acts["attn"], s2, 1)
    l.backward()
    res["attnnorm.value"] = np.asarray(float(l))
    grads(acts, "attnnorm")
    acts, future, s1, s2, em, pm, fg = comp_loss_inputs()
    # as called by calc_comp_face_align_and_mb_suppress_losses (ddpm.py:3701-3707): the SC block's attention, the SC face mask
    sc_attn = {li: a.chunk(4)[1] for li, a in acts["attn"].items()}
    l = RU.calc_subj_masked_bg_suppress_loss(sc_attn, s1, 1, fg[:1])
    l.backward()
    res["mbsuppress.value"] = np.asarray(float(l))
    grads(acts, "mbsuppress")
    res["mbsuppress.allfg"] = np.asarray(float(RU.calc_subj_masked_bg_suppress_loss(sc_attn, s1, 1, torch.ones_like(fg[:1]))))
    xs = np.asarray([0.0, 0.1, 0.2, 0.22, 0.25, 0.5])
    res["dyn.x"] = xs
    res["dyn.scale"] = np.asarray([RU.calc_dyn_loss_scale(x, (0.20, 0.5), (0.25, 2), valid_scale_range=(0.05, 2)) for x in xs])
    np.savez_compressed(os.path.join(out, "comp_losses.npz"), **res)
    print("comp_losses:", {k: v.tolist() for k, v in res.items() if k.endswith("values") or k.endswith(".value")})



def comp_preserve_inputs(device="cpu", scale=1.0):
    """Seeded stand-ins for what the feature-matching losses read (layers 22-24 of a four-block batch, BLOCK_SIZE 1, an 8 x 8 feature
    map): q2 / attn_out [4, 32, 64], outfeat [4, 32, 8, 8], with gradients; the SS and SC face boxes [x1, y1, x2, y2]."""
    from adaface_dev_amd import rng
    acts = {"q2": {}, "attn_out": {}, "outfeat": {}}
    for li in (22, 23, 24):
        acts["q2"][li] = (rng.synth_input(f"cp.q{li}", (4, 32, 64), seed=73) * 0.6).to(device).requires_grad_(True)
        acts["attn_out"][li] = (rng.synth_input(f"cp.ao{li}", (4, 32, 64), seed=73) * scale).to(device).requires_grad_(True)
        acts["outfeat"][li] = (rng.synth_input(f"cp.of{li}", (4, 32, 8, 8), seed=73) * scale).to(device).requires_grad_(True)
    ss_boxes = torch.tensor([[1, 1, 6, 7]], device=device)
    sc_boxes = torch.tensor([[2, 3, 7, 8]], device=device)
    return acts, ss_boxes, sc_boxes


PRESERVE_CASES = (("plain", dict(), 0.5), ("shrunk_suppress", dict(sc_face_shrink_ratio_for_bg_matching_mask=0.3, do_sc_fg_faces_suppress=True), 0.5),
                  ("scaled_down", dict(), 1.0), ("discarded", dict(recon_scaled_loss_threses={"mc": 0.05, "ssfg": 2.0}, recon_max_scale_of_threses=2), 1.0))


def gen_comp_preserve(out):
    """REFERENCE ``calc_comp_subj_bg_preserve_loss`` (ldm/util.py:1920-2045; through ``calc_elastic_matching_loss`` :2549-2759 and
    ``calc_sc_recon_ssfg_mc_losses`` :2314-2547) with ``flow_model=None``, the reference's default: value, every monitor entry, gradients
    w.r.t. q2 / attn_out / outfeat; plus ``calc_recon_and_suppress_losses`` (:1715-1754) of the recon iteration."""
    import contextlib
    import io
    import ldm.util as RU
    res = {}
    for tag, kw, scale in PRESERVE_CASES:
        acts, ssb, scb = comp_preserve_inputs(scale=scale)
        mon = {}
        with contextlib.redirect_stdout(io.StringIO()):
            loss = RU.calc_comp_subj_bg_preserve_loss(mon, "train", torch.device("cpu"), None, acts, ssb, scb, **kw)
        loss.backward()
        res[f"{tag}.loss"] = np.asarray(float(loss))
        for k, v in mon.items():
            res[f"{tag}.mon.{k.replace('/', '__')}"] = np.asarray(float(v))
        for key in ("q2", "attn_out", "outfeat"):
            for li in (22, 23, 24):
                g = acts[key][li].grad
                res[f"{tag}.d{key}{li}"] = np.zeros(1, np.float32) if g is None else g.numpy()
    # the recon iteration's per-step losses on the comp_losses stand-ins (BLOCK_SIZE 4 there: every instance is a subject instance)
    from adaface_dev_amd import rng
    acts, future, s1, s2, em, pm, fg = comp_loss_inputs()
    subj4 = (torch.arange(4).repeat_interleave(4), torch.tensor([4, 5, 6, 7]).repeat(4))
    for tag, pure, with_cls, w in (("image", False, True, torch.tensor([1.0, 0.1, 1.0, 1.0])), ("noise", True, True, torch.ones(4)),
                                   ("nocls", False, False, torch.ones(4))):
        eps = rng.synth_input("cp.eps", (4, 4, 16, 16), seed=73).requires_grad_(True)
        gt, cls = rng.synth_input("cp.gt", (4, 4, 16, 16), seed=73), rng.synth_input("cp.cls", (4, 4, 16, 16), seed=73)
        acts, future, s1, s2, em, pm, fg = comp_loss_inputs()
        ls = RU.calc_recon_and_suppress_losses(gt, eps, cls if with_cls else None, w, acts, subj4, None, fg, 0.025, 4, pure)
        tot = sum(l for l in ls if torch.is_tensor(l) and l.requires_grad)
        tot.backward()
        res[f"recon_{tag}.values"] = np.asarray([float(l) for l in ls], dtype=np.float64)
        res[f"recon_{tag}.deps"] = eps.grad.numpy() if eps.grad is not None else np.zeros(1, np.float32)
        res[f"recon_{tag}.dattn23"] = acts["attn"][23].grad.numpy()
    np.savez_compressed(os.path.join(out, "comp_preserve.npz"), **res)
    print("comp_preserve:", {k: float(v) for k, v in res.items() if k.endswith(".loss")}, {k: v.tolist() for k, v in res.items() if k.endswith(".values")})


def _ref_arcface_shell(detect):
    """The REFERENCE ``ArcFaceWrapper`` and ``RetinaFaceClient`` without their constructors (model files / the external retinaface
    package): the embedding network and the detector are the stand-ins, every method run is the reference's own code."""
    import ldm.modules.arcface_wrapper as ref_aw
    import evaluation.retinaface_pytorch as ref_rf
    from standin import StandInFaceNet
    rf = ref_rf.RetinaFaceClient.__new__(ref_rf.RetinaFaceClient)
    torch.nn.Module.__init__(rf)
    rf.detect_faces = lambda img, T=20: [ref_rf.FacialAreaRegion(x, y, w, h, confidence=c) for (x, y, w, h, c) in detect(img, T)]
    aw = ref_aw.ArcFaceWrapper.__new__(ref_aw.ArcFaceWrapper)
    torch.nn.Module.__init__(aw)
    aw.arcface, aw.retinaface, aw.dtype = StandInFaceNet(), rf, torch.float32
    return aw


STAGE2_DETECTORS = ("standin_detect", "no_faces", "standin_detect_small_second_face", "standin_detect_dark_images_faceless")


def gen_stage2_assembly(out):
    """REFERENCE ``LatentDiffusion.calc_comp_feat_distill_loss`` (ddpm.py:3190-3600, with ``calc_comp_face_align_and_mb_suppress_losses``,
    ``redenoise_subj_single``, ``calc_arcface_align_loss`` and the ldm/util.py losses it calls, flow_model None) and
    ``calc_normal_recon_loss`` (:2593-2883, with ``recon_multistep_denoise``) on a constructor-free shell, through
    tests/stage2_scenario.py: loss, every monitor entry, d loss / d prompt_emb."""
    import ldm.util as RU
    import standin
    import stage2_scenario as SC
    res = {}

    def shell(detect):
        ld = _ref_ddpm_shell(trainer=types.SimpleNamespace(global_rank=0), global_step=0, device=torch.device("cpu"), training=True)
        SC.common_attrs(ld, "cpu")
        ld.arcface = _ref_arcface_shell(detect)
        for name in ("comp_sc_face_detected_frac", "comp_mc_face_detected_frac", "comp_sc_face_suppressed_frac", "comp_sc_face_align_loss_kept_frac",
                     "comp_ss_redenoise_success_frac", "normal_recon_face_align_loss_kept_frac"):
            setattr(ld, name, RU.RollingStats(num_values=1, window_size=200, stat_type="mean"))
        ld.normal_recon_face_images_on_image_stats = RU.RollingStats(num_values=2, window_size=600, stat_type="sum")
        ld.normal_recon_face_images_on_noise_stats = RU.RollingStats(num_values=2, window_size=200, stat_type="sum")
        return ld
    for dname in STAGE2_DETECTORS:
        detect = (lambda img, T=20: []) if dname == "no_faces" else getattr(standin, dname)
        for mix in (False, True):
            r = SC.run_comp_feat_distill(shell(detect), "cpu", mix_sc_mc_attn=mix)
            for k, v in r.items():
                res[f"comp.{dname}.mix{int(mix)}.{k}"] = v
        for pure, steps in ((False, 2), (False, 1), (True, 2)):
            r = SC.run_normal_recon(shell(detect), "cpu", on_pure_noise=pure, steps=steps)
            for k, v in r.items():
                res[f"recon.{dname}.pure{int(pure)}.steps{steps}.{k}"] = v
        # the adversarial face edit between the two steps (do_adv_attack, ddpm.py:1879-1913 + calc_arcface_adv_grad :2536-2582)
        ld = shell(detect)
        ld.adaface_adv_iters_count, ld.adaface_adv_success_iters_count, ld.recon_adv_mod_mag_range = 0, 0, [0.001, 0.003]
        r = SC.run_normal_recon(ld, "cpu", on_pure_noise=False, steps=2, do_adv=True)
        r["adv_iters"], r["adv_success"] = np.asarray(float(ld.adaface_adv_iters_count)), np.asarray(float(ld.adaface_adv_success_iters_count))
        for k, v in r.items():
            res[f"recon_adv.{dname}.{k}"] = v
    np.savez_compressed(os.path.join(out, "stage2_assembly.npz"), **res)
    print("stage2_assembly:", {k: float(v) for k, v in res.items() if k.endswith(".loss")})
    print("  monitors of the first case:", sorted(k.split(".mon.")[1] for k in res if k.startswith("comp.standin_detect.mix0.mon.")))


def gen_comp_multistep(out):
    """REFERENCE ``LatentDiffusion.comp_distill_multistep_denoise`` (ddpm.py:1997-2086) around the stand-in wrapper: 3 steps on a
    four-block batch, subject-compos gradient mode; (a) timesteps / noises drawn inside (seeded), (b) a second pass re-using the first
    pass's x_starts (the 'old x_start' mixing), with SC / MC attention mixing (LoRAs forced off)."""
    import json
    from adaface_dev_amd import rng
    from standin import StandInEps, StandInWrapper
    B, h, T, D = 4, 8, 6, 16
    res = {}
    wrapper = StandInWrapper(StandInEps(D, seed=61))
    un = rng.synth_input("cm.uncond", (B, T, D), seed=67)
    ld = _ref_ddpm_shell(model=wrapper, uncond_context=(un[:1], [""], {}), res_hidden_states_gradscale=0.5)
    x0 = rng.synth_input("cm.x0", (1, 4, h, h), seed=67).repeat(B, 1, 1, 1)
    noise = rng.synth_input("cm.noise", (1, 4, h, h), seed=67).repeat(B, 1, 1, 1)
    emb = rng.synth_input("cm.emb", (B, T, D), seed=67).requires_grad_(True)
    t = torch.tensor([900]).repeat(B)
    subj = (torch.tensor([0, 0]), torch.tensor([2, 3]))
    for tag, kw, reuse in (("draw", dict(normalize_cross_attn=True, mix_sc_mc_attn=False, use_attn_lora=True, use_ffn_lora=True), False),
                           ("mix_reuse", dict(normalize_cross_attn=False, mix_sc_mc_attn=True, use_attn_lora=True, use_ffn_lora=True), True)):
        wrapper.calls.clear()
        if not reuse:
            x_starts, noises, ts = [x0], [noise], [t]
        else:
            x_starts, noises, ts = [x.clone() for x in keep[0]], list(keep[1]), list(keep[2])
        torch.manual_seed(97)
        preds, xs, recons, ns, tss, acts = ld.comp_distill_multistep_denoise(x_starts, noises, ts, (emb, [f"p{i}" for i in range(B)], {}), un,
                                                                            all_subj_indices_1b=subj, cfg_scale=2.5, num_denoising_steps=3,
                                                                            ffn_lora_adapter_name="comp_distill", **kw)
        keep = ([x.clone() for x in xs], list(ns), list(tss))
        for i in range(3):
            res[f"{tag}.eps{i}"], res[f"{tag}.recon{i}"] = preds[i].detach().numpy(), recons[i].detach().numpy()
            res[f"{tag}.x{i}"], res[f"{tag}.t{i}"], res[f"{tag}.noise{i}"] = xs[i].numpy(), tss[i].numpy(), ns[i].numpy()
            res[f"{tag}.attn{i}"] = acts[i]["attn"].detach().numpy()
        if emb.grad is not None:
            emb.grad = None
        sum(p.sum() for p in preds).backward()
        res[f"{tag}.demb"] = emb.grad.numpy().copy()
        res[f"{tag}.calls"] = np.asarray(json.dumps([[n, fl, ad, npr, ge] for n, fl, ad, npr, ge in wrapper.calls]))
    np.savez_compressed(os.path.join(out, "comp_multistep.npz"), **res)
    print("comp_multistep:", [res[f"draw.t{i}"].tolist() for i in range(3)], len(json.loads(str(res["draw.calls"]))))



def _ref_shell(cls, **attrs):
    obj = cls.__new__(cls)
    torch.nn.Module.__init__(obj)
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


ID2ADA_CASES = (
    dict(name="given3", init="3", bs=3),
    dict(name="given1_rep3", init="1", bs=3),
    dict(name="avg_img_prompt", init="3", bs=3, avg="img_prompt_emb"),
    dict(name="perturb_id", init="3", bs=3, pstage="id_emb", pstd=0.2),
    dict(name="perturb_prompt", init="3", bs=3, pstage="img_prompt_emb", pstd=0.3),
    dict(name="random_ids", init=None, bs=2),
)


def gen_id2ada_glue(out):
    """The GLUE of the face -> prompt stack executed by the REFERENCE classes themselves, with the CLIP text transformers (whose
    forward does not run under the installed transformers 5) replaced by a stand-in of the same call protocol
    (tests/standin.py::StandInCLIP) and the tokenizer by a six-word stand-in: ``SubjBasisGenerator.forward`` /
    ``inverse_img_prompt_embs`` (subj_basis_generator.py:443-562, 692-770: template, slot replacement, x5 gradient-scaled layer
    weights, static suffix embeddings, core slicing, mixing with the pad embeddings), ``Arc2Face_ID2AdaPrompt
    .map_init_id_to_img_prompt_embs`` (face_id_to_ada_prompt.py:680-724), ``FaceID2AdaPrompt.get_img_prompt_embs`` (:368-470: given /
    repeated / random IDs, the two perturbation stages, L2 normalisation, the averaging stage) and ``generate_adaface_embeddings``
    (:503-578).  Objects are built without their constructors (those download pretrained CLIP files)."""
    import contextlib
    import io
    import adaface.face_id_to_ada_prompt as ref_f2a
    import adaface.subj_basis_generator as ref_sbg
    from adaface_dev_amd import rng
    from standin import StandInCLIP, WordTokenizer
    D, N_ID, N_SFX = 768, 16, 2
    res = {}

    def make_sbg():
        g = _ref_shell(ref_sbg.SubjBasisGenerator, N_ID=N_ID, N_SFX=N_SFX, max_prompt_length=77, dtype=torch.float32, placeholder_is_bg=False,
                       tokenizer=WordTokenizer(), prompt2token_proj=StandInCLIP(D, seed=73), layerwise_proj=torch.nn.Identity(), output_dim=D)
        g.static_img_suffix_embs = torch.nn.Parameter(rng.synth_input("glue.sfx", (1, N_SFX, D), seed=74))
        g.register_buffer("pad_embeddings", rng.synth_input("glue.pad", (77, D), seed=74))
        with contextlib.redirect_stdout(io.StringIO()):
            g.initialize_hidden_state_layer_weights("per-layer", "cpu")
        return g

    for tag, cfg_scale, sfx in (("plain", 1.0, False), ("cfg07_sfx", 0.7, True), ("cfg13", 1.3, False)):
        g = make_sbg()
        x = rng.synth_input("glue.id2img", (2, N_ID, D), seed=74).requires_grad_(True)
        y = g(x, out_id_embs_cfg_scale=cfg_scale, is_face=True, enable_static_img_suffix_embs=sfx)
        (y * rng.synth_input(f"glue.w.{tag}", tuple(y.shape), seed=74)).sum().backward()
        res[f"sbg.{tag}.out"], res[f"sbg.{tag}.dx"] = y.detach().numpy(), x.grad.numpy()
        res[f"sbg.{tag}.dlayer_w"] = g.hidden_state_layer_weights.grad.numpy()
        res[f"sbg.{tag}.dsfx"] = np.zeros(1, np.float32) if g.static_img_suffix_embs.grad is None else g.static_img_suffix_embs.grad.numpy()

    def make_id2ada():
        enc = StandInCLIP(D, seed=75)
        a = _ref_shell(ref_f2a.Arc2Face_ID2AdaPrompt, name="arc2face", tokenizer=WordTokenizer(), id_img_prompt_max_length=22, dtype=torch.float32,
                       text_to_image_prompt_encoder=enc, clip_image_encoder=types.SimpleNamespace(device="cpu"), use_clip_embs=False,
                       gen_neg_img_prompt=False, num_id_vecs=N_ID, num_id_vecs0=N_ID, num_static_img_suffix_embs=N_SFX,
                       default_enable_static_img_suffix_embs=False, out_id_embs_cfg_scale=0.8, subj_basis_generator=make_sbg())
        a.get_clip_neg_features = lambda BS: None
        return a

    ids3 = rng.synth_input("glue.ids", (3, 512), seed=76)
    a = make_id2ada()
    res["map.out"] = a.map_init_id_to_img_prompt_embs(torch.nn.functional.normalize(ids3, dim=-1)).detach().numpy()
    for c in ID2ADA_CASES:
        a = make_id2ada()
        init = None if c["init"] is None else (ids3 if c["init"] == "3" else ids3[:1])
        torch.manual_seed(808)
        with contextlib.redirect_stdout(io.StringIO()):
            cnt, fid, pos, neg = a.get_img_prompt_embs(init, None, None, None, id_batch_size=c["bs"], avg_at_stage=c.get("avg"),
                                                       perturb_at_stage=c.get("pstage"), perturb_std=c.get("pstd", 0.0))
        assert neg is None
        res[f"get.{c['name']}.faceid"], res[f"get.{c['name']}.pos"] = fid.float().numpy(), pos.float().detach().numpy()
    for tag, kw in (("ids_avg_id", dict(face_id_embs=ids3, avg_at_stage="id_emb")), ("ids_noavg", dict(face_id_embs=ids3, avg_at_stage=None)),
                    ("ids_avg_prompt_sfx", dict(face_id_embs=ids3, avg_at_stage="img_prompt_emb", enable_static_img_suffix_embs=True)),
                    ("prompts_avg", dict(img_prompt_embs=rng.synth_input("glue.ip", (3, N_ID, D), seed=76), avg_at_stage="img_prompt_emb"))):
        a = make_id2ada()
        # the reference counts detected face IMAGES and returns (None, None, lens) when that count is 0 (:549-550), which is also what it
        # does for IDs handed in directly (the count is only raised while reading images); report one image so that the rest of the
        # function -- what the fixture is about -- runs
        orig = a.get_img_prompt_embs
        a.get_img_prompt_embs = lambda *args, _o=orig, **kws: (1,) + tuple(_o(*args, **kws))[1:]
        with contextlib.redirect_stdout(io.StringIO()):
            embs, ip, lens = a.generate_adaface_embeddings(None, **kw)
        res[f"gen.{tag}.embs"], res[f"gen.{tag}.img_prompt"], res[f"gen.{tag}.lens"] = embs.detach().numpy(), ip.detach().numpy(), np.asarray(lens)
    np.savez_compressed(os.path.join(out, "id2ada_glue.npz"), **res)
    print("id2ada_glue:", {k: v.shape for k, v in res.items() if k.startswith("gen.") and k.endswith("embs")})



WRAPPER_PROMPTS = ("a z walking a dog", "portrait of the z, oil painting", "z", "an z and a cat, z smiling", None, "photo of a woman",
                   "a zebra next to z")


def gen_wrapper_glue(out):
    """REFERENCE ``AdaFaceWrapper.update_text_encoder_subj_embeddings`` and ``update_prompt`` (adaface/adaface_wrapper.py:461-532) on a
    constructor-free wrapper with a fake pipeline (token table + name -> id map): which rows of the token table receive the subject
    embeddings, the updated-token strings, and the rewritten prompts (append / prepend, per-encoder repetition, null placeholders)."""
    import json
    import adaface.adaface_wrapper as ref_w
    from adaface_dev_amd import rng
    res = {}
    for tag, enc_types, enabled, lens in (("arc2face", ["arc2face"], None, [16]), ("joint_one_disabled", ["consistentID", "arc2face"], ["arc2face"], [4, 16])):
        vocab = {f"z_{i}_{j}": 1000 + 100 * i + j for i in range(len(enc_types)) for j in range(20)}
        table = torch.zeros(1300, 8)
        tok = types.SimpleNamespace(convert_tokens_to_ids=lambda t, v=vocab: v[t])
        te = types.SimpleNamespace(get_input_embeddings=lambda t=table: types.SimpleNamespace(weight=types.SimpleNamespace(data=t)))
        w = ref_w.AdaFaceWrapper.__new__(ref_w.AdaFaceWrapper)
        torch.nn.Module.__init__(w)
        w.subject_string, w.adaface_encoder_types, w.enabled_encoders = "z", enc_types, enabled
        w.pipeline = types.SimpleNamespace(text_encoder=te, tokenizer=tok)
        w.all_null_placeholder_tokens_str = " ".join([","] * sum(lens))
        embs = rng.synth_input(f"wrap.embs.{tag}", (sum(lens), 8), seed=77)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            w.update_text_encoder_subj_embeddings(embs, lens)
        res[f"{tag}.table_rows"] = np.asarray(sorted(int(i) for i in torch.nonzero(table.abs().sum(1)).flatten()))
        res[f"{tag}.table"] = table.numpy()
        info = {"updated_tokens_str": w.updated_tokens_str, "all_encoders_updated_token_strs": w.all_encoders_updated_token_strs, "prompts": {}}
        for pi, prompt in enumerate(WRAPPER_PROMPTS):
            for pos in ("append", "prepend"):
                for rep in (True, False):
                    for null in (False, True):
                        info["prompts"][f"{pi}|{pos}|{int(rep)}|{int(null)}"] = w.update_prompt(prompt, pos, rep, null)
        res[f"{tag}.info"] = np.asarray(json.dumps(info))
    np.savez_compressed(os.path.join(out, "wrapper_glue.npz"), **res)
    print("wrapper_glue:", json.loads(str(res["arc2face.info"]))["prompts"]["0|append|1|0"][:60])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    install_reference_stubs()
    torch.set_num_threads(8)
    out = HERE
    # host-orchestration fixtures: the reference's ddpm.py / unet_teachers.py / diffusers_attn_lora_capture.py are imported with
    # EMPTY stand-ins for their absent third-party packages (tests/golden/ref_import.py)
    host_jobs = {"teacher": gen_teacher, "sdpa": gen_sdpa, "guided_denoise": gen_guided_denoise, "distill_loss": gen_distill_loss,
                 "comp_losses": gen_comp_losses, "comp_preserve": gen_comp_preserve, "stage2_assembly": gen_stage2_assembly, "comp_multistep": gen_comp_multistep, "id2ada_glue": gen_id2ada_glue, "wrapper_glue": gen_wrapper_glue}
    if args.only in host_jobs or args.only is None:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        import ref_import
        ref_import.install(REF)
        for name, fn in host_jobs.items():
            if args.only in (None, name):
                fn(out)
        if args.only is not None:
            return
    jobs = {"embedding_manager": gen_embedding_manager, "blocks": gen_blocks, "schedule": gen_schedule, "train": gen_train, "clip": gen_clip, "arcface": gen_arcface, "vae": gen_vae, "unet_tiny": gen_unet_tiny, "unet_full": gen_unet_full}
    for name, fn in jobs.items():
        if args.only and name != args.only:
            continue
        if name == "unet_full" and args.skip_full:
            continue
        fn(out)


if __name__ == "__main__":
    main()
