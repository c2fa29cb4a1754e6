"""Training-step parity on a real MI355X (`pytest -m gpu`): the fused CAdamW optimizer against the REFERENCE's
4-step trace (tests/golden/train.npz), and the Stage-1 distillation objective (teacher multi-step targets ->
student epsilon -> fg-masked MSE) with its gradient w.r.t. the subject context against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from trainer_util import trainer_setup

pytestmark = pytest.mark.gpu

CFG = dict(in_channels=4, model_channels=64, out_channels=4, num_res_blocks=2, attention_resolutions=[4, 2, 1],
           channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=64, legacy=False)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def test_cadamw_optimizer_vs_reference_trace(dev):
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.c_adamw import CAdamW
    g = np.load(os.path.join(GOLDEN, "train.npz"))
    ps = [torch.nn.Parameter(rng.synth_input("train.p0", (37, 5), seed=8).to(dev)),
          torch.nn.Parameter(rng.synth_input("train.p1", (130,), seed=8).to(dev))]
    opt = CAdamW([{"params": [ps[0]], "weight_decay": 0.02}, {"params": [ps[1]], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.995), eps=1e-6)
    opt.zero_grad()
    for step in range(4):
        for i, p in enumerate(ps):
            p.grad.copy_(rng.synth_input(f"train.g{i}.{step}", p.shape, seed=8))
        opt.step()
        for i, p in enumerate(ps):
            assert rel_l2(p.detach().cpu().numpy(), g[f"cadamw_p{i}_step{step}"]) < 1e-5, (i, step)


@pytest.mark.parametrize("steps", [1, 3])
def test_unet_distill_loss_and_context_grad_vs_oracle(dev, steps):
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.unet_teachers import Arc2FaceTeacher
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from oracle import diffusion_oracle as D
    from oracle import train_oracle as T
    from oracle import unet_oracle as O

    ld = LatentDiffusion(CFG)
    rng.load_synth_weights(ld.model.diffusion_model, seed=11)
    teacher_unet = UNetModel(**CFG)
    rng.load_synth_weights(teacher_unet, seed=12)
    sd_s = {k: v.detach().clone() for k, v in ld.model.diffusion_model.state_dict().items()}
    sd_t = {k: v.detach().clone() for k, v in teacher_unet.state_dict().items()}
    ld = ld.to(dev)
    ld.unet_teacher = Arc2FaceTeacher(teacher_unet.to(dev))

    B = 2
    x0 = rng.synth_input("dist.x0", (B, 4, 32, 32), seed=13)
    noise = rng.synth_input("dist.noise", (B, 4, 32, 32), seed=13)
    t = torch.tensor([760, 850])
    sctx = rng.synth_input("dist.sctx", (B, 77, 64), seed=13)
    tctx = rng.synth_input("dist.tctx", (B, 21, 64), seed=13)
    fg = (rng.synth_input("dist.fg", (B, 1, 32, 32), seed=13) > -0.3).float()
    pres = [(torch.rand(B, generator=torch.Generator().manual_seed(i)), rng.synth_input(f"dist.n{i}", (B, 4, 32, 32), seed=13))
            for i in range(steps - 1)]

    sg = sctx.clone().to(dev).requires_grad_(True)
    loss = ld.calc_unet_distill_loss(x0.to(dev), noise.to(dev), (sg, ["a"] * B, {}), tctx.to(dev), None, fg.to(dev), steps,
                                     t=t.to(dev), presampled=[(r.to(dev), n.to(dev)) for r, n in pres])
    loss.backward()

    tabs = D.register_schedule(D.make_beta_schedule_linear())
    sr = sctx.clone().requires_grad_(True)
    ref = T.unet_distill_loss(lambda x, tt, c: O.unet_forward(sd_s, CFG, x, tt, c, {"res_hidden_states_gradscale": 0.5}),
                              lambda x, tt, c: O.unet_forward(sd_t, CFG, x, tt, c, {}),
                              tabs, x0, noise, t, sr, tctx, fg, steps, pres)
    ref.backward()
    el = abs(float(loss) - float(ref)) / abs(float(ref))
    eg = rel_l2(sg.grad.cpu().numpy(), sr.grad.numpy())
    print(f"steps={steps}: loss {float(loss):.5f} vs {float(ref):.5f} (rel {el:.2e}); dcontext rel-L2 {eg:.2e}")
    # chained fp16 U-Nets (teacher x0 feeds the next step): 5e-3 on the loss, 1e-2 on the gradient (measured 1e-4 / 2.6e-3)
    assert el < 5e-3 and eg < 1e-2


def _oracle_distill(sds, ucfg, ids512, x0, noise, t, fg, steps, pres, sbg_sd=None, ffn_lora=None, skip_weights=None):
    from adaface_dev_amd.adaface.subj_basis_generator import template_ids
    from oracle import clip_oracle as CO
    from oracle import diffusion_oracle as D
    from oracle import train_oracle as T
    from oracle import unet_oracle as O
    cc = dict(hidden=128, heads=2, layers=3)
    B = x0.shape[0]
    sbg = sbg_sd if sbg_sd is not None else {k: v.clone().requires_grad_(True) for k, v in sds["sbg"].items()}
    lw = torch.tensor([[1.0], [2.0], [4.0]], requires_grad=True)
    with torch.no_grad():
        idn = torch.nn.functional.normalize(ids512, dim=-1)
        id2img = CO.id_to_img_prompt(sds["arc2face"], cc, template_ids(["photo", "of", "a", "id", "person"], 22).repeat(B, 1), 4, idn[:, :128])
        prefix = CO.clip_text_forward(sds["arc2face"], cc, template_ids(["photo", "of", "a"], 22))[0][:, :4]
    ada = CO.inverse_img_prompt(sbg, cc, template_ids(["photo", "of", "a"] + [","] * 18, 77).repeat(B, 1), id2img, lw)
    pid = template_ids(["a", "photo", "of"] + [","] * 16, 77).repeat(B, 1)
    tok = sds["text"]["text_model.embeddings.token_embedding.weight"][pid].clone()
    tok = torch.cat([tok[:, :4], ada, tok[:, 20:]], dim=1)
    ctx = CO.clip_text_forward(sds["text"], cc, pid, tok, skip_weights)[0]      # FrozenCLIPEmbedder: LN(sum_k w_k h_k), modules.py:330-339
    tctx = torch.cat([prefix.repeat(B, 1, 1), id2img], dim=1)
    tabs = D.register_schedule(D.make_beta_schedule_linear())
    loss = 8 * T.unet_distill_loss(lambda x, tt, c: O.unet_forward(sds["student"], ucfg, x, tt, c, {"res_hidden_states_gradscale": 0.5, "ffn_lora": ffn_lora}),
                                   lambda x, tt, c: O.unet_forward(sds["teacher"], ucfg, x, tt, c, {}),
                                   tabs, x0, noise, t, ctx, tctx, fg, steps, pres)
    return loss, sbg, lw


def test_unet_teacher_cfg_and_shared_noise_vs_oracle(dev):
    """UNetTeacher.forward (unet_teachers.py:64-187) with classifier-free guidance: doubled [positive; negative] context, a separate
    negative context, and same_t_noise_across_instances -- 3 denoising steps, every eps / x0 / t against the oracle."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from oracle import diffusion_oracle as D
    from oracle import train_oracle as T
    from oracle import unet_oracle as O
    ucfg = dict(CFG, context_dim=128)
    unet = UNetModel(**ucfg)
    rng.load_synth_weights(unet, seed=42)
    sd = {k: v.detach().clone() for k, v in unet.state_dict().items()}
    ld = LatentDiffusion(ucfg).to(dev)
    teacher = UNetTeacher(unet.to(dev), cfg_scale_range=(1.7, 1.7), p_uses_cfg=0.0)
    x0 = rng.synth_input("tc.x0", (2, 4, 32, 32), seed=47)
    noise = rng.synth_input("tc.noise", (2, 4, 32, 32), seed=47)
    pos = rng.synth_input("tc.pos", (2, 20, 128), seed=47)
    neg = rng.synth_input("tc.neg", (2, 20, 128), seed=47)
    t = torch.tensor([800, 720])
    g = torch.Generator().manual_seed(9)
    pres = [(torch.rand(2, generator=g), rng.synth_input(f"tc.n{i}", (2, 4, 32, 32), seed=47)) for i in range(2)]
    tabs = D.register_schedule(D.make_beta_schedule_linear())
    eps_fn = lambda x, tt, c: O.unet_forward(sd, ucfg, x, tt, c, {})
    to = lambda v: v.to(dev)
    cases = [dict(ctx=torch.cat([pos, neg]), negative=None, same=False), dict(ctx=pos, negative=neg, same=False),
             dict(ctx=torch.cat([pos, neg]), negative=None, same=True)]
    for c in cases:
        preds, xs, ns, ts = teacher(ld, to(x0), to(noise), to(t), to(c["ctx"]), negative_context=None if c["negative"] is None else to(c["negative"]),
                                    num_denoising_steps=3, force_uses_cfg=True, same_t_noise_across_instances=c["same"],
                                    presampled=[(to(r), to(n)) for r, n in pres])
        assert teacher.uses_cfg and abs(teacher.cfg_scale - 1.7) < 1e-9
        rp, rx, rn, rt = T.teacher_multistep(eps_fn, tabs, x0, noise, t, c["ctx"], 3, pres, cfg_scale=1.7, negative_ctx=c["negative"],
                                             same_t_noise_across_instances=c["same"])
        for i in range(3):
            assert torch.equal(ts[i].cpu(), rt[i]), (c["same"], i)
            assert rel_l2(preds[i].cpu().numpy(), rp[i].numpy()) < 1e-2, (c["same"], i)
            assert rel_l2(xs[i + 1].cpu().numpy(), rx[i + 1].numpy()) < 1e-2, (c["same"], i)
        if c["same"]:
            assert torch.equal(ts[1][0], ts[1][1]) and torch.equal(ns[1][0], ns[1][1])
    # without CFG a doubled context is reduced to its positive half (extract_pos_context)
    preds, *_ = teacher(ld, to(x0), to(noise), to(t), to(torch.cat([pos, neg])), num_denoising_steps=1)
    assert not teacher.uses_cfg and teacher.cfg_scale == 1
    assert rel_l2(preds[0].cpu().numpy(), T.teacher_multistep(eps_fn, tabs, x0, noise, t, pos, 1, [])[0][0].numpy()) < 1e-2


def test_distill_trainer_micro_batch_loss_and_weight_gradients_vs_oracle(dev):
    """face IDs -> Arc2Face encoder -> trainable SubjBasisGenerator -> frozen text encoder -> student/teacher U-Nets ->
    loss; the gradients of every SubjBasisGenerator weight (flat arena) against autograd through the fp32 CPU oracles."""
    from adaface_dev_amd import rng
    tr, sds, ucfg = trainer_setup(dev)
    BS, steps = 4, 2                        # HALF_BS = 2 (ddpm.py:1283-1289)
    ids512 = rng.synth_input("tr.ids", (BS, 512), seed=46)
    x0 = rng.synth_input("tr.x0", (BS, 4, 32, 32), seed=46)
    noise = rng.synth_input("tr.noise", (BS, 4, 32, 32), seed=46)
    fg = (rng.synth_input("tr.fg", (BS, 1, 32, 32), seed=46) > -0.3).float()
    t = torch.tensor([760, 850])
    pres = [(torch.rand(2, generator=torch.Generator().manual_seed(3)), rng.synth_input("tr.n1", (2, 4, 32, 32), seed=46))]
    batch = dict(x_start=x0.to(dev), face_id_embs=ids512.to(dev), fg_mask=fg.to(dev), noise=noise.to(dev))
    tr.optimizer.zero_grad()
    loss = tr.shared_step(batch, num_unet_denoising_steps=steps, t=t.to(dev), presampled=[(r.to(dev), n.to(dev)) for r, n in pres])
    S = tr.scaler.scale
    (loss * S).backward()
    ref, sbg, lw = _oracle_distill(sds, ucfg, ids512[:2], x0[:2], noise[:2], t, fg[:2], steps, pres)
    ref.backward()
    el = abs(float(loss) - float(ref)) / abs(float(ref))
    assert el < 5e-3, (float(loss), float(ref))
    worst, n_checked = 0.0, 0
    sb = tr.id2ada.subj_basis_generator
    for n, p in sb.prompt2token_proj.named_parameters():
        if not p.requires_grad:
            assert "embeddings" in n
            continue
        gref = sbg[n].grad
        if n.endswith("k_proj.bias"):
            continue                         # true gradient is 0 (softmax shift invariance)
        e = rel_l2((p.grad / S).cpu().numpy(), gref.numpy())
        worst = max(worst, e)
        n_checked += 1
        assert e < 3e-2, (n, e)
    e_w = rel_l2((sb.hidden_state_layer_weights.grad / S).cpu().numpy(), 5.0 * lw.grad.numpy())      # x5 grad scaler (:716)
    print(f"trainer micro-batch: loss rel err {el:.2e}; worst weight-gradient rel-L2 over {n_checked} tensors {worst:.2e}; layer-mix {e_w:.2e}")
    assert e_w < 3e-2


def test_distill_trainer_through_embedding_manager_vs_oracle(dev):
    """The same micro-batch through the reference's conditioning path: prompts -> FrozenCLIPEmbedder (skip weights 0.5 / 0.5 over the
    last two layers) with the EmbeddingManager generating and patching the ada embeddings inside the embedding step."""
    from adaface_dev_amd import rng
    tr, sds, ucfg = trainer_setup(dev, embedding_manager=True)
    BS, steps = 4, 2
    ids512 = rng.synth_input("tr.ids", (BS, 512), seed=46)
    x0 = rng.synth_input("tr.x0", (BS, 4, 32, 32), seed=46)
    noise = rng.synth_input("tr.noise", (BS, 4, 32, 32), seed=46)
    fg = (rng.synth_input("tr.fg", (BS, 1, 32, 32), seed=46) > -0.3).float()
    t = torch.tensor([760, 850])
    pres = [(torch.rand(2, generator=torch.Generator().manual_seed(3)), rng.synth_input("tr.n1", (2, 4, 32, 32), seed=46))]
    batch = dict(x_start=x0.to(dev), face_id_embs=ids512.to(dev), fg_mask=fg.to(dev), noise=noise.to(dev))
    tr.optimizer.zero_grad()
    loss = tr.shared_step(batch, num_unet_denoising_steps=steps, t=t.to(dev), presampled=[(r.to(dev), n.to(dev)) for r, n in pres])
    S = tr.scaler.scale
    (loss * S).backward()
    em = tr.ldm.embedding_manager
    b, n = em.placeholder2indices["z"]
    assert b.tolist() == [0] * 16 + [1] * 16 and n.tolist() == list(range(4, 20)) * 2
    ref, sbg, lw = _oracle_distill(sds, ucfg, ids512[:2], x0[:2], noise[:2], t, fg[:2], steps, pres, skip_weights=torch.tensor([[0.5], [0.5]]))
    plain = _oracle_distill(sds, ucfg, ids512[:2], x0[:2], noise[:2], t, fg[:2], steps, pres)[0]
    ref.backward()
    el = abs(float(loss) - float(ref)) / abs(float(ref))
    assert el < 5e-3 and abs(float(plain) - float(ref)) / abs(float(ref)) > 2 * el, (float(loss), float(ref), float(plain))
    sb = tr.id2ada.subj_basis_generator
    worst = 0.0
    for name, p in sb.prompt2token_proj.named_parameters():
        if not p.requires_grad or name.endswith("k_proj.bias"):
            continue
        worst = max(worst, rel_l2((p.grad / S).cpu().numpy(), sbg[name].grad.numpy()))
    print(f"trainer through the embedding manager: loss rel err {el:.2e}; worst weight-gradient rel-L2 {worst:.2e}")
    assert worst < 3e-2


def test_distill_trainer_accumulate_and_step(dev):
    """Two micro-batches with accumulate_grad_batches=2: one optimizer step, lr from the warm-up/cosine schedule, update
    direction -lr*sign(g) at step 1 of cautious AdamW on (almost) every element, gradients zeroed afterwards."""
    from adaface_dev_amd import rng
    tr, sds, ucfg = trainer_setup(dev, accum=2)
    p0 = tr.arena.flat_p.clone()
    BS = 4
    batches = []
    for i in range(2):
        batches.append(dict(x_start=rng.synth_input(f"ts.x{i}", (BS, 4, 32, 32), seed=47).to(dev),
                            face_id_embs=rng.synth_input(f"ts.id{i}", (BS, 512), seed=47).to(dev),
                            fg_mask=torch.ones(BS, 1, 32, 32, device=dev)))
    l0 = tr.training_step(batches[0], 0)
    assert tr.global_step == 0 and float(tr.arena.flat_g.abs().sum()) > 0
    g_after_first = tr.arena.flat_g.clone()
    # gradient clipping by value (yaml gradient_clip_val 0.01, 'value') runs once per optimizer step on the accumulated
    # gradients (Lightning automatic optimization), not after the first micro-batch of the window
    seen = {}
    orig_step = tr.optimizer.step

    def step():
        seen["g"] = tr.arena.flat_g.clone()
        return orig_step()
    tr.optimizer.step = step
    l1 = tr.training_step(batches[1], 1)
    assert tr.global_step == 1 and tr.skipped_steps == 0
    assert torch.isfinite(l0) and torch.isfinite(l1)
    assert float(seen["g"].abs().max()) <= 0.01 * (1 + 1e-6)          # what the optimizer saw: unscaled, clipped
    assert float(tr.arena.flat_g.abs().sum()) == 0.0
    dp = tr.arena.flat_p - p0
    lr = tr.learning_rate * tr.lr_lambda(0)
    moved = dp != 0
    assert moved.float().mean() > 0.95
    # first cautious-AdamW step: |dp| <= lr / mask_mean-ish, sign opposite to the accumulated gradient
    assert float(dp.abs().max()) <= lr * 1.01 / 0.999
    assert g_after_first.shape == dp.shape
    assert tr.unet_distill_iters_count == 2


def test_data_parallel_step_two_ranks_equals_accumulated_single_process(dev, tmp_path):
    """world_size 2 (two processes sharing this GPU, gloo transport -- RCCL refuses two ranks on one device), one micro-batch
    per rank, bucketed all-reduce from the backward hooks  ==  one process accumulating the same two micro-batches:
    identical mean gradient, identical lr (accum * world * bs * base_lr), identical parameters after the CAdamW step."""
    import subprocess
    import sys
    from adaface_dev_amd import rng
    out = tmp_path / "ddp_params.pt"
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=os.path.dirname(here) + os.pathsep + here)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(here, "ddp_train_worker.py"), str(out)], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    tr, _, _ = trainer_setup(dev, accum=2)
    t = torch.tensor([760, 850, 800, 720], device=dev)
    for i in range(2):
        b = dict(x_start=rng.synth_input(f"dp.x{i}", (4, 4, 32, 32), seed=48).to(dev), face_id_embs=rng.synth_input(f"dp.id{i}", (4, 512), seed=48).to(dev),
                 fg_mask=torch.ones(4, 1, 32, 32, device=dev), noise=rng.synth_input(f"dp.n{i}", (4, 4, 32, 32), seed=48).to(dev))
        tr.training_step(b, i, num_unet_denoising_steps=1, t=t)
    assert tr.global_step == 1 and got["global_step"] == 1 and got["world"] == 2
    assert abs(got["lr"] - tr.learning_rate) < 1e-12
    ref = tr.arena.flat_p.cpu()
    # gradients agree to fp32 summation order; the sign-like first CAdamW step turns the few near-zero elements whose sign
    # flips into +-lr outliers, so compare the update direction on (almost) all elements and the mean gradient exactly-ish
    same = ((got["flat_p"] - got["p0"]).sign() == (ref - got["p0"]).sign()).float().mean()
    assert same > 0.999, float(same)
    assert rel_l2(got["mean_grad"].numpy(), got["mean_grad_expected_from_rank_sums"].numpy()) < 1e-5


def test_unet_ffn_dora_training_gradients_vs_oracle(dev):
    """Stage-1 student pass with the `unet_distill` FFN DoRA adapters ON (ddpm.py:3130-3134): eps, d/dcontext and the gradients
    of all 18 adapter tensors (A, B, magnitude of conv1 / conv2 / conv_shortcut of the last two output blocks) against autograd
    through the CPU oracle with the same adapters and the same dropout masks."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules import dora as DR
    from oracle import unet_oracle as O
    ld = LatentDiffusion(CFG)
    rng.load_synth_weights(ld.model.diffusion_model, seed=11)
    sd = {k: v.detach().clone() for k, v in ld.model.diffusion_model.state_dict().items()}
    ld = ld.to(dev)
    lora = ld.model.set_up_ffn_loras(lora_rank=16, lora_dropout=0.1)
    with torch.no_grad():
        for n, p in lora.named_parameters():
            if "lora_B" in n:
                p.copy_(rng.synth_input(n, p.shape, seed=81, scale=0.3))
            elif "magnitude" in n:
                p.mul_(1.0 + 0.2 * rng.synth_input(n, p.shape, seed=81).abs().to(dev))
    lora.train()
    B = 2
    x = rng.synth_input("t64.x", (B, 4, 32, 32), seed=11)
    ctx = rng.synth_input("t64.ctx", (B, 77, 64), seed=11)
    cot = rng.synth_input("t64.cot", (B, 4, 32, 32), seed=11)
    t = torch.tensor([10, 500])
    # fixed dropout masks shared with the oracle: patch draw_mask of every active adapter
    masks = {}
    act = lora.active("unet_distill")
    for bi, ads in act.items():
        for key, ad in ads.items():
            def draw(shape, device, generator=None, _k=(bi, key)):
                g = torch.Generator().manual_seed(hash(_k) % 1000)
                m = ((torch.rand(shape, generator=g) >= 0.1).float() / 0.9).half()
                masks[_k] = m
                return m.to(device)
            ad.draw_mask = draw
    cg = ctx.clone().to(dev).requires_grad_(True)
    eps = ld.apply_model(x.to(dev), t.to(dev), (cg, ["a"] * B, {}), use_ffn_lora=True, ffn_lora_adapter_name="unet_distill")
    (eps * cot.to(dev)).sum().backward()
    # oracle
    P = {}
    ffn = {}
    for bi, ads in act.items():
        pre = f"output_blocks.{bi}.0."
        ffn[pre] = {}
        for key, ad in ads.items():
            A, Bm, m = (getattr(ad, n).detach().float().cpu().requires_grad_(True) for n in ("lora_A", "lora_B", "lora_magnitude_vector"))
            P[(bi, key)] = (A, Bm, m)
            ffn[pre][key] = (A, Bm, m, ad.scaling, masks[(bi, key)].float().permute(0, 3, 1, 2))
    cr = ctx.clone().requires_grad_(True)
    ref = O.unet_forward(sd, CFG, x, t, cr, {"ffn_lora": ffn, "res_hidden_states_gradscale": 1})
    (ref * cot).sum().backward()
    assert rel_l2(eps.detach().cpu().numpy(), ref.detach().numpy()) < 5e-3
    assert rel_l2(cg.grad.cpu().numpy(), cr.grad.numpy()) < 2e-2
    worst = 0.0
    for (bi, key), (A, Bm, m) in P.items():
        ad = act[bi][key]
        for got, want, nm in ((ad.lora_A.grad, A.grad, "A"), (ad.lora_B.grad, Bm.grad, "B"), (ad.lora_magnitude_vector.grad, m.grad, "m")):
            assert got is not None, (bi, key, nm)
            e = rel_l2(got.cpu().numpy(), want.numpy())
            worst = max(worst, e)
            assert e < 2e-2, (bi, key, nm, e)
    print(f"FFN DoRA: eps {rel_l2(eps.detach().cpu().numpy(), ref.detach().numpy()):.2e}; worst adapter-gradient rel-L2 over {3 * len(P)} tensors {worst:.2e}")
    # inactive adapters received nothing; inference with the module-held adapters goes through the merged weights
    assert all(p.grad is None for p in lora.adapters["recon_loss"].parameters())
    lora.eval()
    with torch.no_grad():
        e_merge = ld.apply_model(x.to(dev), t.to(dev), (ctx.to(dev), ["a"] * B, {}), use_ffn_lora=True, ffn_lora_adapter_name="unet_distill")
        e_base = ld.apply_model(x.to(dev), t.to(dev), (ctx.to(dev), ["a"] * B, {}))
    ffn_eval = {pre: {k: v[:4] + (None,) for k, v in d.items()} for pre, d in ffn.items()}
    with torch.no_grad():
        ref_eval = O.unet_forward(sd, CFG, x, t, ctx, {"ffn_lora": ffn_eval})
    assert rel_l2(e_merge.cpu().numpy(), ref_eval.numpy()) < 5e-3 and rel_l2(e_base.cpu().numpy(), ref_eval.numpy()) > 1e-2


def test_distill_trainer_with_ffn_dora_param_group(dev):
    """The Stage-1 micro-batch with the U-Net's `unet_distill` FFN DoRA adapters trainable next to the SubjBasisGenerator (two
    parameter groups / arenas, the adapters' with weight decay 0.02): loss and the adapter gradients against the oracle, then one
    accumulated optimizer step moves both groups and leaves the inactive adapters alone."""
    from adaface_dev_amd import rng
    tr, sds, ucfg = trainer_setup(dev, accum=2, ffn_lora=True)
    assert len(tr.arenas) == 2 and tr.optimizer.param_groups[1]["weight_decay"] == 0.02
    lora = tr.ffn_lora
    BS, steps = 4, 2
    ids512 = rng.synth_input("tr.ids", (BS, 512), seed=46)
    x0 = rng.synth_input("tr.x0", (BS, 4, 32, 32), seed=46)
    noise = rng.synth_input("tr.noise", (BS, 4, 32, 32), seed=46)
    fg = (rng.synth_input("tr.fg", (BS, 1, 32, 32), seed=46) > -0.3).float()
    t = torch.tensor([760, 850])
    pres = [(torch.rand(2, generator=torch.Generator().manual_seed(3)), rng.synth_input("tr.n1", (2, 4, 32, 32), seed=46))]
    batch = dict(x_start=x0.to(dev), face_id_embs=ids512.to(dev), fg_mask=fg.to(dev), noise=noise.to(dev))
    tr.optimizer.zero_grad()
    loss = tr.shared_step(batch, num_unet_denoising_steps=steps, t=t.to(dev), presampled=[(r.to(dev), n.to(dev)) for r, n in pres])
    S = tr.scaler.scale
    (loss * S).backward()
    act = lora.active("unet_distill")
    P, ffn = {}, {}
    for bi, ads in act.items():
        pre = f"output_blocks.{bi}.0."
        ffn[pre] = {}
        for key, ad in ads.items():
            A, Bm, m = (getattr(ad, n).detach().float().cpu().requires_grad_(True) for n in ("lora_A", "lora_B", "lora_magnitude_vector"))
            P[(bi, key)] = (A, Bm, m)
            ffn[pre][key] = (A, Bm, m, ad.scaling, None)
    ref, _, _ = _oracle_distill(sds, ucfg, ids512[:2], x0[:2], noise[:2], t, fg[:2], steps, pres, ffn_lora=ffn)
    ref.backward()
    assert abs(float(loss) - float(ref)) / abs(float(ref)) < 5e-3
    worst = 0.0
    for (bi, key), (A, Bm, m) in P.items():
        ad = act[bi][key]
        for got, want in ((ad.lora_A.grad, A.grad), (ad.lora_B.grad, Bm.grad), (ad.lora_magnitude_vector.grad, m.grad)):
            e = rel_l2((got / S).cpu().numpy(), want.numpy())
            worst = max(worst, e)
            assert e < 3e-2, (bi, key, e)
    print(f"trainer with FFN DoRA: loss {float(loss):.5f} vs {float(ref):.5f}; worst adapter-gradient rel-L2 {worst:.2e}")
    # one accumulated optimizer step
    tr.optimizer.zero_grad()
    p_sbg, p_lora = tr.arenas[0].flat_p.clone(), tr.arenas[1].flat_p.clone()
    other = [p.detach().clone() for p in lora.adapters["recon_loss"].parameters()]
    for i in range(2):
        tr.training_step(batch, i, num_unet_denoising_steps=steps, t=t.to(dev), presampled=[(r.to(dev), n.to(dev)) for r, n in pres])
    assert tr.global_step == 1 and tr.skipped_steps == 0
    assert float((tr.arenas[0].flat_p - p_sbg).abs().max()) > 0 and float((tr.arenas[1].flat_p - p_lora).abs().max()) > 0
    assert all(torch.equal(a, b.detach()) for a, b in zip(other, lora.adapters["recon_loss"].parameters()))


def test_weight_packs_follow_the_optimizer(dev):
    """The fused CAdamW kernel rewrites the flat fp32 arena behind autograd's back (p._version never moves): every cached
    fp16 weight pack of a trainable layer must be rebuilt after optimizer.step().  Two optimizer steps with a large lr; after
    each one the generator's output must equal what a FRESHLY built module holding the same fp32 masters computes, and must
    differ from the output before the step (frozen packs would reproduce it bit for bit)."""
    import copy
    from adaface_dev_amd import rng
    tr, _, _ = trainer_setup(dev, accum=1, ffn_lora=True)
    for g in tr.optimizer.param_groups:
        g["lr"] = 1e-2
    tr.learning_rate = 1e-2
    sbg = tr.id2ada.subj_basis_generator
    x = rng.synth_input("stale.id2img", (2, 16, 128), seed=77).to(dev)

    def out_of(m):
        with torch.no_grad():
            return m(x, out_id_embs_cfg_scale=1.0, is_face=True).float().cpu().numpy()

    b = dict(x_start=rng.synth_input("stale.x", (2, 4, 32, 32), seed=77).to(dev), face_id_embs=rng.synth_input("stale.id", (2, 512), seed=77).to(dev),
             fg_mask=torch.ones(2, 1, 32, 32, device=dev))
    prev = out_of(sbg)
    lora = tr.ldm.model.ffn_lora
    ad = next(iter(next(iter(lora.active("unet_distill").values())).values()))
    for it in range(2):
        pk_before = ad._packs()[0].wt.clone()
        tr.training_step(b, it, num_unet_denoising_steps=1)
        assert tr.global_step == it + 1
        now = out_of(sbg)
        assert rel_l2(now, prev) > 1e-4, "the forward did not see the optimizer step (stale fp16 packs)"
        fresh = copy.deepcopy(sbg)                     # new module objects: empty caches, same fp32 master values
        for m in fresh.modules():
            for name in ("_cache", "_cache_bwd"):
                if hasattr(m, name):
                    setattr(m, name, type(getattr(m, name))())
        assert rel_l2(now, out_of(fresh)) < 1e-6
        # the DoRA adapter packs are rebuilt too
        assert not torch.equal(ad._packs()[0].wt, pk_before)
        prev = now


def test_rccl_world1_train_step_under_launcher(dev, tmp_path):
    """One rank under torch.distributed.run with backend "nccl" (= RCCL): process-group init on the device, the start-up
    parameter broadcast, every gradient bucket all-reduced from the backward hooks on RCCL's stream (reduce_single_rank),
    the overflow-flag all-reduce and the optimizer step.  With one rank every collective is an identity, so the result must
    equal the plain single-process step."""
    import subprocess
    import sys
    from adaface_dev_amd import rng
    out = tmp_path / "rccl1.pt"
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=os.path.dirname(here) + os.pathsep + here, AF_DDP_BACKEND="nccl",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29619", os.path.join(here, "ddp_train_worker.py"), str(out)], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    assert got["backend"] == "nccl" and got["world"] == 1 and got["global_step"] == 1
    assert got["collectives"] >= 2 and got["launch_log"] == sorted(got["launch_log"])
    tr, _, _ = trainer_setup(dev, accum=1)
    t = torch.tensor([760, 850, 800, 720], device=dev)
    b = dict(x_start=rng.synth_input("dp.x0", (4, 4, 32, 32), seed=48).to(dev), face_id_embs=rng.synth_input("dp.id0", (4, 512), seed=48).to(dev),
             fg_mask=torch.ones(4, 1, 32, 32, device=dev), noise=rng.synth_input("dp.n0", (4, 4, 32, 32), seed=48).to(dev))
    tr.training_step(b, 0, num_unet_denoising_steps=1, t=t)
    assert rel_l2(got["mean_grad"].numpy(), got["mean_grad_expected_from_rank_sums"].numpy()) < 1e-6
    same = ((got["flat_p"] - got["p0"]).sign() == (tr.arena.flat_p.cpu() - got["p0"]).sign()).float().mean()
    assert same > 0.999, float(same)


def test_comp_distill_iteration_does_not_stall_the_host(dev):
    """A compositional-distillation micro-batch used to make the host wait for the device ~480 times (slices indexed by device-resident
    face boxes, index lists, one read per monitor / per `if loss > 0`): every wait drains the launch queue, and the leg was host-bound
    because of it (profiles/r04p_host_syncs_train2_before.txt).  The detector's boxes / confidences / masks are host data now and the
    values the assembly branches on are read in batches: count the synchronisations of one micro-batch (torch's sync-debug mode reports
    each blocking copy / .item() / nonzero as a warning) and hold the line at 80 (51 at full size, profiles/r04q_host_syncs_train2_after.txt)."""
    import warnings
    from adaface_dev_amd import rng
    tr, sds, ucfg = trainer_setup(dev, accum=1, ffn_lora=True, stage2=True)
    b = dict(x_start=rng.synth_input("s2.x", (2, 4, 32, 32), seed=49).to(dev), face_id_embs=rng.synth_input("s2.id", (2, 512), seed=49).to(dev))
    for aug in ("normalize_cross_attn", "mix_sc_mc_attn"):                     # warm: packs, template ids, grey weights
        tr.optimizer.zero_grad()
        tr.comp_distill_step(b, attn_aug=aug)
    counts = {}
    for aug in ("normalize_cross_attn", "mix_sc_mc_attn"):
        tr.optimizer.zero_grad()
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            torch.cuda.set_sync_debug_mode("warn")
            try:
                loss = tr.comp_distill_step(b, attn_aug=aug)
            finally:
                torch.cuda.set_sync_debug_mode("default")
        counts[aug] = sum(1 for x in w if "synchroniz" in str(x.message).lower())
        assert torch.isfinite(loss)
    print("host<->device synchronisations per compositional micro-batch:", counts)
    assert max(counts.values()) <= 80, counts


def test_gradient_free_captured_pass_takes_the_inference_trunk(dev):
    """`unet_forward_captured` of an instance that needs no gradient below the captured layers (the no-grad SS / SR instances, the
    re-denoising pass) walks the trunk with the inference kernels (`UNetModel.hip_trunk`) instead of the activation-saving walk: same
    function, different launches -- eps and every captured tensor agree within the fp16 tolerance of two kernel paths."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.diffusionmodules import capture_graph as CG
    tr, sds, ucfg = trainer_setup(dev, accum=1, ffn_lora=True, stage2=True)
    unet = tr.ldm.model.diffusion_model
    x = rng.synth_input("trunk.x", (2, 4, 32, 32), seed=51).to(dev)
    ctx = rng.synth_input("trunk.c", (2, 77, ucfg["context_dim"]), seed=52).to(dev)
    t = torch.tensor([500, 300], device=dev)
    outs = []
    for flag in (True, False):
        CG.INFER_TRUNK = flag
        try:
            ei = {"capture_ca_activations": True, "normalize_cross_attn": False, "subj_indices": None}
            with torch.no_grad():
                eps = CG.unet_forward_captured(unet, x, t, ctx, ei)
            outs.append((eps.float(), ei["ca_layers_activations"]))
        finally:
            CG.INFER_TRUNK = True
    (e1, a1), (e0, a0) = outs
    assert torch.isfinite(e1).all()
    assert rel_l2(e1.cpu().numpy(), e0.cpu().numpy()) < 5e-3
    for key in ("outfeat", "attn", "q", "k", "v", "attn_out"):
        for li in a0[key]:
            assert rel_l2(a1[key][li].float().cpu().numpy(), a0[key][li].float().cpu().numpy()) < 5e-3, (key, li)


@pytest.mark.parametrize("attn_aug", ["normalize_cross_attn", "mix_sc_mc_attn"])
def test_comp_distill_iteration_reduced_width(dev, attn_aug):
    """One Stage-2 compositional-distillation micro-batch end to end on the HIP path (reference ddpm.py:2371-2480): priming by the
    second U-Net with classifier-free guidance, four subject-compos denoising steps with capture (no-grad SS / SR / MC passes, the SC
    pass -- or the joint SC+MC pass with mixed scores -- with gradients, attention + FFN adapters), the captured-activation losses
    with a supplied face box, backward.  Checks: finite positive loss with every expected term, gradients reach the
    SubjBasisGenerator, the attention adapters (not while scores are mixed: LoRAs are off then, :2004-2006) and, with normalisation,
    the score scale factors; then a full optimizer step through training_step."""
    from adaface_dev_amd import rng
    tr, sds, ucfg = trainer_setup(dev, accum=1, ffn_lora=True, stage2=True)
    assert tr.iter_type == "comp_distill" and len(tr.arenas) == 2
    b = dict(x_start=rng.synth_input("s2.x", (2, 4, 32, 32), seed=49).to(dev), face_id_embs=rng.synth_input("s2.id", (2, 512), seed=49).to(dev))
    tr.optimizer.zero_grad()
    torch.manual_seed(5)
    loss = tr.comp_distill_step(b, attn_aug=attn_aug)
    mon = tr.mon_loss_dict
    assert torch.isfinite(loss) and float(loss) > 0
    # the fixed detector box (trainer_util.fixed_face_detector) covers 20 x 20 of the 32 x 32 latent: 'too-large' for the reference's
    # bounds, so on top of the rep-distillation / mb-suppress terms the face suppression and the feature-matching loss are live
    for k in ("comp_rep_distill_subj_attn", "comp_rep_distill_nonsubj_k", "comp_sc_subj_mb_suppress", "pred_l2", "sc_fg_mask_percent", "comp_rep_distill_total",
              "comp_fg_bg_preserve", "sc_recon_mc_min", "arcface_align_comp", "comp_ss_redenoise_success_frac", "comp_sc_face_suppressed_frac"):
        assert f"train/{k}" in mon, (k, sorted(mon))
    assert abs(mon["train/sc_fg_mask_percent"] - 400 / 1024) < 1e-6 and tr.ldm.sc_face_proportion_type in ("too-large", "little-no-overlap")
    (loss * tr.scaler.scale).backward()
    g_sbg = tr.arenas[0].flat_g
    assert torch.isfinite(g_sbg).all() and float(g_sbg.abs().sum()) > 0
    alora = tr.ldm.model.attn_lora
    g_attn = sum(float(p.grad.abs().sum()) for p in alora.parameters() if p.grad is not None)
    g_fac = sum(float(p.grad.abs().sum()) for p in tr.ldm.model.cross_attn_scale_factors.values() if p.grad is not None)
    if attn_aug == "mix_sc_mc_attn":
        assert g_attn == 0 and g_fac == 0
    else:
        assert g_attn > 0 and g_fac > 0
    # the whole training step (loss scaling, clip, fused CAdamW on both arenas)
    p0 = [a.flat_p.clone() for a in tr.arenas]
    tr.optimizer.zero_grad()
    l = tr.training_step(b, 0, attn_aug=attn_aug)
    assert torch.isfinite(l) and tr.global_step == 1 and tr.skipped_steps == 0
    assert float((tr.arenas[0].flat_p - p0[0]).abs().sum()) > 0


def test_arcface_terms_back_propagate_into_the_trainable_parameters(dev):
    """The Stage-2 iteration twice from identical state and seeds: as built, and with the decoded x0 prediction detached in front of the
    face pipeline (what the package did before the VAE decoder / ResNetFace-18 had input-gradient kernels).  Same loss value, same
    monitors -- but the SubjBasisGenerator's gradient differs, by the ArcFace alignment / face-suppression terms' contribution
    (ddpm.py:2511-2535 through decode_first_stage_with_grad, :899-908)."""
    from adaface_dev_amd import rng

    def run(sever):
        tr, _, _ = trainer_setup(dev, accum=1, ffn_lora=True, stage2=True)
        b = dict(x_start=rng.synth_input("s2.x", (2, 4, 32, 32), seed=49).to(dev), face_id_embs=rng.synth_input("s2.id", (2, 512), seed=49).to(dev))
        ld = tr.ldm
        ld.arcface_align_loss_weight = 1.0           # 100 x the default: the terms' gradient share becomes percents instead of 5e-4
        if sever:
            attached = ld.decode_first_stage_with_grad
            ld.decode_first_stage_with_grad = lambda z: attached(z.detach())
        tr.optimizer.zero_grad()
        torch.manual_seed(5)
        loss = tr.comp_distill_step(b, attn_aug="normalize_cross_attn")
        (loss * tr.scaler.scale).backward()
        return float(loss.detach()), dict(tr.mon_loss_dict), tr.arenas[0].flat_g.clone()
    l1, m1, g1 = run(False)
    l0, m0, g0 = run(True)
    # (the attached decode keeps GroupNorm statistics for its backward and so runs the two-launch GroupNorm where the plain decode uses the
    # one-launch forms: the decoded images agree to fp16 rounding, which the face embedding's cosine amplifies to ~1 % of the alignment loss)
    assert abs(l1 - l0) <= 2e-2 * abs(l0) and "train/arcface_align_comp" in m1
    assert abs(m1["train/arcface_align_comp"] - m0["train/arcface_align_comp"]) <= 5e-2 * abs(m0["train/arcface_align_comp"])
    assert torch.isfinite(g1).all() and torch.isfinite(g0).all()
    d = float((g1 - g0).norm()) / float(g0.norm())
    print(f"SubjBasisGenerator gradient: relative change by the ArcFace terms' gradient {d:.3e}")
    assert d > 1e-2


def test_no_vendor_gemm_or_convolution_on_the_training_path(dev):
    """One micro-batch of every iteration type (distillation, normal recon with the face pipeline, compositional distillation) under a
    dispatch spy: no aten mm / addmm / bmm / mv / convolution may run on a device tensor -- every matrix product of the path is one of this
    package's kernels (torch's ``@`` / ``F.conv2d`` on the GPU are rocBLAS / hipBLASLt / MIOpen calls).  Found three such call sites in
    round 3 (the DoRA weight norm, the merged LoRA weights of the no-grad passes, a 3x3 Laplacian)."""
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    from adaface_dev_amd import rng
    hits = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if any(k in name for k in ("aten.mm", "aten.addmm", "aten.bmm", "aten.baddbmm", "aten.convolution", "aten.mv", "aten.addmv", "aten._scaled_mm")):
                if any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
                    where = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "adaface" in f.filename]
                    hits.append((name, where[-3:]))
            return func(*args, **(kwargs or {}))

    tr, _, _ = trainer_setup(dev, accum=2, ffn_lora=True, faces=True)
    tr.ldm.uncond_context = (rng.synth_input("s2.uncond", (1, 77, 128), seed=46).to(dev), [""], {})
    tr.unet_distill_iter_gap = 2
    b = dict(x_start=rng.synth_input("nv.x", (2, 4, 32, 32), seed=61).to(dev), face_id_embs=rng.synth_input("nv.id", (2, 512), seed=61).to(dev),
             fg_mask=torch.ones(2, 1, 32, 32, device=dev))
    kinds = []
    with Spy():
        for i in range(2):                                   # a recon micro-batch, then a distillation one (+ the optimizer step)
            tr.training_step(b, i, **(dict(on_pure_noise=False) if i == 0 else {}))
            kinds.append(tr.last_iter_type)
    assert kinds == ["normal_recon", "unet_distill"]
    tr2, _, _ = trainer_setup(dev, accum=1, ffn_lora=True, stage2=True)
    with Spy():
        tr2.training_step(b, 0)
    assert tr2.last_iter_type == "comp_distill"
    assert not hits, hits[:5]


def test_graph_replayed_segments_reproduce_the_eager_micro_batch(dev):
    """use_graphs: the teacher's forward and the student U-Net's forward / backward walks are captured into hipGraphs on their second
    call per signature and replayed afterwards.  Six micro-batches of one signature (explicit timesteps / noise, one denoising step, no
    dropout: nothing random inside the segments) -- eager, captured and replayed ones -- must give bit-identical losses and
    parameter trajectories to a trainer without graphs, across optimizer steps that move the FFN adapters (their packs are
    refreshed in place for the replays)."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.trainer import DistillTrainer, LossScaler

    def run(use_graphs):
        tr, _, _ = trainer_setup(dev, accum=1, ffn_lora=True)
        tr.scaler = LossScaler(init_scale=2.0 ** 8)     # headroom for the fp16 weight-gradient GEMMs of the adapters at lr 1e-3
        for ad in (a for d in tr.ldm.model.ffn_lora.active("unet_distill").values() for a in d.values()):
            ad.p = 0.0
        if use_graphs:
            tr2 = DistillTrainer(tr.ldm, tr.id2ada, tr.text_encoder, accumulate_grad_batches=1, warm_up_steps=0,
                                 loss_scaler=LossScaler(init_scale=2.0 ** 8), use_graphs=True)
            tr.reducer.remove()
            tr = tr2
        for g in tr.optimizer.param_groups:
            g["lr"] = 1e-3
        tr.learning_rate = 1e-3
        t = torch.tensor([760, 850, 800, 720], device=dev)
        losses = []
        for i in range(6):
            b = dict(x_start=rng.synth_input(f"dp.x{i % 2}", (4, 4, 32, 32), seed=48).to(dev), face_id_embs=rng.synth_input(f"dp.id{i % 2}", (4, 512), seed=48).to(dev),
                     fg_mask=torch.ones(4, 1, 32, 32, device=dev), noise=rng.synth_input(f"dp.n{i % 2}", (4, 4, 32, 32), seed=48).to(dev))
            losses.append(float(tr.training_step(b, i, num_unet_denoising_steps=1, t=t)))
        assert tr.global_step == 6 and tr.skipped_steps == 0, (tr.global_step, tr.skipped_steps)
        return losses, [a.flat_p.clone() for a in tr.arenas], tr
    l0, p0, _ = run(False)
    l1, p1, tr = run(True)
    states = [[e.get("state") for e in g.entries.values()] for g in tr.graph_segments]
    assert all(st == ["graph"] for st in states), states                    # every segment was captured (and then replayed)
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    assert len(set(l0)) > 2                                                  # the optimizer really moved things between micro-batches


def test_graph_replay_is_not_reentered_while_its_buffers_are_held(dev):
    """batch_student_steps=False calls the student once per denoising step with ONE signature before any backward runs.  A replayed
    hipGraph returns the same output / saved-activation buffers every time, so the second and third call must not replay while the
    first call's autograd node still holds them (graphs.GraphedSegment.busy / claim): they run eagerly, and the losses and parameter
    trajectories equal the graph-free trainer bit for bit.  Extra-step timesteps / noise are presampled (the teacher then runs eagerly)."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.trainer import DistillTrainer, LossScaler

    def run(use_graphs):
        tr, _, _ = trainer_setup(dev, accum=1, ffn_lora=True)
        tr.scaler = LossScaler(init_scale=2.0 ** 6)
        for ad in (a for d in tr.ldm.model.ffn_lora.active("unet_distill").values() for a in d.values()):
            ad.p = 0.0
        if use_graphs:
            tr2 = DistillTrainer(tr.ldm, tr.id2ada, tr.text_encoder, accumulate_grad_batches=1, warm_up_steps=0,
                                 loss_scaler=LossScaler(init_scale=2.0 ** 6), use_graphs=True)
            tr.reducer.remove()
            tr = tr2
        tr.ldm.batch_student_steps = False
        for g in tr.optimizer.param_groups:
            g["lr"] = 1e-4
        tr.learning_rate = 1e-4
        t = torch.tensor([760, 850], device=dev)                  # three denoising steps: the micro-batch keeps ceil(4 / 3) = 2 instances
        pre = [(rng.synth_input(f"re.rel{j}", (2,), seed=49).abs().clamp(0, 1).to(dev), rng.synth_input(f"re.n{j}", (2, 4, 32, 32), seed=49).to(dev))
               for j in range(2)]
        losses = []
        try:
            for i in range(4):
                b = dict(x_start=rng.synth_input(f"dp.x{i % 2}", (4, 4, 32, 32), seed=48).to(dev), face_id_embs=rng.synth_input(f"dp.id{i % 2}", (4, 512), seed=48).to(dev),
                         fg_mask=torch.ones(4, 1, 32, 32, device=dev), noise=rng.synth_input(f"dp.n{i % 2}", (4, 4, 32, 32), seed=48).to(dev))
                losses.append(float(tr.training_step(b, i, num_unet_denoising_steps=3, t=t, presampled=pre)))
        finally:
            tr.ldm.batch_student_steps = True
        return losses, [a.flat_p.clone() for a in tr.arenas] + [torch.tensor([tr.global_step, tr.skipped_steps])], tr
    l0, p0, _ = run(False)
    l1, p1, tr = run(True)
    assert any(e.get("state") == "graph" for g in tr.graph_segments for e in g.entries.values())     # the student segment was captured
    assert l0 == l1, (l0, l1)
    assert int(p0[-1][0]) >= 2, p0[-1]                                        # optimizer steps really happened (identically on both sides, below)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def test_iteration_scheduler_and_normal_recon_iteration(dev):
    """The reference's iteration typing (ddpm.py:451-470) and the do_normal_recon iteration (ddpm.py:2296-2352, 2593-2883) on the HIP
    path at reduced width.  With unet_distill_iter_gap = 2 (v1-distill-arc2face-ada.yaml:28) non-compositional micro-batches
    alternate normal recon / U-Net distillation (non_comp_iters_count counts micro-batches); a recon micro-batch
    (two denoising steps with CFG against the null prompt, class-prompt passes, captured attention for the subject-attention
    suppression, the decoded x0 handed to the face pipeline) gives a finite loss with its monitors and moves the parameters."""
    from adaface_dev_amd import rng
    tr, _, _ = trainer_setup(dev, accum=2, faces=True)
    tr.ldm.uncond_context = (rng.synth_input("s2.uncond", (1, 77, 128), seed=46).to(dev), [""], {})
    tr.unet_distill_iter_gap = 2
    kinds = []
    p0 = tr.arenas[0].flat_p.clone()
    for i in range(8):
        b = dict(x_start=rng.synth_input(f"dp.x{i % 2}", (4, 4, 32, 32), seed=48).to(dev), face_id_embs=rng.synth_input(f"dp.id{i % 2}", (4, 512), seed=48).to(dev),
                 fg_mask=torch.ones(4, 1, 32, 32, device=dev))
        kw = dict(on_pure_noise=(i == 4)) if i % 2 == 0 else {}             # even micro-batches are the recon ones (asserted below)
        l = tr.training_step(b, i, **kw)
        kinds.append(tr.last_iter_type)
        assert torch.isfinite(l), (i, kinds)
        if tr.last_iter_type == "normal_recon":
            mon = tr.mon_loss_dict
            for k in ("pred_l2", "loss_recon_cls", "normal_recon_total", "recon_face_images_on_image_frac"):
                assert f"train/{k}" in mon, (k, sorted(mon))
    assert kinds == ["normal_recon", "unet_distill"] * 4, kinds
    assert tr.global_step + tr.skipped_steps == 4 and tr.global_step >= 2
    assert float((tr.arenas[0].flat_p - p0).abs().sum()) > 0


def test_scratch_of_a_captured_launch_lives_in_the_graph_pool(dev):
    """ops._grow_scratch: a launch recorded into a hipGraph must not be handed the growable per-device scratch (a later, larger eager
    call frees it and the replay would write through a stale pointer): under capture the scratch comes from the capturing graph's pool."""
    from adaface_dev_amd import ops
    cache = {}
    a = ops._grow_scratch(cache, dev, 1024, torch.uint8, floor=4096)
    assert a.numel() == 4096 and ops._grow_scratch(cache, dev, 2048, torch.uint8, floor=4096) is a
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        inside = ops._grow_scratch(cache, dev, 1024, torch.uint8, floor=4096)
        inside.zero_()
    assert inside.data_ptr() != a.data_ptr() and inside.numel() == 1024
    b = ops._grow_scratch(cache, dev, 1 << 20, torch.uint8, floor=4096)                 # growth replaces only the eager buffer
    assert b.numel() == 1 << 20 and cache[dev] is b
    g.replay()
    torch.cuda.synchronize()


def test_full_size_distill_loss_gradient_of_bs4_equals_its_bs1_slices(dev):
    """BASELINE configs[2] at FULL size (SD-1.5 student + teacher, 64x64 latents, 97 context tokens, bs 4, FFN adapters on): the
    distillation loss is finite, and -- a size-independent property, the oracle being far too slow here -- the context gradient
    of every sample of the bs-4 micro-batch equals the gradient of that sample run alone (x 1/4: the loss is a batch mean;
    GroupNorm, attention and the masked MSE are per sample).  Different GEMM tiles / split-K serve M = 4 x 4096 and 1 x 4096."""
    from adaface_dev_amd import SD15_UNET_CONFIG, rng
    from adaface_dev_amd.adaface.unet_teachers import Arc2FaceTeacher
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    ld = LatentDiffusion(SD15_UNET_CONFIG)
    rng.load_synth_weights(ld.model.diffusion_model, seed=0)
    teacher = UNetModel(**SD15_UNET_CONFIG)
    rng.load_synth_weights(teacher, seed=1)
    ld = ld.to(dev)
    for p in ld.model.diffusion_model.parameters():
        p.requires_grad_(False)
    ld.unet_teacher = Arc2FaceTeacher(teacher.to(dev))
    lora = ld.model.set_up_ffn_loras(lora_dropout=0.0)
    with torch.no_grad():
        for n, p in lora.named_parameters():
            if "lora_B" in n:
                p.copy_(rng.synth_input(n, p.shape, seed=7, scale=0.02))
            elif "lora_A" in n:
                p.copy_(rng.synth_input(n, p.shape, seed=7, scale=p[0].numel() ** -0.5))
    B = 4
    x0 = rng.synth_input("fs.x0", (B, 4, 64, 64), seed=21).to(dev)
    noise = rng.synth_input("fs.noise", (B, 4, 64, 64), seed=21).to(dev)
    t = torch.tensor([760, 850, 800, 720], device=dev)
    sctx = rng.synth_input("fs.sctx", (B, 97, 768), seed=21).to(dev)
    tctx = rng.synth_input("fs.tctx", (B, 21, 768), seed=21).to(dev)
    fg = (rng.synth_input("fs.fg", (B, 1, 64, 64), seed=21) > -0.3).float().to(dev)

    def run(sl):
        sg = sctx[sl].clone().requires_grad_(True)
        n = sg.shape[0]
        loss = ld.calc_unet_distill_loss(x0[sl], noise[sl], (sg, ["a"] * n, {}), tctx[sl], None, fg[sl], 1, t=t[sl])
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in lora.named_parameters() if "unet_distill" in k and p.grad is not None}
        for p in lora.parameters():
            p.grad = None
        return float(loss), sg.grad.detach().float(), grads
    l4, g4, w4 = run(slice(0, 4))
    assert np.isfinite(l4) and torch.isfinite(g4).all() and len(w4) == 18 and all(torch.isfinite(g).all() for g in w4.values())
    l1s, wsum = [], None
    for i in range(B):
        l1, g1, w1 = run(slice(i, i + 1))
        l1s.append(l1)
        e = rel_l2((4 * g4[i]).cpu().numpy(), g1[0].cpu().numpy())
        print(f"sample {i}: loss {l1:.5f}, dcontext bs4-slice vs bs1 rel-L2 {e:.2e}")
        assert e < 1.6e-2          # two fp16 passes against each other: each is within GRAD_TOL = 8e-3 of exact (tests/test_hip_unet.py)
        wsum = w1 if wsum is None else {k: wsum[k] + w1[k] for k in w1}
    assert abs(l4 - float(np.mean(l1s))) < 2e-3 * abs(l4)
    # round 6: an ORACLE-side value at full size too (the round-5 review: the full-size training tests were property checks only).  The loss of sample 0
    # run alone -- one student pass with the FFN adapters + one teacher pass of the full SD-1.5 U-Net in fp32 on the CPU (oracle/train_oracle.py
    # unet_distill_loss over oracle/unet_oracle.py; ~10 s) -- against the bs-1 HIP value above.  Chained fp16 U-Nets: 5e-3 as at reduced width.
    from oracle import diffusion_oracle as D
    from oracle import train_oracle as T
    from oracle import unet_oracle as O
    with torch.no_grad():
        sd_s = {k: v.detach().float().cpu() for k, v in ld.model.diffusion_model.state_dict().items()}
        sd_t = {k: v.detach().float().cpu() for k, v in teacher.state_dict().items()}
        ffn = {}
        for bi, ads in lora.active("unet_distill").items():
            ffn[f"output_blocks.{bi}.0."] = {key: (ad.lora_A.detach().float().cpu(), ad.lora_B.detach().float().cpu(), ad.lora_magnitude_vector.detach().float().cpu(),
                                                   ad.scaling, None) for key, ad in ads.items()}
        tabs = D.register_schedule(D.make_beta_schedule_linear())
        c = lambda v: v[0:1].detach().float().cpu()
        ref = T.unet_distill_loss(lambda x, tt, cc: O.unet_forward(sd_s, SD15_UNET_CONFIG, x, tt, cc, {"res_hidden_states_gradscale": 0.5, "ffn_lora": ffn}),
                                  lambda x, tt, cc: O.unet_forward(sd_t, SD15_UNET_CONFIG, x, tt, cc, {}),
                                  tabs, c(x0), c(noise), t[0:1].cpu(), c(sctx), c(tctx), c(fg), 1, [])
    el = abs(l1s[0] - float(ref)) / abs(float(ref))
    print(f"full size, sample 0: distillation loss {l1s[0]:.5f} (HIP) vs {float(ref):.5f} (CPU oracle), rel {el:.2e}")
    assert el < 5e-3
    for k in w4:                                                      # adapter weight gradients add over the samples
        e = rel_l2(w4[k].float().cpu().numpy(), (wsum[k] / 4).float().cpu().numpy())
        assert e < 1.5e-2, (k, e)


def test_full_size_bs4_train_step_of_the_bench_leg(dev):
    """The micro-batch bench.py times (BASELINE configs[2]: full-size student + teacher + 3 CLIP-L encoders, bs 4, 97 tokens, FFN
    adapters, accumulate 2, CAdamW) through its own construction code: two accumulation windows, finite losses, optimizer steps
    taken, none skipped, graphs captured."""
    import argparse
    import bench
    ns = argparse.Namespace(batch=4, no_ffn_lora=False, no_train_graphs=False, train_steps=4, train_warmup=2, no_roofline=True, distill_only=True)
    out = bench.run_train(ns, (1, 0, 0, False), dev, stage=1)
    cfg = out["config"]
    assert cfg["finite"] and cfg["skipped_steps"] == 0 and cfg["optimizer_steps"] == 3, cfg
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["value"] > 0
    assert any("captured" in s and not s.endswith(" 0 captured") for s in cfg["hipgraph_segments"]), cfg["hipgraph_segments"]


def test_full_size_stage1_iteration_mix_and_stage2_step_of_the_bench_legs(dev):
    """BASELINE configs[2] with the reference's iteration mix (normal recon / U-Net distillation alternating per optimizer step) and
    configs[4] (Stage-2 compositional distillation, bs 3, dual U-Net, the whole loss assembly with decoded x0 predictions) at FULL size
    through bench.py's own construction code: finite losses, optimizer steps taken, both iteration types met."""
    import argparse
    import bench
    ns = argparse.Namespace(batch=4, no_ffn_lora=False, no_train_graphs=False, train_steps=4, train_warmup=4, no_roofline=True, distill_only=False)
    out = bench.run_train(ns, (1, 0, 0, False), dev, stage=1)
    cfg = out["config"]
    assert cfg["finite"] and cfg["optimizer_steps"] + cfg["skipped_steps"] == 8, cfg
    assert cfg["optimizer_steps"] >= 6 and set(cfg["per_iteration_type"]) == {"normal_recon", "unet_distill"}, cfg
    out2 = bench.run_train(ns, (1, 0, 0, False), dev, stage=2)
    cfg2 = out2["config"]
    assert cfg2["finite"] and out2["value"] > 0 and cfg2["optimizer_steps"] >= 3, cfg2


def test_bench_train_leg_on_two_gpus_over_rccl(dev):
    """`bench.py --gpus 2 --mode train`: the launcher parent starts two ranks (one per GPU) under torch.distributed.run, RCCL with more
    than one rank -- start-up broadcast, bucketed gradient all-reduces from the backward hooks, overflow all-reduce.  Skips on boxes
    with a single GPU (every box this build has had); the first multi-GPU box exercises the N > 1 exchange."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train", "--train-steps", "4", "--train-warmup", "4",
                        "--no-roofline", "--distill-only"], env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["finite"] and out["config"]["parallelism"].startswith("dp2"), out
    assert out["value"] > 0 and out["config"]["optimizer_steps"] >= 3


def test_bench_train_leg_under_the_launcher_with_rccl_and_graph_replay(dev):
    """`bench.py --mode train` as the driver starts it for N > 1 -- under torch.distributed.run, backend nccl (= RCCL), here with one
    rank: process-group init, start-up broadcast, the gradient buckets' all-reduces issued from the backward hooks WHILE the student /
    teacher segments are captured into hipGraphs and then replayed, overflow all-reduce, optimizer steps; one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29623", os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "train", "--train-steps", "6",
                        "--train-warmup", "6", "--no-roofline", "--distill-only"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["unit"] == "images/s" and out["value"] > 0
    assert cfg["finite"] and cfg["skipped_steps"] == 0 and cfg["optimizer_steps"] == 6
    assert sorted(cfg["hipgraph_segments"]) == ["student.backward: 2 captured", "student.forward: 2 captured", "teacher.multistep: 3 captured"]


def test_guided_teacher_graph_replay_follows_the_per_call_guidance_scale(dev):
    """UNetTeacher with classifier-free guidance under graph replay (the Stage-2 priming U-Net): the guidance scale is drawn on the
    host per call and enters the captured segment as a 0-dim device tensor, so replays must follow it.  One denoising step (nothing
    random inside): eager, captured and replayed calls equal a teacher without graphs that is fed the same draws; with 2 steps the
    replays draw fresh timesteps / noise every time."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
    from adaface_dev_amd.graphs import GraphedSegment
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    ucfg = dict(CFG, context_dim=128)
    unet = UNetModel(**ucfg)
    rng.load_synth_weights(unet, seed=42)
    ld = LatentDiffusion(ucfg).to(dev)
    unet = unet.to(dev)
    plain = UNetTeacher(unet, cfg_scale_range=(2, 4), p_uses_cfg=1.0)
    graphed = UNetTeacher(unet, cfg_scale_range=(2, 4), p_uses_cfg=1.0)
    graphed.graphs = GraphedSegment("priming.multistep")
    to = lambda name, shape: rng.synth_input(name, shape, seed=47).to(dev)
    pos, neg = to("tg.pos", (2, 20, 128)), to("tg.neg", (1, 20, 128))
    t = torch.tensor([800, 720], device=dev)
    scales = []
    for call in range(5):
        x0, noise = to(f"tg.x{call}", (2, 4, 32, 32)), to(f"tg.n{call}", (2, 4, 32, 32))
        outs = []
        for teacher in (plain, graphed):
            np.random.seed(100 + call)                                   # the same two host draws (uses_cfg, cfg_scale) for both
            preds, xs, _, _ = teacher(ld, x0, noise, t, pos, negative_context=neg, num_denoising_steps=1)
            outs.append((preds[0].clone(), xs[1].clone(), teacher.cfg_scale))
        scales.append(outs[0][2])
        assert outs[0][2] == outs[1][2] and torch.isfinite(outs[1][0]).all()
        assert rel_l2(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy()) < 1e-5, call       # (cfg - 1) in fp32 vs double: last-bit differences
        assert rel_l2(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy()) < 1e-5, call
    assert len(set(scales)) == 5 and [e["state"] for e in graphed.graphs.entries.values()] == ["graph"]
    seen = []
    for call in range(4):                                                # 2 steps: the second step's timestep / noise are drawn inside the graph
        _, xs, ns, ts = graphed(ld, to("tg.x0", (2, 4, 32, 32)), to("tg.n0", (2, 4, 32, 32)), t, pos, negative_context=neg, num_denoising_steps=2)
        assert torch.isfinite(xs[2]).all() and (ts[1] < t).all()
        seen.append(ns[1].clone())
    assert not torch.equal(seen[2], seen[3]) and not torch.equal(seen[1], seen[2])
