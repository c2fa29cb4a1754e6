"""Stage-2 U-Net pass on a real MI355X (`pytest -m gpu`): explicit cross-attention kernels (csrc/af_xattn_explicit.hip) against
fp32 torch, and the capture graph (activations of layers 22-24 captured WITH gradients, score rewrites of
adaface/diffusers_attn_lora_capture.py:108-133) against autograd through the CPU oracle -- whose explicit attention is pinned on the
REFERENCE's own scaled_dot_product_attention (tests/golden/sdpa.npz, tests/test_host_orchestration.py)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from test_hip_unet import GPU_TINY_CONFIG, GRAD_TOL, NET_TOL, _build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,N,L,heads,d", [(2, 131, 77, 8, 40), (1, 64, 97, 4, 8), (3, 50, 1, 2, 16), (1, 256, 128, 8, 160)])
def test_explicit_attention_kernels_vs_torch(dev, B, N, L, heads, d):
    """scores, softmax + PV, and every backward product (dscore with and without a gradient on the probabilities, dq, dk, dv) against
    fp32 torch on the same fp16 operands; operands are column slices of wider buffers (leading dimension > C)."""
    from adaface_dev_amd import ops, rng
    C = heads * d
    wide = lambda name, rows: (rng.synth_input(name, (rows, C + 16), seed=31).half().to(dev))[:, 8:8 + C]
    q, k, v, do = wide("xa.q", B * N), wide("xa.k", B * L), wide("xa.v", B * L), wide("xa.do", B * N)
    scale = d ** -0.5
    kw = dict(B=B, Nq=N, L=L, heads=heads, d=d)
    split = lambda t, n: t.float().reshape(B, n, heads, d).permute(0, 2, 1, 3)
    qf, kf, vf, dof = split(q, N), split(k, L), split(v, L), split(do, N)
    score = ops.xattn_scores(q, k, scale=scale, **kw)
    ref_score = qf @ kf.transpose(-1, -2) * scale
    assert rel_l2(score.cpu().numpy(), ref_score.cpu().numpy()) < 1e-6
    prob, o = ops.xattn_softmax_pv(score, v, **kw)
    ref_prob = ref_score.softmax(-1)
    ref_o = (ref_prob @ vf).permute(0, 2, 1, 3).reshape(B * N, C)
    assert rel_l2(prob.cpu().numpy(), ref_prob.cpu().numpy()) < 1e-6
    assert rel_l2(o.float().cpu().numpy(), ref_o.cpu().numpy()) < 6e-4                       # one fp16 rounding of the output
    dp_ext = rng.synth_input("xa.dp", (B, heads, N, L), seed=31).to(dev)
    for ext in (None, dp_ext):
        dP = dof @ vf.transpose(-1, -2) + (0 if ext is None else ext)
        ref_ds = ref_prob * (dP - (ref_prob * dP).sum(-1, keepdim=True))
        ds = ops.xattn_softmax_pv_bwd(prob, v, do, ext, **kw)
        assert rel_l2(ds.cpu().numpy(), ref_ds.cpu().numpy()) < 2e-5, ext is None
    merge = lambda t, n: t.permute(0, 2, 1, 3).reshape(B * n, C)
    dq = ops.xattn_rowmix(ref_ds.contiguous(), k, scale, **kw)
    assert rel_l2(dq.float().cpu().numpy(), merge(ref_ds @ kf * scale, N).cpu().numpy()) < 6e-4
    dk = ops.xattn_colmix(ref_ds.contiguous(), q, scale, **kw)
    assert rel_l2(dk.float().cpu().numpy(), merge(ref_ds.transpose(-1, -2) @ qf * scale, L).cpu().numpy()) < 6e-4
    dv = ops.xattn_colmix(prob, do, 1.0, **kw)
    assert rel_l2(dv.float().cpu().numpy(), merge(ref_prob.transpose(-1, -2) @ dof, L).cpu().numpy()) < 6e-4


def _loss_on_captures(eps, acts, w):
    """A scalar that touches eps and every captured tensor of every layer with fixed random weights (so each gradient path is exercised)."""
    tot = (eps * w["eps"]).sum()
    for key in ("outfeat", "attn", "attnscore", "q", "q2", "k", "v", "attn_out"):
        for li in (22, 23, 24):
            tot = tot + (acts[key][li].float() * w[f"{key}{li}"]).sum() * w["gain"][key]
    return tot


CASES = [dict(name="capture_grad"), dict(name="normalize", normalize=True), dict(name="mix", mix=True), dict(name="capture_grad_masked", mask=True)]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_unet_capture_with_gradients_and_score_rewrites_vs_oracle(dev, case):
    """Reduced-width U-Net (model_channels 64: the captured layers have 8 heads of 8), latent 16x16, 20 context tokens: eps, all eight
    captured tensors of layers 22-24, and the gradients of a loss over eps AND the captures w.r.t. x, the context and (normalize) the
    three learnable scale factors -- HIP capture graph vs torch autograd through the oracle, for the plain captured pass, the
    subject-token normalisation, the SC/MC score mixing, and with the self-attention key mask."""
    from adaface_dev_amd import rng
    from oracle import unet_oracle as O
    m, sd = _build(GPU_TINY_CONFIG, 11, dev)
    B, Hh, T = 2, 16, 20
    x = rng.synth_input("cg.x", (B, 4, Hh, Hh), seed=12)
    ctx = rng.synth_input("cg.ctx", (B, T, 64), seed=12)
    t = torch.tensor([30, 620])
    mask = None
    if case.get("mask"):
        mask = torch.ones(B, 1, Hh, Hh)
        mask[0, :, :, :5] = 0
    subj = (torch.tensor([0, 0, 0, 1, 1]), torch.tensor([4, 5, 6, 4, 5]))
    factors_ref = [torch.tensor(v, requires_grad=True) for v in (0.8, 0.65, 1.1)]
    factors_hip = [torch.tensor(v, device=dev, requires_grad=True) for v in (0.8, 0.65, 1.1)]
    flags = dict(capture_ca_activations=True, normalize_cross_attn=bool(case.get("normalize")), mix_attn_mats_in_batch=bool(case.get("mix")),
                 res_hidden_states_gradscale=0.5)
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    ei_ref = dict(flags, img_mask=mask, subj_indices=subj, cross_attn_scale_factors=factors_ref)
    ref = O.unet_forward(sd, GPU_TINY_CONFIG, xr, t, cr, ei_ref)
    racts = ei_ref["ca_layers_activations"]
    w = {"eps": rng.synth_input("cg.w.eps", tuple(ref.shape), seed=12),
         "gain": dict(outfeat=0.05, attn=3.0, attnscore=0.05, q=0.05, q2=0.02, k=0.05, v=0.05, attn_out=0.05)}
    for key in racts:
        for li in (22, 23, 24):
            w[f"{key}{li}"] = rng.synth_input(f"cg.w.{key}{li}", tuple(racts[key][li].shape), seed=12)
    _loss_on_captures(ref, racts, w).backward()

    xg, cg = x.clone().to(dev).requires_grad_(True), ctx.clone().to(dev).requires_grad_(True)
    ei = dict(flags, img_mask=None if mask is None else mask.to(dev), subj_indices=(subj[0].to(dev), subj[1].to(dev)),
              _cross_attn_scale_factors=factors_hip)
    eps = m(xg, t.to(dev), cg, extra_info=ei)
    acts = ei["ca_layers_activations"]
    assert rel_l2(eps.detach().cpu().numpy(), ref.detach().numpy()) < NET_TOL
    for key in ("outfeat", "attn", "attnscore", "q", "q2", "k", "v", "attn_out"):
        assert sorted(acts[key].keys()) == [22, 23, 24]
        for li in (22, 23, 24):
            assert tuple(acts[key][li].shape) == tuple(racts[key][li].shape), (key, li)
            assert rel_l2(acts[key][li].detach().float().cpu().numpy(), racts[key][li].detach().numpy()) < 2 * NET_TOL, (key, li)
    wd = {k: (v if isinstance(v, dict) else v.to(dev)) for k, v in w.items()}
    # fp16 activation gradients: scale the loss like the trainer's LossScaler does, unscale the results
    S = 256.0
    (_loss_on_captures(eps, acts, wd) * S).backward()
    ex = rel_l2((xg.grad / S).cpu().numpy(), xr.grad.numpy())
    ec = rel_l2((cg.grad / S).cpu().numpy(), cr.grad.numpy())
    assert ex < GRAD_TOL and ec < GRAD_TOL, (ex, ec)
    if case.get("normalize"):
        # one scalar = a signed sum over 5 subject columns x 8 heads x 256 pixels of dscore * centred score (x10): the fp16 noise of
        # the incoming gradients does not cancel the way the terms do, so the bound is looser than for the tensors (measured 6 %)
        for fh, fr in zip(factors_hip, factors_ref):
            assert abs(float(fh.grad) / S - float(fr.grad)) < 0.12 * max(1.0, abs(float(fr.grad))), (float(fh.grad) / S, float(fr.grad))
    if case.get("mix"):
        # both halves of the batch attend with the same (averaged) scores
        for li in (22, 23, 24):
            assert torch.equal(acts["attnscore"][li][0], acts["attnscore"][li][1])


def test_unet_score_rewrite_without_capture_and_without_grad(dev):
    """The no-grad instances of a subject-compos batch (ddpm.py:1641-1660) run the rewritten attention too; captures only on request."""
    from adaface_dev_amd import rng
    from oracle import unet_oracle as O
    m, sd = _build(GPU_TINY_CONFIG, 11, dev)
    x = rng.synth_input("cg.x", (1, 4, 16, 16), seed=12)
    ctx = rng.synth_input("cg.ctx", (1, 20, 64), seed=12)
    t = torch.tensor([400])
    subj = (torch.tensor([0, 0]), torch.tensor([4, 5]))
    ei = dict(normalize_cross_attn=True, subj_indices=(subj[0].to(dev), subj[1].to(dev)))
    with torch.no_grad():
        eps = m(x.to(dev), t.to(dev), ctx.to(dev), extra_info=ei)
    ref = O.unet_forward(sd, GPU_TINY_CONFIG, x, t, ctx, dict(normalize_cross_attn=True, subj_indices=subj))
    assert "ca_layers_activations" not in ei and not eps.requires_grad
    assert rel_l2(eps.cpu().numpy(), ref.numpy()) < NET_TOL
    plain = O.unet_forward(sd, GPU_TINY_CONFIG, x, t, ctx, {})
    assert rel_l2(ref.numpy(), plain.numpy()) > 1e-3                       # the rewrite really changes the result


@pytest.mark.parametrize("q_updates", [False, True])
def test_unet_attention_dora_training_gradients_vs_oracle(dev, q_updates):
    """Trainable attention DoRA adapters on to_q / to_k / to_v / to_out.0 of the three captured cross-attention layers
    (diffusers_attn_lora_capture.py:171-181, 239-249, 280-288, 328-331): the q adapter feeds the captured query2 only (unless
    q_lora_updates_query), the others key / value / output.  eps, q / q2 captures, d/dcontext and the gradients of all 36 adapter
    tensors against autograd through the oracle with the same adapters (dropout 0)."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import unet_oracle as O
    ld = LatentDiffusion(GPU_TINY_CONFIG)
    rng.load_synth_weights(ld.model.diffusion_model, seed=11)
    sd = {k: v.detach().clone() for k, v in ld.model.diffusion_model.state_dict().items()}
    ld = ld.to(dev)
    for p in ld.model.diffusion_model.parameters():
        p.requires_grad_(False)
    lora = ld.model.set_up_attn_loras(lora_rank=16, lora_scale_down=8, lora_dropout=0.0, q_lora_updates_query=q_updates)
    with torch.no_grad():
        for n, p in lora.named_parameters():
            if "lora_B" in n:
                p.copy_(rng.synth_input(n, p.shape, seed=83, scale=0.3))
            if "magnitude" in n:
                p.mul_(1.0 + 0.1 * rng.synth_input(n, p.shape, seed=83).to(dev).abs())
    B, Hh, T = 2, 16, 20
    x = rng.synth_input("cg.x", (B, 4, Hh, Hh), seed=12)
    ctx = rng.synth_input("cg.ctx", (B, T, 64), seed=12)
    t = torch.tensor([30, 620])
    # oracle side: the same adapters as fp32 leaf tensors
    ref_lora, leaves = {}, {}
    for bi, d in lora.active().items():
        li = 22 + (bi - 9)
        ref_lora[li] = {}
        for name, ad in d.items():
            ts = [getattr(ad, pn).detach().cpu().float().clone().requires_grad_(True) for pn in ("lora_A", "lora_B", "lora_magnitude_vector")]
            ref_lora[li][name] = (ts[0], ts[1], ts[2], ad.scaling)
            leaves[(bi, name)] = ts
    cr = ctx.clone().requires_grad_(True)
    ei_ref = dict(capture_ca_activations=True, attn_lora=ref_lora, q_lora_updates_query=q_updates)
    ref = O.unet_forward(sd, GPU_TINY_CONFIG, x, t, cr, ei_ref)
    racts = ei_ref["ca_layers_activations"]
    w_eps = rng.synth_input("al.w.eps", tuple(ref.shape), seed=12)
    w_q2 = {li: rng.synth_input(f"al.w.q2{li}", tuple(racts["q2"][li].shape), seed=12) for li in (22, 23, 24)}
    loss_of = lambda e, a, we, wq: (e * we).sum() + sum((a["q2"][li].float() * wq[li]).sum() * 0.05 for li in (22, 23, 24))
    loss_of(ref, racts, w_eps, w_q2).backward()

    cg = ctx.clone().to(dev).requires_grad_(True)
    ei = dict(capture_ca_activations=True)
    eps = ld.apply_model(x.to(dev), t.to(dev), (cg, ["a", "b"], ei), use_attn_lora=True)
    acts = ei["ca_layers_activations"]
    assert rel_l2(eps.detach().cpu().numpy(), ref.detach().numpy()) < NET_TOL
    for li in (22, 23, 24):
        assert rel_l2(acts["q2"][li].detach().cpu().numpy(), racts["q2"][li].detach().numpy()) < 2 * NET_TOL
        same = torch.equal(acts["q"][li], acts["q2"][li])
        assert same == q_updates                                   # query2 differs from query unless the adapter updates the query
    S = 256.0
    (loss_of(eps, acts, w_eps.to(dev), {k: v.to(dev) for k, v in w_q2.items()}) * S).backward()
    assert rel_l2((cg.grad / S).cpu().numpy(), cr.grad.numpy()) < GRAD_TOL
    worst = 0.0
    for bi, d in lora.active().items():
        for name, ad in d.items():
            for pn, leaf in zip(("lora_A", "lora_B", "lora_magnitude_vector"), leaves[(bi, name)]):
                g = getattr(ad, pn).grad
                assert g is not None, (bi, name, pn)
                err = rel_l2((g / S).cpu().numpy(), leaf.grad.numpy())
                worst = max(worst, err)
                assert err < 2 * GRAD_TOL, (bi, name, pn, err)
    print(f"attention DoRA: worst adapter-gradient rel-L2 {worst:.2e}")
