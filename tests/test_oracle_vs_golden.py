"""Pin the CPU oracle (oracle/*.py) against the reference's own outputs (tests/golden/*.npz,
made by tests/golden/gen_golden.py from /root/reference).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from adaface_dev_amd import SD15_UNET_CONFIG, TINY_UNET_CONFIG, rng
from oracle import diffusion_oracle as D
from oracle import unet_oracle as O

TOL = 1e-5  # fp32 restatement vs fp32 reference


def _sd_from_shapes(cfg, seed):
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import unet_param_shapes
    return rng.synth_state_dict(unet_param_shapes(cfg), seed=seed)


@pytest.fixture(scope="module")
def tiny():
    g = np.load(os.path.join(GOLDEN, "unet_tiny.npz"))
    sd = _sd_from_shapes(TINY_UNET_CONFIG, seed=1)
    x = rng.synth_input("tiny.x", (2, 4, 16, 16), seed=1)
    ctx = rng.synth_input("tiny.ctx", (2, 77, 64), seed=1)
    t = torch.tensor([10, 500], dtype=torch.int64)
    return g, sd, x, ctx, t


def test_tiny_unet_eps(tiny):
    g, sd, x, ctx, t = tiny
    eps = O.unet_forward(sd, TINY_UNET_CONFIG, x, t, ctx, {})
    assert rel_l2(eps.numpy(), g["eps"]) < TOL


def test_tiny_unet_grads(tiny):
    g, sd, x, ctx, t = tiny
    xg = x.clone().requires_grad_(True)
    cg = ctx.clone().requires_grad_(True)
    eps = O.unet_forward(sd, TINY_UNET_CONFIG, xg, t, cg, {})
    cot = rng.synth_input("tiny.cot", eps.shape, seed=1)
    (eps * cot).sum().backward()
    assert rel_l2(xg.grad.numpy(), g["grad_x"]) < 1e-4
    assert rel_l2(cg.grad.numpy(), g["grad_ctx"]) < 1e-4


def test_tiny_unet_mask_and_capture(tiny):
    g, sd, x, ctx, t = tiny
    mask = torch.ones(2, 1, 16, 16)
    mask[0, :, :, :5] = 0
    mask[1, :, 9:, :] = 0
    ei = {"img_mask": mask, "capture_ca_activations": True}
    eps = O.unet_forward(sd, TINY_UNET_CONFIG, x, t, ctx, ei)
    assert rel_l2(eps.numpy(), g["eps_masked"]) < TOL
    acts = ei["ca_layers_activations"]
    assert sorted(acts["attn"].keys()) == [22, 23, 24]
    assert rel_l2(acts["attn"][24].numpy(), g["cap_attn_24_full"].astype(np.float32)) < 2e-3  # fp16-stored
    assert rel_l2(acts["q"][24].numpy(), g["cap_q_24_full"]) < TOL
    assert rel_l2(acts["attn_out"][24].numpy(), g["cap_attn_out_24_full"]) < TOL
    for key in ("outfeat", "attn", "attnscore", "q", "attn_out"):
        for li in (22, 23, 24):
            assert tuple(acts[key][li].shape) == tuple(g[f"cap_{key}_{li}_shape"])
            f = acts[key][li].float().reshape(-1)
            assert abs(f.abs().mean().item() - g[f"cap_{key}_{li}"][1]) <= 1e-4 * max(1.0, g[f"cap_{key}_{li}"][1])


def test_full_unet_eps():
    path = os.path.join(GOLDEN, "unet_full.npz")
    g = np.load(path)
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import unet_param_shapes
    shapes = unet_param_shapes(SD15_UNET_CONFIG)
    assert sum(int(np.prod(s)) for _, s in shapes) == int(g["nparams"]) == 859520964   # SURVEY.md section 4 KAT
    assert len(shapes) == int(g["ntensors"]) == 686
    sd = rng.synth_state_dict(shapes, seed=0)
    x = rng.synth_input("full.x", (1, 4, 64, 64), seed=0)
    ctx = rng.synth_input("full.ctx", (1, 77, 768), seed=0)
    x.requires_grad_(True)
    ctx.requires_grad_(True)
    eps = O.unet_forward(sd, SD15_UNET_CONFIG, x, torch.tensor([500]), ctx, {})
    assert rel_l2(eps.detach().numpy(), g["eps"]) < 1e-4
    # the oracle's autograd against the reference module's at full size (what the full-size GPU backward test is held to)
    (eps * rng.synth_input("full.cot", (1, 4, 64, 64), seed=0)).sum().backward()
    assert rel_l2(x.grad.numpy(), g["grad_x"]) < 1e-4 and rel_l2(ctx.grad.numpy(), g["grad_ctx"]) < 1e-4


def test_schedule_tables():
    g = np.load(os.path.join(GOLDEN, "schedule.npz"))
    betas = D.make_beta_schedule_linear()
    assert np.array_equal(betas, g["betas"])
    tabs = D.register_schedule(betas)
    assert np.allclose(tabs["alphas_cumprod"], g["alphas_cumprod"].astype(np.float32), rtol=0, atol=0)
    ts = D.make_ddim_timesteps(50)
    assert np.array_equal(ts, g["ddim_timesteps"])
    assert list(ts[:3]) == [1, 21, 41] and ts[-1] == 981            # ddim.py:29-35
    sig, a, ap = D.make_ddim_sampling_parameters(tabs["alphas_cumprod"], ts, 0.0)
    assert np.array_equal(np.asarray(a, np.float32), g["ddim_alphas"])
    assert np.array_equal(np.asarray(ap, np.float32), g["ddim_alphas_prev"])
    # in-code known answers printed in fp16 at ddim.py:265-270
    assert np.allclose(a[:5], g["kat_alphas_first"], atol=3e-3)
    assert np.allclose(a[-5:], g["kat_alphas_last"], atol=2e-4)


def test_ddim_trajectory():
    g = np.load(os.path.join(GOLDEN, "ddim_step.npz"))
    tabs = D.register_schedule(D.make_beta_schedule_linear())
    ts = D.make_ddim_timesteps(50)
    _, a, ap = D.make_ddim_sampling_parameters(tabs["alphas_cumprod"], ts, 0.0)
    x = rng.synth_input("ddim.xT", (2, 4, 8, 8), seed=7)
    c = rng.synth_input("ddim.c", (2, 77, 16), seed=7)
    uc = rng.synth_input("ddim.uc", (2, 77, 16), seed=7)
    fake = lambda x, t, ctx: torch.tanh(x) * 0.7 + 0.05 * ctx.mean(dim=(1, 2)).reshape(-1, 1, 1, 1) + 1e-4 * t.reshape(-1, 1, 1, 1).float()
    scales = D.guide_scale_sequence(50, (4.0, 1.0))
    for i, step in enumerate(np.flip(ts)):
        index = 50 - i - 1
        t = torch.full((2,), int(step), dtype=torch.long)
        e = fake(torch.cat([x, x]), torch.cat([t, t]), torch.cat([c, uc]))
        e_c, e_u = e.chunk(2)
        e_t = D.cfg_combine(e_c, e_u, scales[i])
        x, _ = D.ddim_update(x, e_t, float(a[index]), float(ap[index]))
    assert rel_l2(x.numpy(), g["x_final"]) < 1e-5


def test_block_fixtures():
    """Leaf/blocks of the oracle against reference module outputs at reduced widths."""
    import torch.nn.functional as F
    g = np.load(os.path.join(GOLDEN, "blocks.npz"))
    assert rel_l2(O.timestep_embedding(torch.from_numpy(g["temb_t"]), 320).numpy(), g["temb_320"]) < 1e-6

    w, b = rng.synth_tensor("weight", (64,), 2), rng.synth_tensor("bias", (64,), 2)
    x = rng.synth_input("blk.gn.x", (2, 64, 8, 8), seed=2, scale=2.0) + 0.5
    assert rel_l2(O.group_norm(x, w, b, 1e-5, silu=True).numpy(), g["gn_silu"]) < TOL

    def sd_of(named_shapes, seed):
        return rng.synth_state_dict(named_shapes, seed)

    for tag, qd, cd, heads, dh, n, l in (("self", 64, None, 8, 8, 64, 64), ("cross", 64, 48, 8, 8, 64, 77),
                                         ("d40", 80, 96, 2, 40, 48, 77)):
        inner = heads * dh
        sd = sd_of([("to_q.weight", (inner, qd)), ("to_k.weight", (inner, cd or qd)), ("to_v.weight", (inner, cd or qd)),
                    ("to_out.0.weight", (qd, inner)), ("to_out.0.bias", (qd,))], 3)
        xq = rng.synth_input(f"blk.ca.{tag}.x", (2, n, qd), seed=3)
        cx = None if cd is None else rng.synth_input(f"blk.ca.{tag}.ctx", (2, l, cd), seed=3)
        assert rel_l2(O.cross_attention(sd, "", xq, cx, None, heads).numpy(), g[f"ca_{tag}"]) < TOL
        if cd is None:
            mask = torch.ones(2, 1, 8, 8)
            mask[0, :, :3] = 0
            mask[1, :, :, 6:] = 0
            assert rel_l2(O.cross_attention(sd, "", xq, None, mask, heads).numpy(), g[f"ca_{tag}_masked"]) < TOL
            assert rel_l2(O.cross_attention(sd, "", xq, None, torch.zeros(2, 1, 8, 8), heads).numpy(),
                          g[f"ca_{tag}_allmasked"]) < TOL
