"""Stage-2 losses on captured activations (adaface_dev_amd.ldm.comp_losses) against values AND gradients written by the reference's
own functions (ldm/util.py: calc_sc_rep_attn_distill_loss, calc_subj_attn_cross_t_diff_loss, calc_attn_norm_loss,
calc_subj_masked_bg_suppress_loss, calc_dyn_loss_scale; tests/golden/gen_golden.py::gen_comp_losses imports them from
/root/reference).  Pure tensor bookkeeping: runs on the CPU."""
import os
import sys

import numpy as np
import torch

from conftest import GOLDEN, rel_l2

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _inputs():
    from gen_golden import comp_loss_inputs
    return comp_loss_inputs()


def _check_grads(g, acts, tag):
    for key in ("attn", "k", "v"):
        for li in (23, 24):
            want = g[f"{tag}.d{key}{li}"]
            got = acts[key][li].grad
            if want.size == 1 and want.reshape(-1)[0] == 0 and want.ndim == 1:
                assert got is None or float(got.abs().sum()) == 0.0, (tag, key, li)
            else:
                assert got is not None and rel_l2(got.numpy(), want) < 1e-5, (tag, key, li)


def test_rep_attn_distill_loss_vs_reference():
    from adaface_dev_amd.ldm import comp_losses as CL
    g = np.load(os.path.join(GOLDEN, "comp_losses.npz"))
    for pct in (0.05, 0.15, 0.22, 0.3):
        acts, future, s1, s2, em, pm, fg = _inputs()
        ls = CL.calc_sc_rep_attn_distill_loss(acts, s1, em, pm, pct, FG_THRES=0.1)
        want = g[f"rep{pct}.values"]
        assert np.allclose([float(v) for v in ls], want, rtol=1e-5, atol=1e-9), (pct, [float(v) for v in ls], want)
        if pct >= 0.1:
            sum(ls).backward()
            _check_grads(g, acts, f"rep{pct}")
            assert float(want[0]) > 0
        else:
            assert all(float(v) == 0 for v in ls)                   # face too small: no rep distillation


def test_cross_t_attn_norm_and_mb_suppress_losses_vs_reference():
    from adaface_dev_amd.ldm import comp_losses as CL
    g = np.load(os.path.join(GOLDEN, "comp_losses.npz"))
    acts, future, s1, s2, em, pm, fg = _inputs()
    l = CL.calc_subj_attn_cross_t_diff_loss(acts, future, s1)
    l.backward()
    assert abs(float(l) - float(g["crosst.value"])) < 1e-6 * abs(float(g["crosst.value"]))
    _check_grads(g, acts, "crosst")
    acts, future, s1, s2, em, pm, fg = _inputs()
    l = CL.calc_attn_norm_loss(acts["outfeat"], acts["attn"], s2, 1)
    l.backward()
    assert abs(float(l) - float(g["attnnorm.value"])) < 1e-6 * abs(float(g["attnnorm.value"]))
    _check_grads(g, acts, "attnnorm")
    acts, future, s1, s2, em, pm, fg = _inputs()
    sc_attn = {li: a.chunk(4)[1] for li, a in acts["attn"].items()}
    l = CL.calc_subj_masked_bg_suppress_loss(sc_attn, s1, 1, fg[:1])
    l.backward()
    assert float(g["mbsuppress.value"]) > 0 and abs(float(l) - float(g["mbsuppress.value"])) < 1e-6 * float(g["mbsuppress.value"])
    _check_grads(g, acts, "mbsuppress")
    assert float(CL.calc_subj_masked_bg_suppress_loss(sc_attn, s1, 1, torch.ones_like(fg[:1]))) == float(g["mbsuppress.allfg"]) == 0.0


def test_dyn_loss_scale_and_rep_distill_weighting():
    from adaface_dev_amd.ldm import comp_losses as CL
    g = np.load(os.path.join(GOLDEN, "comp_losses.npz"))
    got = [CL.calc_dyn_loss_scale(x, (0.20, 0.5), (0.25, 2), valid_scale_range=(0.05, 2)) for x in g["dyn.x"]]
    assert np.allclose(got, g["dyn.scale"], rtol=1e-12)
    # the weighting of ddpm.py:3557-3589 on the reference's five values
    v = g["rep0.22.values"]
    scale = CL.calc_dyn_loss_scale(0.22, (0.20, 0.5), (0.25, 2), valid_scale_range=(0.05, 2))
    want = ((v[0] + v[1] + v[3]) * 2 + v[2] * 5 + v[4] * 2) * scale
    assert abs(float(CL.comp_rep_distill_total(tuple(torch.tensor(x) for x in v), 0.22)) - want) < 1e-9
    assert float(CL.comp_rep_distill_total(tuple(torch.tensor(x) for x in v), 0.0)) == 0.0


def _preserve_inputs(scale, device="cpu"):
    from gen_golden import comp_preserve_inputs
    return comp_preserve_inputs(device=device, scale=scale)


def check_preserve_case(g, tag, kw, scale, device="cpu", tol=1e-5):
    """One case of comp_preserve.npz (written by the reference's calc_comp_subj_bg_preserve_loss with flow_model=None): value, every
    monitor entry the reference left in mon_loss_dict, gradients w.r.t. q2 / attn_out / outfeat of the three layers."""
    from adaface_dev_amd.ldm import comp_losses as CL
    acts, ssb, scb = _preserve_inputs(scale, device)
    mon = {}
    loss = CL.calc_comp_subj_bg_preserve_loss(mon, "train", torch.device(device), None, acts, ssb, scb, **kw)
    want = float(g[f"{tag}.loss"])
    assert abs(float(loss) - want) <= tol * max(abs(want), 1e-3), (tag, float(loss), want)
    keys = [k for k in g.files if k.startswith(f"{tag}.mon.")]
    assert len(keys) == len(mon) and len(keys) >= 17, (tag, sorted(mon), keys)
    for k in keys:
        name = k[len(tag) + 5:].replace("__", "/")
        assert abs(float(mon[name]) - float(g[k])) <= 10 * tol * max(abs(float(g[k])), 1e-3), (tag, name, float(mon[name]), float(g[k]))
    if loss.requires_grad:
        loss.backward()
    for key in ("q2", "attn_out", "outfeat"):
        for li in (22, 23, 24):
            wantg, got = g[f"{tag}.d{key}{li}"], acts[key][li].grad
            if wantg.ndim == 1 and wantg.size == 1:
                assert got is None or float(got.abs().sum()) == 0.0, (tag, key, li)
            else:
                assert got is not None and rel_l2(got.cpu().numpy(), wantg) < 10 * tol, (tag, key, li, rel_l2(got.cpu().numpy(), wantg))


def test_comp_subj_bg_preserve_loss_vs_reference():
    from gen_golden import PRESERVE_CASES
    g = np.load(os.path.join(GOLDEN, "comp_preserve.npz"))
    for tag, kw, scale in PRESERVE_CASES:
        check_preserve_case(g, tag, kw, scale)
    assert float(g["plain.loss"]) > 0 and float(g["discarded.loss"]) > 0
    assert float(g["discarded.mon.train__discarded_loss_ratio"]) > 0        # a min-loss past thres x max_scale was dropped


def test_flow_model_is_refused():
    from adaface_dev_amd.ldm import comp_losses as CL
    acts, ssb, scb = _preserve_inputs(0.5)
    try:
        CL.calc_comp_subj_bg_preserve_loss({}, "train", torch.device("cpu"), object(), acts, ssb, scb)
    except NotImplementedError as e:
        assert "flow" in str(e)
    else:
        raise AssertionError("a flow model must be refused")


def check_recon_and_suppress(g, device="cpu", tol=1e-5):
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm import comp_losses as CL
    from gen_golden import comp_loss_inputs
    subj4 = (torch.arange(4, device=device).repeat_interleave(4), torch.tensor([4, 5, 6, 7], device=device).repeat(4))
    for tag, pure, with_cls, w in (("image", False, True, torch.tensor([1.0, 0.1, 1.0, 1.0])), ("noise", True, True, torch.ones(4)),
                                   ("nocls", False, False, torch.ones(4))):
        eps = rng.synth_input("cp.eps", (4, 4, 16, 16), seed=73).to(device).requires_grad_(True)
        gt, cls = rng.synth_input("cp.gt", (4, 4, 16, 16), seed=73).to(device), rng.synth_input("cp.cls", (4, 4, 16, 16), seed=73).to(device)
        acts, future, s1, s2, em, pm, fg = comp_loss_inputs(device)
        ls = CL.calc_recon_and_suppress_losses(gt, eps, cls if with_cls else None, w.to(device), acts, subj4, None, fg, 0.025, 4, pure)
        assert np.allclose([float(l) for l in ls], g[f"recon_{tag}.values"], rtol=10 * tol, atol=1e-8), (tag, [float(l) for l in ls])
        sum(l for l in ls if torch.is_tensor(l) and l.requires_grad).backward()
        assert rel_l2(eps.grad.cpu().numpy(), g[f"recon_{tag}.deps"]) < 10 * tol
        assert rel_l2(acts["attn"][23].grad.cpu().numpy(), g[f"recon_{tag}.dattn23"]) < 10 * tol


def test_recon_and_suppress_losses_vs_reference():
    check_recon_and_suppress(np.load(os.path.join(GOLDEN, "comp_preserve.npz")))


def test_tensor_attached_memos_are_voided_by_in_place_writes():
    """The index split, the instance count and the host copy of a box mask ride on tensor objects (no host<->device wait when the same
    tensor comes back); each records the version counter it was made at, so an in-place write to the tensor voids it."""
    from adaface_dev_amd.ldm import comp_losses as CL
    from adaface_dev_amd.ldm.models.diffusion.ddpm_losses import box_mask, host_of
    b, n = torch.tensor([0, 0, 1, 1, 1]), torch.tensor([4, 5, 4, 5, 6])
    first = CL.split_indices_by_instance((b, n))
    assert CL.split_indices_by_instance((b, n)) is first and CL.count_instances(b) == 2
    b[4] = 2                                             # in place: three instances now
    again = CL.split_indices_by_instance((b, n))
    assert again is not first and len(again) == 3 and CL.count_instances(b) == 3
    n.add_(1)
    assert [t[1].tolist() for t in CL.split_indices_by_instance((b, n))] == [[5, 6], [5, 6], [7]]
    host = box_mask([(1, 1, 3, 3)], 1, 4, 4, torch.device("cpu"))
    assert float(host_of(host).sum()) == 4.0
    m = host.clone()                                     # stands for the device copy (on the CPU box_mask returns the host tensor itself)
    m.af_host = (host, m._version)
    assert host_of(m) is host
    m.mul_(0)                                            # the device mask changed: its host copy no longer describes it
    assert host_of(m) is m
