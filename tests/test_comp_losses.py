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
