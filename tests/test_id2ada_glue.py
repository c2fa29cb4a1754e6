"""The glue of the face -> prompt stack against fixtures written by the REFERENCE classes themselves
(tests/golden/gen_golden.py::gen_id2ada_glue: reference ``SubjBasisGenerator.forward`` / ``inverse_img_prompt_embs``,
``Arc2Face_ID2AdaPrompt.map_init_id_to_img_prompt_embs``, ``get_img_prompt_embs``, ``generate_adaface_embeddings``), with the CLIP text
transformers replaced on BOTH sides by the same stand-in (tests/standin.py::StandInCLIP; their arithmetic is pinned separately in
tests/test_clip_oracle.py / test_hip_clip.py).  With the stand-in in place the mirrors contain no kernel call: CPU test."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from standin import StandInCLIP

D, N_ID, N_SFX = 768, 16, 2


def _sbg():
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.arc2face_models import clip_text_config
    from adaface_dev_amd.adaface.subj_basis_generator import SubjBasisGenerator
    g = SubjBasisGenerator(dtype=torch.float32, num_id_vecs=N_ID, num_static_img_suffix_embs=N_SFX, output_dim=D,
                           clip_config=clip_text_config(hidden_size=64, num_attention_heads=2, num_hidden_layers=1, intermediate_size=64))
    g.prompt2token_proj = StandInCLIP(D, seed=73)
    with torch.no_grad():
        g.static_img_suffix_embs.copy_(rng.synth_input("glue.sfx", (1, N_SFX, D), seed=74))
        g.pad_embeddings = rng.synth_input("glue.pad", (77, D), seed=74)
    return g


def _id2ada():
    from adaface_dev_amd.adaface.arc2face_models import clip_text_config
    from adaface_dev_amd.adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
    a = Arc2Face_ID2AdaPrompt(clip_config=clip_text_config(hidden_size=64, num_attention_heads=2, num_hidden_layers=1, intermediate_size=64),
                              num_static_img_suffix_embs=N_SFX, out_id_embs_cfg_scale=0.8)
    a.text_to_image_prompt_encoder = StandInCLIP(D, seed=75)
    a.subj_basis_generator = _sbg()
    return a


@pytest.mark.parametrize("tag,cfg_scale,sfx", [("plain", 1.0, False), ("cfg07_sfx", 0.7, True), ("cfg13", 1.3, False)])
def test_subj_basis_generator_glue_vs_reference(tag, cfg_scale, sfx):
    from adaface_dev_amd import rng
    g = np.load(os.path.join(GOLDEN, "id2ada_glue.npz"))
    m = _sbg()
    x = rng.synth_input("glue.id2img", (2, N_ID, D), seed=74).requires_grad_(True)
    y = m(x, out_id_embs_cfg_scale=cfg_scale, is_face=True, enable_static_img_suffix_embs=sfx)
    assert tuple(y.shape) == g[f"sbg.{tag}.out"].shape
    assert rel_l2(y.detach().numpy(), g[f"sbg.{tag}.out"]) < 1e-6
    (y * rng.synth_input(f"glue.w.{tag}", tuple(y.shape), seed=74)).sum().backward()
    assert rel_l2(x.grad.numpy(), g[f"sbg.{tag}.dx"]) < 1e-5
    assert rel_l2(m.hidden_state_layer_weights.grad.numpy(), g[f"sbg.{tag}.dlayer_w"]) < 1e-5          # x5 gradient scaler included
    want = g[f"sbg.{tag}.dsfx"]
    got = m.static_img_suffix_embs.grad
    if want.size == 1:
        assert got is None or float(got.abs().sum()) == 0
    else:
        assert rel_l2(got.numpy(), want) < 1e-5


CASES = (dict(name="given3", init="3", bs=3), dict(name="given1_rep3", init="1", bs=3), dict(name="avg_img_prompt", init="3", bs=3, avg="img_prompt_emb"),
         dict(name="perturb_id", init="3", bs=3, pstage="id_emb", pstd=0.2), dict(name="perturb_prompt", init="3", bs=3, pstage="img_prompt_emb", pstd=0.3),
         dict(name="random_ids", init=None, bs=2))


def test_id_to_img_prompt_glue_vs_reference():
    from adaface_dev_amd import rng
    g = np.load(os.path.join(GOLDEN, "id2ada_glue.npz"))
    ids3 = rng.synth_input("glue.ids", (3, 512), seed=76)
    a = _id2ada()
    a.__class__.dtype = property(lambda self: torch.float32)          # fixtures are fp32 (the live dtype is fp16 on the GPU)
    out = a.map_init_id_to_img_prompt_embs(torch.nn.functional.normalize(ids3, dim=-1))
    assert rel_l2(out.numpy(), g["map.out"]) < 1e-6
    for c in CASES:
        init = None if c["init"] is None else (ids3 if c["init"] == "3" else ids3[:1])
        torch.manual_seed(808)
        _, fid, pos, neg = a.get_img_prompt_embs(init, None, None, None, id_batch_size=c["bs"], avg_at_stage=c.get("avg"),
                                                 perturb_at_stage=c.get("pstage"), perturb_std=c.get("pstd", 0.0))
        assert neg is None
        tol = 2e-3 if c["init"] is None else 1e-5                      # random IDs are drawn in fp16 (face_id_to_ada_prompt.py:384)
        assert rel_l2(fid.float().numpy(), g[f"get.{c['name']}.faceid"]) < tol, c["name"]
        assert rel_l2(pos.float().numpy(), g[f"get.{c['name']}.pos"]) < tol, c["name"]


def test_generate_adaface_embeddings_glue_vs_reference():
    from adaface_dev_amd import rng
    g = np.load(os.path.join(GOLDEN, "id2ada_glue.npz"))
    ids3 = rng.synth_input("glue.ids", (3, 512), seed=76)
    a = _id2ada()
    a.__class__.dtype = property(lambda self: torch.float32)
    for tag, kw in (("ids_avg_id", dict(face_id_embs=ids3, avg_at_stage="id_emb")), ("ids_noavg", dict(face_id_embs=ids3, avg_at_stage=None)),
                    ("ids_avg_prompt_sfx", dict(face_id_embs=ids3, avg_at_stage="img_prompt_emb", enable_static_img_suffix_embs=True)),
                    ("prompts_avg", dict(img_prompt_embs=rng.synth_input("glue.ip", (3, N_ID, D), seed=76), avg_at_stage="img_prompt_emb"))):
        with torch.no_grad():
            embs, ip, lens = a.generate_adaface_embeddings(None, **kw)
        assert tuple(embs.shape) == g[f"gen.{tag}.embs"].shape, (tag, tuple(embs.shape), g[f"gen.{tag}.embs"].shape)
        assert rel_l2(embs.numpy(), g[f"gen.{tag}.embs"]) < 1e-5, tag
        assert rel_l2(ip.numpy(), g[f"gen.{tag}.img_prompt"]) < 1e-5, tag
        assert list(lens) == g[f"gen.{tag}.lens"].tolist(), tag
