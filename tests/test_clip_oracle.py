"""Pin oracle/clip_oracle.py: MKV attention against the REFERENCE's CLIPAttentionMKV outputs, the text model (m = 1)
against transformers' CLIPTextModel (tests/golden/clip.npz).  CPU only."""
import os

import numpy as np
import torch

from conftest import GOLDEN, rel_l2
from adaface_dev_amd import rng
from oracle import clip_oracle as CO

CLIP_SMALL = dict(hidden=128, heads=2, layers=3, inter=512, vocab=1000, max_pos=77)


def small_clip_shapes(c=CLIP_SMALL, multipliers=None):
    E, I = c["hidden"], c["inter"]
    s = [("text_model.embeddings.token_embedding.weight", (c["vocab"], E)), ("text_model.embeddings.position_embedding.weight", (c["max_pos"], E))]
    for li in range(c["layers"]):
        p = f"text_model.encoder.layers.{li}."
        m = 1 if multipliers is None else multipliers[li]
        s += [(p + "self_attn.k_proj.weight", (E * m, E)), (p + "self_attn.k_proj.bias", (E * m,)), (p + "self_attn.v_proj.weight", (E * m, E)),
              (p + "self_attn.v_proj.bias", (E * m,)), (p + "self_attn.q_proj.weight", (E, E)), (p + "self_attn.q_proj.bias", (E,)),
              (p + "self_attn.out_proj.weight", (E, E)), (p + "self_attn.out_proj.bias", (E,)), (p + "layer_norm1.weight", (E,)),
              (p + "layer_norm1.bias", (E,)), (p + "mlp.fc1.weight", (I, E)), (p + "mlp.fc1.bias", (I,)), (p + "mlp.fc2.weight", (E, I)),
              (p + "mlp.fc2.bias", (E,)), (p + "layer_norm2.weight", (E,)), (p + "layer_norm2.bias", (E,))]
    s += [("text_model.final_layer_norm.weight", (E,)), ("text_model.final_layer_norm.bias", (E,))]
    return s


def test_mkv_attention_vs_reference_module():
    g = np.load(os.path.join(GOLDEN, "clip.npz"))
    E = CLIP_SMALL["hidden"]
    for m in (1, 2):
        sd = {n: rng.synth_tensor(f"mkv{m}." + n, shp, seed=20) for n, shp in (
            ("k_proj.weight", (E * m, E)), ("k_proj.bias", (E * m,)), ("v_proj.weight", (E * m, E)), ("v_proj.bias", (E * m,)),
            ("q_proj.weight", (E, E)), ("q_proj.bias", (E,)), ("out_proj.weight", (E, E)), ("out_proj.bias", (E,)))}
        for T in (22, 77):
            h = rng.synth_input(f"clip.h{T}", (2, T, E), seed=20)
            out = CO.mkv_attention(sd, "", h, CLIP_SMALL["heads"], m, True)
            assert rel_l2(out.numpy(), g[f"mkv_m{m}_T{T}"]) < 1e-5, (m, T)


def test_text_model_vs_transformers():
    g = np.load(os.path.join(GOLDEN, "clip.npz"))
    assert str(g["transformers_version"]).startswith("5.")
    sd = rng.synth_state_dict(small_clip_shapes(), seed=21)
    ids = torch.from_numpy(g["text_ids"])
    last, hs = CO.clip_text_forward(sd, CLIP_SMALL, ids)
    assert len(hs) == CLIP_SMALL["layers"] + 1
    assert rel_l2(hs[0].numpy(), g["text_hidden_0"]) < 1e-6
    assert rel_l2(hs[-3].numpy(), g["text_hidden_m3"]) < 1e-5
    assert rel_l2(last.numpy(), g["text_last"]) < 1e-5
    # weighted mix of the last 3 hidden states with weights e_3 == plain last hidden state (arc2face_models.py:291-304)
    w = torch.tensor([[0.0], [0.0], [1.0]])
    assert rel_l2(CO.clip_text_forward(sd, CLIP_SMALL, ids, None, w)[0].numpy(), g["text_last"]) < 1e-5
    # pre-computed token embeddings path == input_ids path
    tok = sd["text_model.embeddings.token_embedding.weight"][ids]
    assert rel_l2(CO.clip_text_forward(sd, CLIP_SMALL, ids, tok)[0].numpy(), g["text_last"]) < 1e-6


def test_extend_position_embeddings_like_reference():
    """77 -> 97 positions by repeating the last 20 rows (ldm/modules/encoders/modules.py:373-382)."""
    from adaface_dev_amd.adaface.arc2face_models import CLIPTextModelWrapper, clip_text_config
    m = CLIPTextModelWrapper(clip_text_config(hidden_size=32, num_attention_heads=2, num_hidden_layers=1, intermediate_size=64, vocab_size=100))
    w0 = m.text_model.embeddings.position_embedding.weight.detach().clone()
    m.extend_position_embeddings(97)
    w1 = m.text_model.embeddings.position_embedding.weight
    assert w1.shape == (97, 32) and torch.equal(w1[:77], w0) and torch.equal(w1[77:], w0[-20:])
    m.extend_position_embeddings(80)                           # never shrinks
    assert m.text_model.embeddings.position_embedding.weight.shape == (97, 32)
