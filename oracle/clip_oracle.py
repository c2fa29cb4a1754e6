"""CPU oracle for the face -> prompt encoder stack (CLIP text transformer with the reference's extensions).
TEST INFRASTRUCTURE ONLY.

The per-layer arithmetic of CLIPTextModel lives in the THIRD-PARTY package ``transformers`` (the reference pins
``transformers>=4.44.2``, requirements.txt:13, and adapts v4.34.1's forward, arc2face_models.py:234); it is absent from
/root/reference.  This file restates the published algorithm (pre-LN transformer: h += MHA(LN1(h)) with causal mask,
h += fc2(quick_gelu(fc1(LN2(h)))), final LN) and the IN-TREE extensions:

* ``CLIPAttentionMKV.forward`` (adaface/arc2face_models.py:145-231): K/V projections widened x m, each token
  contributing m keys/values, causal mask broadcast over the m copies, q pre-scaled by d^-0.5;
* ``CLIPTextModelWrapper.forward`` (arc2face_models.py:236-338): pre-computed token embeddings, normalised weighted sum
  of the last-k hidden states before the final LayerNorm (:291-304);
* slot replacement / slicing of ``inverse_img_prompt_embs`` and ``map_init_id_to_img_prompt_embs``
  (adaface/subj_basis_generator.py:488-522, adaface/face_id_to_ada_prompt.py:680-724).

Parity status: PINNED for (a) ``mkv_attention`` against outputs of the reference's own ``CLIPAttentionMKV`` module
(m = 1, 2; importable here with import-time stubs) and (b) the whole text model at m = 1 against ``transformers``
5.15.0's ``CLIPTextModel`` -- both committed in tests/golden/clip.npz by tests/golden/gen_golden.py.  The reference has
no tests for this boundary, and its wrapper does not run under transformers 5, so the hidden-state mixing
and slot replacement are anchored on the reference call sites only.
"""
import torch
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def mkv_attention(sd, p, h, heads, m=1, causal=True):
    """CLIPAttentionMKV.forward, arc2face_models.py:145-231.  h [B,T,E]."""
    B, T, E = h.shape
    d = E // heads
    q = F.linear(h, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * d ** -0.5
    k = F.linear(h, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, -1, heads, d).transpose(1, 2)   # [B,h,T*m,d]
    v = F.linear(h, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, -1, heads, d).transpose(1, 2)
    q = q.view(B, T, heads, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)                                                                             # [B,h,T,T*m]
    if causal:
        i = torch.arange(T)[:, None]
        j = torch.arange(T * m)[None, :] // m            # key j belongs to token j // m (:186-193)
        w = w.masked_fill(j > i, torch.finfo(w.dtype).min)
    w = w.softmax(dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, T, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def clip_text_forward(sd, cfg, input_ids, input_token_embs=None, hidden_state_layer_weights=None, multipliers=None):
    """cfg: dict(hidden, heads, layers).  Returns (last_hidden_state, hidden_states[list of layers+1])."""
    heads, L = cfg["heads"], cfg["layers"]
    E = cfg["hidden"]
    T = input_ids.shape[1]
    tok = sd["text_model.embeddings.token_embedding.weight"][input_ids] if input_token_embs is None else input_token_embs
    h = tok + sd["text_model.embeddings.position_embedding.weight"][:T][None]
    hs = [h]
    for li in range(L):
        p = f"text_model.encoder.layers.{li}."
        m = 1 if multipliers is None else multipliers[li]
        n1 = F.layer_norm(h, (E,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], 1e-5)
        h = h + mkv_attention(sd, p + "self_attn.", n1, heads, m, True)
        n2 = F.layer_norm(h, (E,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], 1e-5)
        f = F.linear(quick_gelu(F.linear(n2, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        h = h + f
        hs.append(h)
    if hidden_state_layer_weights is None:
        last = h
    else:
        w = hidden_state_layer_weights / hidden_state_layer_weights.sum(dim=0, keepdim=True)       # :296-299
        k = w.shape[0]
        last = (torch.stack(hs[-k:], dim=0) * w.unsqueeze(1).unsqueeze(1)).sum(dim=0)               # :300-304
    last = F.layer_norm(last, (E,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"], 1e-5)
    return last, hs


def id_to_img_prompt(sd, cfg, input_ids, id_token_pos, init_id_embs):
    """map_init_id_to_img_prompt_embs, face_id_to_ada_prompt.py:680-724: unit-norm ID (512-d) zero-padded to the hidden size
    overwrites the embedding of the 'id' token; CLIP text forward; tokens 4:20."""
    E = cfg["hidden"]
    tok = sd["text_model.embeddings.token_embedding.weight"][input_ids].clone()
    tok[:, id_token_pos] = F.pad(init_id_embs, (0, E - init_id_embs.shape[-1]))
    return clip_text_forward(sd, cfg, input_ids, tok)[0][:, 4:20]


def inverse_img_prompt(sd, cfg, input_ids, face_prompt_embs, layer_weights, n_id=16):
    """inverse_img_prompt_embs ('core'), subj_basis_generator.py:443-562 + SubjBasisGenerator.forward 692-770
    (face branch, layerwise_proj = Identity, out_id_embs_cfg_scale = 1)."""
    tok = sd["text_model.embeddings.token_embedding.weight"][input_ids].clone()
    tok[:, 4:4 + n_id] = face_prompt_embs
    return clip_text_forward(sd, cfg, input_ids, tok, layer_weights)[0][:, 4:4 + n_id]


def frozen_clip_embedder_forward(sd, cfg, input_ids, token_embs=None, last_layers_skip_weights=(0.5, 0.5)):
    """The hooked text encoder of the LDM side, ldm/modules/encoders/modules.py: ``embeddings_forward`` (:180-208: token embeddings,
    possibly patched by the embedding manager, + positions), ``encoder_forward`` (:212-259: all encoder states kept),
    ``text_model_forward`` (:264-341): ``final_layer_norm(sum_k w_k * states[-K + k])`` with w normalised to sum 1 by
    ``set_last_layers_skip_weights`` (:424-428); the last weight belongs to the last layer.  None = plain last state."""
    if last_layers_skip_weights is None:
        return clip_text_forward(sd, cfg, input_ids, token_embs)[0]
    w = torch.as_tensor(last_layers_skip_weights, dtype=torch.float32).reshape(-1, 1)
    return clip_text_forward(sd, cfg, input_ids, token_embs, w / w.sum())[0]
