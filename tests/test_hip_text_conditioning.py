"""Text-conditioning path on a real MI355X: FrozenCLIPEmbedder (hooked CLIP text encoder: embedding-manager
patching, last-layers skip weights, 97-token positions) and LatentDiffusion.get_text_conditioning with the EmbeddingManager driving
the real ID->prompt encoder stack, against the CPU oracle composition.  Reduced width (hidden 128, 3 layers), full CLIP vocabulary."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
CC = dict(hidden=128, heads=2, layers=3)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _cfg():
    from adaface_dev_amd.adaface.arc2face_models import clip_text_config
    return clip_text_config(hidden_size=128, num_attention_heads=2, num_hidden_layers=3, intermediate_size=512)


def _embedder(dev, max_length=97, weights=(1, 1), seed=61):
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
    fe = FrozenCLIPEmbedder(max_length=77, clip_config=_cfg(), last_layers_skip_weights=weights)
    rng.load_synth_weights(fe.transformer, seed=seed)
    if max_length != 77:
        fe.transformer.extend_position_embeddings(max_length)
        fe.max_length = max_length
    fe.freeze()
    sd = {k: v.detach().clone() for k, v in fe.transformer.state_dict().items()}
    return fe.to(dev), sd


def test_frozen_clip_embedder_vs_oracle(dev):
    from oracle import clip_oracle as CO
    fe, sd = _embedder(dev, 97, (1, 3))
    assert np.allclose(fe.transformer.text_model.last_layers_skip_weights, [0.25, 0.75])
    prompts = ["a photo of a dog on the moon", "", "portrait of z, , , in a park " + "very " * 120]      # empty + truncated prompts
    ids = fe.tokenize(prompts)
    assert ids.shape == (3, 97) and ids[1, 1] == 49407 and ids[2, -1] == 49407 and ids[2, -2] != 49407
    with torch.no_grad():
        z = fe.encode(prompts)
    ref = CO.frozen_clip_embedder_forward(sd, CC, ids, None, (1, 3))
    last_only = CO.frozen_clip_embedder_forward(sd, CC, ids, None, None)
    assert z.shape == (3, 97, 128) and rel_l2(z.float().cpu().numpy(), ref.numpy()) < 5e-3
    assert rel_l2(last_only.numpy(), ref.numpy()) > 2e-2                      # the skip weights matter
    # a manager hook sees the ids and the table embeddings and its output is what the encoder consumes
    seen = {}

    def hook(tok, emb):
        seen["shape"] = (tuple(tok.shape), tuple(emb.shape))
        out = emb.clone()
        out[:, 5] = 0.25
        return out
    with torch.no_grad():
        z2 = fe.encode(prompts, embedding_manager=hook)
    tok = sd["text_model.embeddings.token_embedding.weight"][ids].clone()
    tok[:, 5] = 0.25
    assert seen["shape"] == ((3, 97), (3, 97, 128))
    assert rel_l2(z2.float().cpu().numpy(), CO.frozen_clip_embedder_forward(sd, CC, ids, tok, (1, 3)).numpy()) < 5e-3
    # Dirichlet-sampled weights (randomize_clip_skip_weights): normalised, used by the next forward
    fe.set_last_layers_skip_weights([1.0, 1.0], use_as_dirichlet_weights=True)
    torch.manual_seed(4)
    fe.sample_last_layers_skip_weights()
    w = fe.transformer.text_model.last_layers_skip_weights
    assert abs(float(w.sum()) - 1) < 1e-6
    with torch.no_grad():
        z3 = fe.encode(prompts)
    assert rel_l2(z3.float().cpu().numpy(), CO.frozen_clip_embedder_forward(sd, CC, ids, None, tuple(w)).numpy()) < 5e-3
    with pytest.raises(NotImplementedError):
        fe.encode(prompts, attention_mask=None)


def _ldm_with_manager(dev, static_sfx=0):
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from trainer_util import CFG
    fe, sd_text = _embedder(dev, 97, (1, 1), seed=62)
    id2ada = Arc2Face_ID2AdaPrompt(clip_config=_cfg(), num_static_img_suffix_embs=static_sfx)
    rng.load_synth_weights(id2ada.subj_basis_generator.prompt2token_proj, seed=63)
    sd_sbg = {k: v.detach().clone() for k, v in id2ada.subj_basis_generator.prompt2token_proj.state_dict().items()}
    ld = LatentDiffusion(dict(CFG, context_dim=128))
    ld.instantiate_cond_stage(fe)
    ld.instantiate_embedding_manager({"id2ada_prompt_encoder": id2ada, "subj_name_to_cls_delta_string": {"alice": "young woman"}})
    return ld.to(dev), sd_text, sd_sbg


def _oracle_conditioning(sd_text, sd_sbg, ids, subj_rows, slot0, id2img, lw):
    from adaface_dev_amd.adaface.subj_basis_generator import template_ids
    from oracle import clip_oracle as CO
    ada = CO.inverse_img_prompt(sd_sbg, CC, template_ids(["photo", "of", "a"] + [","] * 18, 77).repeat(id2img.shape[0], 1), id2img, lw)
    tok = sd_text["text_model.embeddings.token_embedding.weight"][ids].clone()
    for j, b in enumerate(subj_rows):
        tok[b, slot0[j]:slot0[j] + 16] = ada[j % ada.shape[0]]
    return CO.frozen_clip_embedder_forward(sd_text, CC, ids, tok, (0.5, 0.5))


def test_get_text_conditioning_distill_iter_forward_and_gradients_vs_oracle(dev):
    from adaface_dev_amd import rng
    ld, sd_text, sd_sbg = _ldm_with_manager(dev)
    ld.train()
    fill = ", " * 15
    prompts = ["a photo of z" + fill + " in a park", "z" + fill]
    id2img = rng.synth_input("tc.id2img", (2, 16, 128), seed=64, scale=0.5)
    c = ld.get_text_conditioning(prompts, subj_id2img_prompt_embs=id2img.to(dev), text_conditioning_iter_type="unet_distill_iter",
                                 real_batch_size=2)
    emb, cond_in, extra = c
    assert cond_in is prompts and emb.shape == (2, 97, 128) and emb.requires_grad
    assert set(extra) == {"placeholder2indices", "prompt_emb_mask", "prompt_pad_mask", "capture_ca_activations", "use_attn_lora", "use_ffn_lora"}
    b, n = extra["placeholder2indices"]["z"]
    assert b.tolist() == [0] * 16 + [1] * 16 and n.tolist() == list(range(4, 20)) + list(range(1, 17))
    assert extra["prompt_emb_mask"].shape == (2, 97, 1) and int(extra["prompt_emb_mask"][1].sum()) == 16
    ids = ld.cond_stage_model.tokenize(prompts)
    R = rng.synth_input("tc.R", (2, 97, 128), seed=64)
    S = 64.0
    ((emb.float() * R.to(dev)).sum() * S).backward()
    sbg_r = {k: v.clone().requires_grad_(True) for k, v in sd_sbg.items()}
    lw = torch.tensor([[1.0], [2.0], [4.0]], requires_grad=True)
    ref = _oracle_conditioning(sd_text, sbg_r, ids, [0, 1], [4, 1], id2img, lw)
    (ref * R).sum().backward()
    assert rel_l2(emb.detach().float().cpu().numpy(), ref.detach().numpy()) < 5e-3
    sb = ld.embedding_manager.id2ada_prompt_encoder.subj_basis_generator
    worst = 0.0
    for name, p in sb.prompt2token_proj.named_parameters():
        if not p.requires_grad or name.endswith("k_proj.bias"):
            continue
        worst = max(worst, rel_l2((p.grad / S).cpu().numpy(), sbg_r[name].grad.numpy()))
    e_w = rel_l2((sb.hidden_state_layer_weights.grad / S).cpu().numpy(), 5.0 * lw.grad.numpy())
    print(f"get_text_conditioning: worst SubjBasisGenerator weight-gradient rel-L2 {worst:.2e}; layer-mix weights {e_w:.2e}")
    assert worst < 2e-2 and e_w < 2e-2
    assert all(p.grad is None for p in ld.cond_stage_model.parameters())          # the text encoder is frozen


def test_get_text_conditioning_compos_iter_merges_class_tokens(dev):
    """Stage-2 style batch: 2 subject prompts + 2 class prompts; one subject embedding set for the whole batch; in training the
    class string "young woman" (2 tokens) is merged into one slot and the rest of the prompt shifts left (ddpm.py:771-784)."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.util import merge_cls_token_embeddings
    ld, sd_text, sd_sbg = _ldm_with_manager(dev)
    ld.train()
    ld.iter_flags["do_comp_feat_distill"] = True
    em = ld.embedding_manager
    em.set_curr_batch_subject_names(["alice"])
    fill = ", " * 15
    prompts = ["a z" + fill + " riding a horse", "a z" + fill + " on the moon", "a young woman" + fill + " riding a horse",
               "a young woman" + fill + " on the moon"]
    id2img = rng.synth_input("tc.id2img2", (1, 16, 128), seed=65, scale=0.5)
    with torch.no_grad():
        emb, _, extra = ld.get_text_conditioning(prompts, subj_id2img_prompt_embs=id2img.to(dev), real_batch_size=1)
    assert em.iter_type == "compos_distill_iter" and em.cls_delta_string_indices == [(2, 2, 2, "alice"), (3, 2, 2, "alice")]
    ids = ld.cond_stage_model.tokenize(prompts)
    with torch.no_grad():
        unmerged = _oracle_conditioning(sd_text, sd_sbg, ids, [0, 1], [2, 2], id2img, torch.tensor([[1.0], [2.0], [4.0]]))
    ref = merge_cls_token_embeddings(unmerged, em.cls_delta_string_indices)
    assert rel_l2(emb.float().cpu().numpy(), ref.numpy()) < 5e-3
    assert rel_l2(unmerged.numpy(), ref.numpy()) > 1e-2
    ld.eval()                                                                     # inference: no merging
    with torch.no_grad():
        emb_eval = ld.get_text_conditioning(prompts, subj_id2img_prompt_embs=id2img.to(dev), real_batch_size=1)[0]
    assert rel_l2(emb_eval.float().cpu().numpy(), unmerged.numpy()) < 5e-3
    # 'text_id' ablation: the image-prompt embeddings are appended after the text embeddings
    with torch.no_grad():
        both = ld.get_text_conditioning(prompts, subj_id2img_prompt_embs=id2img.to(dev), real_batch_size=1, return_prompt_embs_type="text_id")[0]
    assert both.shape == (4, 97 + 16, 128) and torch.allclose(both[:, 97:].float().cpu(), id2img.repeat(4, 1, 1), atol=1e-3)
