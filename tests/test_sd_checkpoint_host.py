"""LatentDiffusion.init_from_ckpt: an LDM-format Stable Diffusion checkpoint (key prefixes model.diffusion_model. /
first_stage_model. / cond_stage_model.transformer.text_model. + schedule buffers + EMA keys) loads into the mirrors by name.
CPU only: the checkpoint here is synthetic (reduced widths) but laid out exactly like v1-5-pruned-emaonly.safetensors."""
import pytest
import torch

from test_vae_oracle import VAE_SMALL
from trainer_util import CFG


def _build(max_length=77):
    from adaface_dev_amd.adaface.arc2face_models import clip_text_config
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    ld = LatentDiffusion(dict(CFG, context_dim=64))
    ld.instantiate_first_stage(dict(VAE_SMALL, double_z=True))
    ld.instantiate_cond_stage(dict(max_length=max_length, clip_config=clip_text_config(
        hidden_size=64, num_attention_heads=2, num_hidden_layers=2, intermediate_size=128, vocab_size=512)))
    return ld


@pytest.mark.parametrize("fmt", ["safetensors", "ckpt"])
def test_init_from_ckpt_loads_unet_vae_and_text_encoder(tmp_path, fmt):
    from safetensors.torch import save_file
    src = _build()
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(i)) * 0.1)
    sd = {k: v.clone().contiguous() for k, v in src.state_dict().items()}
    sd["model_ema.decay"] = torch.tensor(0.9999)                                             # extras a real file carries
    sd["cond_stage_model.transformer.text_model.embeddings.position_ids"] = torch.arange(77)[None]
    path = str(tmp_path / f"sd15.{fmt}")
    if fmt == "safetensors":
        save_file(sd, path)
    else:
        torch.save({"state_dict": sd, "global_step": 1}, path)
    dst = _build(max_length=97)                                                               # training-time prompt length
    missing, unexpected = dst.init_from_ckpt(path)
    assert missing == [] and sorted(unexpected) == ["cond_stage_model.transformer.text_model.embeddings.position_ids", "model_ema.decay"]
    a, b = src.state_dict(), dst.state_dict()
    pos = "cond_stage_model.transformer.text_model.embeddings.position_embedding.weight"
    for k, v in a.items():
        if k == pos:
            assert b[k].shape[0] == 97 and torch.equal(b[k][:77], v) and torch.equal(b[k][77:], v[-20:])
        else:
            assert torch.equal(b[k], v), k
    # ignore_keys: the reference drops the U-Net / VAE when diffusers objects replace them (ddpm.py:555-566)
    dst2 = _build()
    before = dst2.model.diffusion_model.state_dict()["out.2.weight"].clone()
    missing, _ = dst2.init_from_ckpt(path, ignore_keys=["model.", "first_stage_model"])
    assert torch.equal(dst2.model.diffusion_model.state_dict()["out.2.weight"], before)
    assert any(k.startswith("model.diffusion_model.") for k in missing) and any(k.startswith("first_stage_model.") for k in missing)
    assert torch.equal(dst2.cond_stage_model.transformer.state_dict()["text_model.final_layer_norm.weight"],
                       a["cond_stage_model.transformer.text_model.final_layer_norm.weight"])
    # only_model: just the U-Net wrapper
    dst3 = _build()
    dst3.init_from_ckpt(path, only_model=True)
    assert torch.equal(dst3.model.diffusion_model.state_dict()["out.2.weight"], a["model.diffusion_model.out.2.weight"])
    assert not torch.equal(dst3.first_stage_model.state_dict()["decoder.conv_in.weight"], a["first_stage_model.decoder.conv_in.weight"])
    with pytest.raises(ValueError):
        dst3.init_from_ckpt(str(tmp_path / "weights.bin"))


def test_adaface_wrapper_base_model_path_and_adaface_ckpt(tmp_path):
    """AdaFaceWrapper(base_model_path=..., adaface_ckpt_paths=[embeddings_gs-N.pt]): the SD checkpoint fills U-Net, VAE and the
    wrapper's own text encoder BEFORE the token table grows by the 16 subject tokens; the AdaFace checkpoint fills the generator."""
    from safetensors.torch import save_file
    from adaface_dev_amd.adaface.adaface_wrapper import AdaFaceWrapper, WordTokenizer
    from adaface_dev_amd.adaface.arc2face_models import clip_text_config
    from adaface_dev_amd.adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.embedding_manager import EmbeddingManager
    import em_fixture_util as U
    ccfg = lambda: clip_text_config(hidden_size=64, num_attention_heads=2, num_hidden_layers=2, intermediate_size=128)
    src = _build()
    src.instantiate_cond_stage(dict(clip_config=ccfg()))
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(100 + i)) * 0.1)
    path = str(tmp_path / "sd15.safetensors")
    save_file({k: v.clone().contiguous() for k, v in src.state_dict().items()}, path)
    gen_src = Arc2Face_ID2AdaPrompt(clip_config=ccfg())
    with torch.no_grad():
        for i, p in enumerate(gen_src.subj_basis_generator.parameters()):
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(500 + i)) * 0.1)
    em = EmbeddingManager(U.text_embedder(WordTokenizer(), U.token_table()), ["z"], id2ada_prompt_encoder=gen_src)
    ada_path = str(tmp_path / "embeddings_gs-30.pt")
    em.save(ada_path)
    ld = LatentDiffusion(dict(CFG, context_dim=64))
    ld.instantiate_first_stage(dict(VAE_SMALL, double_z=True))
    w = AdaFaceWrapper(base_model_path=path, adaface_ckpt_paths=[ada_path], device="cpu", ldm=ld, clip_config=ccfg(),
                       id2ada_prompt_encoder=Arc2Face_ID2AdaPrompt(clip_config=ccfg()))
    a = src.state_dict()
    assert torch.equal(w.ldm.model.diffusion_model.state_dict()["input_blocks.0.0.weight"], a["model.diffusion_model.input_blocks.0.0.weight"])
    assert w.vae is w.ldm.first_stage_model and torch.equal(w.vae.state_dict()["decoder.conv_out.bias"], a["first_stage_model.decoder.conv_out.bias"])
    table = w.text_encoder.text_model.embeddings.token_embedding.weight
    assert table.shape[0] == 49408 + 16                                                      # subject tokens appended afterwards
    assert torch.equal(table[:49408], a["cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"])
    g1, g2 = gen_src.subj_basis_generator.state_dict(), w.id2ada_prompt_encoder.subj_basis_generator.state_dict()
    assert all(torch.equal(g2[k], v) for k, v in g1.items())
