"""Host-side mirror of the SD-1.5 ``text2img`` branch of the reference's ``adaface/adaface_wrapper.py`` (BASELINE configs[1],
the AdaFace inference path): ``AdaFaceWrapper.extend_tokenizer_and_text_encoder`` (:414-458), ``update_text_encoder_subj_embeddings``
(:461-489), ``update_prompt`` (:491-532), ``prepare_adaface_embeddings`` (:541-569), ``encode_prompt`` (:671-727) and ``forward``
(:730-809), with the same argument names and meaning.

The reference drives a diffusers ``StableDiffusionPipeline``; here the same steps run on this package's components -- the CLIP text
transformer (``CLIPTextModelWrapper``), ``LatentDiffusion`` + ``DDIMSampler`` (50 DDIM steps, (cond, uncond) CFG batches of 2 x
out_image_count, fused guidance + DDIM update) -- all through the gfx950 C ABI.  Offline there are no checkpoints, CLIP BPE vocabulary
files or VAE weights, so every component can be passed in, and:

* ``tokenizer``: any object with the transformers tokenizer protocol (``add_tokens``, ``convert_tokens_to_ids``, ``__len__``,
  ``__call__(..., padding="max_length", max_length=, truncation=True, return_tensors="pt").input_ids``), e.g.
  ``transformers.CLIPTokenizer.from_pretrained(local_dir)``.  The fallback ``WordTokenizer`` is a deterministic word-level stand-in
  (NOT CLIP BPE) so that synthetic-weight runs and tests exercise the same token-registration / prompt-rewriting logic;
* ``vae``: optional object with ``decode(latents / 0.18215) -> images in [-1, 1]`` (VAE decode is SURVEY.md 8f rank 3); without it
  ``forward`` returns the final latents ``[BS, 4, 64, 64]`` instead of PIL images.

SDXL / SD3 / flux / img2img pipelines, LCM, U-Net ensembles and the ConsistentID encoder are out of scope (external packages)."""
import re
import zlib

import numpy as np
import torch
import torch.nn as nn

from .. import SD15_UNET_CONFIG
from ..ldm.models.diffusion.ddim import DDIMSampler
from ..ldm.models.diffusion.ddpm import LatentDiffusion
from .arc2face_models import CLIPTextModelWrapper, clip_text_config
from .face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
from .subj_basis_generator import CLIP_BOS, CLIP_EOS, CLIP_IDS


class WordTokenizer:
    """Deterministic word-level tokenizer with the CLIP special ids (BOS 49406, EOS = pad 49407) -- a stand-in for CLIP BPE."""

    def __init__(self, vocab_size=49408):
        self.base_vocab = vocab_size
        self.added = {}

    def __len__(self):
        return self.base_vocab + len(self.added)

    def add_tokens(self, tokens):
        n = 0
        for t in tokens:
            if t not in self.added:
                self.added[t] = self.base_vocab + len(self.added)
                n += 1
        return n

    def convert_tokens_to_ids(self, tokens):
        one = isinstance(tokens, str)
        ids = [self._id(t) for t in ([tokens] if one else tokens)]
        return ids[0] if one else ids

    def _id(self, w):
        if w in self.added:
            return self.added[w]
        if w in CLIP_IDS:
            return CLIP_IDS[w]
        return 1 + zlib.crc32(w.encode()) % (CLIP_BOS - 1)

    def tokenize(self, text):
        return re.findall(r"[A-Za-z0-9_]+|[^\sA-Za-z0-9_]", text.lower() if not self.added else self._lower_keep_added(text))

    def _lower_keep_added(self, text):
        return " ".join(w if w in self.added else w.lower() for w in text.split())

    def __call__(self, texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt", **unused):
        texts = [texts] if isinstance(texts, str) else list(texts)
        rows = []
        for t in texts:
            ids = [CLIP_BOS] + [self._id(w) for w in self.tokenize(t)][:max_length - 2] + [CLIP_EOS]
            rows.append(ids + [CLIP_EOS] * (max_length - len(ids)))
        return _Encoding(input_ids=torch.tensor(rows, dtype=torch.long))


class _Encoding(dict):
    """transformers' BatchEncoding protocol: enc["input_ids"] and enc.input_ids."""
    __getattr__ = dict.__getitem__


class AdaFaceWrapper(nn.Module):
    def __init__(self, pipeline_name="text2img", base_model_path=None, adaface_encoder_types=("arc2face",), adaface_ckpt_paths=None,
                 adaface_encoder_cfg_scales=None, enabled_encoders=None, use_lcm=False, default_scheduler_name="ddim",
                 num_inference_steps=50, subject_string="z", negative_prompt=None, max_prompt_length=77,
                 enable_static_img_suffix_embs=None, device="cuda", is_training=False,
                 tokenizer=None, text_encoder=None, ldm=None, vae=None, id2ada_prompt_encoder=None, unet_config=None, clip_config=None):
        super().__init__()
        if pipeline_name not in ("text2img", None):
            raise NotImplementedError(f"pipeline {pipeline_name!r}: only the SD-1.5 text2img path (and None = face encoder only) is built")
        if list(adaface_encoder_types) != ["arc2face"]:
            raise NotImplementedError("only the Arc2Face ID encoder is in scope (ConsistentID needs an external package)")
        if use_lcm or default_scheduler_name != "ddim":
            raise NotImplementedError("only the DDIM scheduler without LCM is built")
        self.pipeline_name = pipeline_name
        self.adaface_encoder_types = list(adaface_encoder_types)
        self.adaface_ckpt_paths = adaface_ckpt_paths
        self.enabled_encoders = enabled_encoders
        self.enable_static_img_suffix_embs = enable_static_img_suffix_embs
        self.subject_string = subject_string
        self.num_inference_steps = num_inference_steps
        self.max_prompt_length = max_prompt_length
        self.device = device
        self.is_training = is_training
        self.negative_prompt = negative_prompt if negative_prompt is not None else (
            "flaws in the eyes, flaws in the face, lowres, non-HDRi, low quality, worst quality, artifacts, noise, text, watermark, glitch, "
            "mutated, ugly, disfigured, hands, partially rendered objects, partially rendered eyes, deformed eyeballs, cross-eyed, blurry, "
            "mutation, duplicate, out of frame, cropped, mutilated, bad anatomy, deformed, bad proportions, "
            "nude, naked, nsfw, topless, bare breasts")
        ccfg = clip_config or clip_text_config()
        self.tokenizer = tokenizer or WordTokenizer(ccfg.vocab_size)
        self.text_encoder = text_encoder or CLIPTextModelWrapper(ccfg)
        self.id2ada_prompt_encoder = id2ada_prompt_encoder or Arc2Face_ID2AdaPrompt(clip_config=ccfg)
        if adaface_encoder_cfg_scales is not None:
            self.id2ada_prompt_encoder.out_id_embs_cfg_scale = adaface_encoder_cfg_scales[0]
        self.encoders_num_id_vecs = [self.id2ada_prompt_encoder.num_id_vecs]
        self.ldm = None if pipeline_name is None else (ldm or LatentDiffusion(unet_config or SD15_UNET_CONFIG))
        self.vae = vae
        self.img_prompt_embs = None
        if base_model_path is not None:
            self.load_base_model(base_model_path)
        self.extend_tokenizer_and_text_encoder()
        if adaface_ckpt_paths:
            self.load_subj_basis_generator(adaface_ckpt_paths)

    # ------------------------------------------------------------------ checkpoints
    def load_base_model(self, base_model_path):
        """An LDM-format SD-1.5 checkpoint (.safetensors / .ckpt): U-Net, VAE and CLIP text encoder by their key prefixes
        (the reference hands the same file to diffusers' ``from_single_file``, adaface_wrapper.py:236-246, 301-308)."""
        from ..ldm.modules.encoders.modules import FrozenCLIPEmbedder
        if self.ldm is None:
            raise RuntimeError("pipeline_name=None builds the face encoder only: there is no U-Net to load a base model into")
        if self.ldm.first_stage_model is None:
            self.ldm.instantiate_first_stage()
        if self.ldm.cond_stage_model is None:       # share the wrapper's text encoder so that the checkpoint's CLIP weights reach it
            self.ldm.instantiate_cond_stage(FrozenCLIPEmbedder(tokenizer=self.tokenizer, transformer=self.text_encoder,
                                                               last_layers_skip_weights=None))
        missing, unexpected = self.ldm.init_from_ckpt(base_model_path)
        if self.vae is None:
            self.vae = self.ldm.first_stage_model
        return missing, unexpected

    def load_subj_basis_generator(self, adaface_ckpt_paths):
        """``embeddings_gs-N.pt`` as written by ``EmbeddingManager.save`` (pickled generator modules; the reference's class paths
        are mapped to the mirrors by adaface/ckpt.py), or a plain state dict of ``subj_basis_generator``."""
        from .ckpt import load_adaface_ckpt_file
        path = adaface_ckpt_paths[0] if isinstance(adaface_ckpt_paths, (list, tuple)) else adaface_ckpt_paths
        ck = load_adaface_ckpt_file(path.split(":")[0])
        if isinstance(ck, dict) and "string_to_subj_basis_generator_dict" in ck:
            self.id2ada_prompt_encoder.subject_string = self.subject_string
            return self.id2ada_prompt_encoder.load_adaface_ckpt(path)
        sd = ck.get("subj_basis_generator", ck)
        if not isinstance(sd, dict):
            sd = sd.state_dict()
        return self.id2ada_prompt_encoder.subj_basis_generator.load_state_dict(sd, strict=False)

    # ------------------------------------------------------------------ tokens (reference :414-489)
    def extend_tokenizer_and_text_encoder(self):
        if np.sum(self.encoders_num_id_vecs) < 1:
            raise ValueError(f"encoders_num_id_vecs has to be larger or equal to 1, but is {self.encoders_num_id_vecs}")
        self.all_placeholder_tokens, self.placeholder_tokens_strs, self.encoder_placeholder_tokens = [], [], []
        for i in range(len(self.adaface_encoder_types)):
            toks = [f"{self.subject_string}_{i}_{j}" for j in range(self.encoders_num_id_vecs[i])]
            self.all_placeholder_tokens.extend(toks)
            self.encoder_placeholder_tokens.append(toks)
            self.placeholder_tokens_strs.append(" ".join(toks))
        self.all_placeholder_tokens_str = " ".join(self.placeholder_tokens_strs)
        self.updated_tokens_str = self.all_placeholder_tokens_str
        self.all_encoders_updated_token_strs = list(self.placeholder_tokens_strs)
        self.all_null_placeholder_tokens_str = " ".join([", "] * len(self.all_placeholder_tokens))
        n = self.tokenizer.add_tokens(self.all_placeholder_tokens)
        if n != np.sum(self.encoders_num_id_vecs):
            raise ValueError(f"The tokenizer already contains some of the tokens {self.all_placeholder_tokens_str}. Please pass a different"
                             " `subject_string` that is not already in the tokenizer.")
        self.placeholder_token_ids = self.tokenizer.convert_tokens_to_ids(self.all_placeholder_tokens)
        emb = self.text_encoder.text_model.embeddings.token_embedding                  # resize_token_embeddings(len(tokenizer))
        if emb.num_embeddings < len(self.tokenizer):
            new = nn.Embedding(len(self.tokenizer), emb.embedding_dim).to(device=emb.weight.device, dtype=emb.weight.dtype)
            with torch.no_grad():
                new.weight[:emb.num_embeddings] = emb.weight
                new.weight[emb.num_embeddings:] = emb.weight.mean(dim=0, keepdim=True)
            new.weight.requires_grad_(emb.weight.requires_grad)
            self.text_encoder.text_model.embeddings.token_embedding = new

    def update_text_encoder_subj_embeddings(self, subj_embs, lens_subj_emb_segments):
        token_embeds = self.text_encoder.text_model.embeddings.token_embedding.weight.data
        all_tokens, all_strs, idx = [], [], 0
        with torch.no_grad():
            for i, encoder_type in enumerate(self.adaface_encoder_types):
                if (self.enabled_encoders is not None) and (encoder_type not in self.enabled_encoders):
                    idx += lens_subj_emb_segments[i]
                    continue
                toks = []
                for j in range(lens_subj_emb_segments[i]):
                    tok = f"{self.subject_string}_{i}_{j}"
                    token_embeds[self.tokenizer.convert_tokens_to_ids(tok)] = subj_embs[idx].to(token_embeds.dtype)
                    toks.append(tok)
                    idx += 1
                all_tokens.extend(toks)
                all_strs.append(" ".join(toks))
        self.updated_tokens_str = " ".join(all_strs)
        self.all_encoders_updated_token_strs = all_strs

    def update_prompt(self, prompt, placeholder_tokens_pos="append", repeat_prompt_for_each_encoder=True, use_null_placeholders=False):
        if prompt is None:
            prompt = ""
        if use_null_placeholders:
            all_placeholder_tokens_str = self.all_null_placeholder_tokens_str
            if not re.search(r"\b(man|woman|person|child|girl|boy)\b", prompt.lower()):
                all_placeholder_tokens_str = "person " + all_placeholder_tokens_str
            repeat_prompt_for_each_encoder = False
        else:
            all_placeholder_tokens_str = self.updated_tokens_str
        prompt = re.sub(r"\b(a|an|the)\s+" + self.subject_string + r"\b,?", "", prompt)
        prompt = re.sub(r"\b" + self.subject_string + r"\b,?", "", prompt)
        if placeholder_tokens_pos not in ("prepend", "append"):
            raise ValueError(f"placeholder_tokens_pos {placeholder_tokens_pos!r}")
        join = (lambda toks: toks + " " + prompt) if placeholder_tokens_pos == "prepend" else (lambda toks: prompt + " " + toks)
        if repeat_prompt_for_each_encoder:
            return ", ".join(join(s) for s in self.all_encoders_updated_token_strs)
        return join(all_placeholder_tokens_str)

    # ------------------------------------------------------------------ embeddings (reference :541-569, 671-727)
    def prepare_adaface_embeddings(self, image_paths, face_id_embs=None, avg_at_stage="id_emb", perturb_at_stage=None, perturb_std=0,
                                   update_text_encoder=True):
        if face_id_embs is not None and face_id_embs.shape[0] > 1 and avg_at_stage == "id_emb":
            # pre-extracted IDs of several images of the subject: the 'id_emb' averaging of the reference lives in its image -> ID
            # extraction (insightface, absent here), so it is applied to the IDs brought in instead
            face_id_embs = self.id2ada_prompt_encoder.average_id_embs(face_id_embs)
        embs, img_prompt_embs, lens = self.id2ada_prompt_encoder.generate_adaface_embeddings(
            image_paths, face_id_embs=face_id_embs, img_prompt_embs=None, avg_at_stage=avg_at_stage, perturb_at_stage=perturb_at_stage,
            perturb_std=perturb_std, enable_static_img_suffix_embs=self.enable_static_img_suffix_embs)
        if embs is None:
            return None
        self.img_prompt_embs = img_prompt_embs
        if embs.ndim == 4:
            embs = embs.squeeze(0).squeeze(0)
        elif embs.ndim == 3:
            embs = embs.squeeze(0)
        if update_text_encoder:
            self.update_text_encoder_subj_embeddings(embs, lens)
        return embs

    @torch.no_grad()
    def _encode(self, texts, device):
        ids = self.tokenizer(texts, padding="max_length", max_length=self.max_prompt_length, truncation=True, return_tensors="pt").input_ids
        return self.text_encoder(input_ids=ids.to(device))[0]

    def encode_prompt(self, prompt, negative_prompt=None, placeholder_tokens_pos="append", ablate_prompt_only_placeholders=False,
                      ablate_prompt_no_placeholders=False, ablate_prompt_embed_type="ada", nonmix_prompt_emb_weight=0,
                      repeat_prompt_for_each_encoder=True, device=None, verbose=False):
        if negative_prompt is None:
            negative_prompt = self.negative_prompt
        device = device or self.device
        if ablate_prompt_embed_type != "ada" or nonmix_prompt_emb_weight > 0:
            raise NotImplementedError("image-prompt mixing ablations (mix_ada_embs_with_other_embs) are not built")
        if ablate_prompt_only_placeholders:
            prompt = self.updated_tokens_str
        else:
            prompt = self.update_prompt(prompt, placeholder_tokens_pos=placeholder_tokens_pos,
                                        repeat_prompt_for_each_encoder=repeat_prompt_for_each_encoder,
                                        use_null_placeholders=ablate_prompt_no_placeholders)
        if verbose:
            print(f"Subject prompt:\n{prompt}")
        self.text_encoder.to(device)
        return self._encode([prompt], device), self._encode([negative_prompt], device), None, None

    # ------------------------------------------------------------------ generation (reference :730-809)
    @torch.no_grad()
    def forward(self, noise, prompt, prompt_embeds=None, negative_prompt=None, placeholder_tokens_pos="append", guidance_scale=6.0,
                out_image_count=4, ref_img_strength=0.8, generator=None, ablate_prompt_only_placeholders=False,
                ablate_prompt_no_placeholders=False, ablate_prompt_embed_type="ada", nonmix_prompt_emb_weight=0,
                repeat_prompt_for_each_encoder=True, verbose=False):
        if self.ldm is None:
            raise RuntimeError("pipeline_name=None builds the face encoder only")
        if prompt_embeds is None:
            pe, ne, _, _ = self.encode_prompt(prompt, negative_prompt, placeholder_tokens_pos=placeholder_tokens_pos,
                                              ablate_prompt_only_placeholders=ablate_prompt_only_placeholders,
                                              ablate_prompt_no_placeholders=ablate_prompt_no_placeholders,
                                              ablate_prompt_embed_type=ablate_prompt_embed_type,
                                              nonmix_prompt_emb_weight=nonmix_prompt_emb_weight,
                                              repeat_prompt_for_each_encoder=repeat_prompt_for_each_encoder, device=self.device, verbose=verbose)
        elif len(prompt_embeds) in (2, 4):
            pe, ne = prompt_embeds[0], prompt_embeds[1]
        else:
            raise ValueError("prompt_embeds must be a 2- or 4-tuple")
        pe = pe.repeat(out_image_count, 1, 1)
        ne = None if ne is None else ne.repeat(out_image_count, 1, 1)
        noise = noise.to(device=self.device, dtype=torch.float32)
        self.ldm.to(self.device)
        sampler = DDIMSampler(self.ldm)
        cond = (pe, [prompt or ""] * out_image_count, {})
        uncond = None if ne is None else (ne, [negative_prompt or self.negative_prompt] * out_image_count, {})
        latents, _ = sampler.sample(self.num_inference_steps, out_image_count, tuple(noise.shape[1:]), conditioning=cond, x_T=noise,
                                    verbose=False, guidance_scale=guidance_scale, unconditional_conditioning=uncond)
        if self.vae is None:
            return latents
        images = self.vae.decode(latents / 0.18215)
        images = ((images.float() / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        from PIL import Image
        return [Image.fromarray(im) for im in images]
