"""Host-side mirror of the hot-path slice of the reference's ``ldm/models/diffusion/ddpm.py``:
the noise schedule (``DDPM.register_schedule`` 294-345), ``q_sample`` /
``predict_start_from_noise`` (389-398), ``LatentDiffusion.apply_model`` (1560-1569) and the
U-Net wrapper contract ``self.model(x, t, cond_context, out_dtype)`` of
``DiffusersUNetWrapper.forward`` (4187-4252).

The loss assemblies of the compositional-distillation and normal-recon iterations live in ``ddpm_losses.py`` (a mixin of
``LatentDiffusion``).  Out of scope: Lightning glue, image logging.  The class is a plain ``nn.Module`` so samplers and trainers written against the reference's
``LatentDiffusion`` attribute surface (``num_timesteps``, ``alphas_cumprod``, ``betas``,
``device``, ``q_sample``, ``apply_model``, ``model.diffusion_model``) keep working.
"""
import copy
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ...modules.diffusionmodules.openaimodel import UNetModel
from ...modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from ...util import calc_recon_loss
from .ddpm_losses import CompReconLossesMixin


class _QSampleGrad(torch.autograd.Function):
    """Attaches the value computed by the q_sample kernel to the graph of x_start."""

    @staticmethod
    def forward(ctx, x_start, xt, sa):
        ctx.save_for_backward(sa)
        ctx.dt = x_start.dtype
        return xt.view_as(xt)

    @staticmethod
    def backward(ctx, g):
        (sa,) = ctx.saved_tensors
        return (g * sa.view(-1, *([1] * (g.dim() - 1)))).to(ctx.dt), None, None


class UNetWrapper(nn.Module):
    """``self.model`` of LatentDiffusion: forward(x, t, cond_context, out_dtype=float32) with
    cond_context = (prompt_emb [b,L,768], prompt_in list[str], extra_info dict).  extra_info is
    read (img_mask, capture_ca_activations) and written (ca_layers_activations), like the
    reference wrapper (ddpm.py:4187-4252).  ``use_attn_lora`` / ``use_ffn_lora`` / ``ffn_lora_adapter_name`` select DoRA adapters
    (``load_unet_loras``), which on this path are MERGED into the layer weights when the requested set changes and un-merged
    when it is switched off (adaface/lora.py): inference only -- with gradients enabled they raise (training-time DoRA is
    SURVEY.md 8f rank 1)."""

    def __init__(self, unet_config):
        super().__init__()
        self.diffusion_model = UNetModel(**unet_config)
        for m in self.diffusion_model.modules():              # this wrapper is the LIVE path's seam: its processor's key-mask rule
            if hasattr(m, "live_mask_rule"):
                m.live_mask_rule = True
        self.use_attn_lora = False
        self.use_ffn_lora = False
        self.unet_lora_modules = nn.ModuleDict()
        self.unet_lora_state_dict = None
        self.ffn_lora = None                  # modules/dora.py::UNetLoRA (trainable adapters), see set_up_ffn_loras
        self.attn_lora = None                 # modules/dora.py::UNetAttnLoRA (trainable attention adapters), see set_up_attn_loras
        self.q_lora_updates_query = False
        self._merged = (None, False)          # (ffn adapter name | None, attention LoRA on)
        self._merge_saved = {}
        # learnable scale of the normalised subject-token scores, one per captured cross-attention layer (init 0.8,
        # diffusers_attn_lora_capture.py:166-168), under the reference's unet_lora_modules key names (:512-514)
        self.cross_attn_scale_factors = nn.ParameterDict({
            f"up_blocks_3_attentions_{i}_transformer_blocks_0_attn2_processor_cross_attn_scale_factor": nn.Parameter(torch.tensor(0.8))
            for i in range(3)})

    @property
    def dtype(self):
        return torch.float16

    def load_unet_state_dict(self, unet_state_dict):
        self._set_loras((None, False))
        self.diffusion_model.load_state_dict(unet_state_dict)

    def load_unet_loras(self, lora_state_dict):
        """peft-style state dict with diffusers layer names (``up_blocks.3.resnets.1.conv1.lora_A.unet_distill.weight`` ...)."""
        self._set_loras((None, False))
        self.unet_lora_state_dict = dict(lora_state_dict)

    def set_up_ffn_loras(self, adapter_names=("recon_loss", "unet_distill", "comp_distill"), lora_rank=192, lora_alpha=16,
                         lora_dropout=0.1):
        """Reference ``set_up_ffn_loras`` (diffusers_attn_lora_capture.py:541-591): create the trainable DoRA adapters.  Call after
        the U-Net is on its device."""
        from ...modules.dora import UNetLoRA
        self.ffn_lora = UNetLoRA(self.diffusion_model, adapter_names, lora_rank, lora_alpha, lora_dropout)
        dev = next(self.diffusion_model.parameters()).device
        self.ffn_lora.to(dev)
        self.unet_lora_modules = self.ffn_lora.adapters
        return self.ffn_lora

    def _set_loras(self, want):
        if want == self._merged:
            return
        from ....adaface import lora
        if self._merge_saved:
            lora.unmerge_unet_loras(self.diffusion_model, self._merge_saved)
            self._merge_saved = {}
        if want != (None, False):
            if self.unet_lora_state_dict is None:
                raise RuntimeError("use_attn_lora / use_ffn_lora requested but no adapters are loaded (UNetWrapper.load_unet_loras)")
            if want[0] is not None:
                self._merge_saved.update(lora.merge_unet_loras(self.diffusion_model, self.unet_lora_state_dict, want[0],
                                                               use_ffn_lora=True, use_attn_lora=False))
            if want[1]:
                self._merge_saved.update(lora.merge_unet_loras(self.diffusion_model, self.unet_lora_state_dict, "default",
                                                               use_ffn_lora=False, use_attn_lora=True))
        self._merged = want

    def forward(self, x, t, cond_context, out_dtype=torch.float32):
        prompt_emb, prompt_in, extra_info = cond_context
        ei = extra_info or {}
        if extra_info is not None:
            # read by the capture / score-rewrite pass (modules/diffusionmodules/capture_graph.py), ignored by the plain passes
            extra_info["_cross_attn_scale_factors"] = list(self.cross_attn_scale_factors.values())
        try:
            return self._forward(x, t, cond_context, out_dtype)
        finally:
            if extra_info is not None:
                extra_info.pop("_cross_attn_scale_factors", None)

    def precompute_trunks(self, entries, n_tail=3):
        """entries: [(key, x_noisy row [1, 4, h, w], t row [1], prompt row [1, L, D])] -> {key: (h [1, H, W, C], [skip tensors])}: the U-Net
        below its last ``n_tail`` decoder blocks for all rows as ONE gradient-free batch (UNetModel.hip_trunk, the inference walk).  Nothing in
        that part of the network depends on the adapter / score-rewrite / capture flags of a pass (they live in the tail: FFN adapters on
        output blocks 10-11, attention adapters and rewrites on layers 22-24), so gradient-free passes that share (x, t, prompt) rows may share
        it (LatentDiffusion.guided_denoise, 'subject-compos')."""
        unet = self.diffusion_model
        with torch.no_grad():
            x = torch.cat([e[1] for e in entries], dim=0).detach()
            tt = torch.cat([e[2] for e in entries], dim=0)
            ctx = torch.cat([e[3] for e in entries], dim=0).detach().to(torch.float16).contiguous()
            from ...modules.diffusionmodules.util import to_nhwc_f16
            h, skips = unet.hip_trunk(to_nhwc_f16(x, ops.round_up(unet.in_channels, 8)), unet._embed(tt), ctx, None, n_tail)
        return {e[0]: (h[i:i + 1], [sk[i:i + 1] for sk in skips]) for i, e in enumerate(entries)}

    def set_up_attn_loras(self, layer_names=("q", "k", "v", "out"), lora_rank=192, lora_scale_down=8, lora_dropout=0.1,
                          q_lora_updates_query=False):
        """Reference ``set_up_attn_processors`` with ``use_attn_lora`` (diffusers_attn_lora_capture.py:451-537): trainable DoRA adapters
        on to_q / to_k / to_v / to_out.0 of the three captured cross-attention layers.  Call after the U-Net is on its device."""
        from ...modules.dora import UNetAttnLoRA
        self.attn_lora = UNetAttnLoRA(self.diffusion_model, layer_names, lora_rank, lora_scale_down, lora_dropout)
        self.attn_lora.to(next(self.diffusion_model.parameters()).device)
        self.q_lora_updates_query = q_lora_updates_query
        return self.attn_lora

    def _forward(self, x, t, cond_context, out_dtype):
        prompt_emb, prompt_in, extra_info = cond_context
        ei = extra_info or {}
        want = (ei.get("ffn_lora_adapter_name") if ei.get("use_ffn_lora", False) else None, bool(ei.get("use_attn_lora", False)))
        if want[1] and self.attn_lora is not None:
            # module-held attention adapters: applied UN-merged by the explicit attention of the capture pass, with or without
            # gradients (the q adapter feeds query2 only, the others key / value / output: diffusers_attn_lora_capture.py:239-288, 328-331)
            extra_info["_attn_lora_adapters"] = self.attn_lora.active()
            extra_info["q_lora_updates_query"] = self.q_lora_updates_query
            want = (want[0], False)
            try:
                return self._forward_ffn(x, t, cond_context, out_dtype, want)
            finally:
                extra_info.pop("_attn_lora_adapters", None)
        return self._forward_ffn(x, t, cond_context, out_dtype, want)

    def _forward_ffn(self, x, t, cond_context, out_dtype, want):
        prompt_emb, prompt_in, extra_info = cond_context
        ei = extra_info or {}
        training_pass = torch.is_grad_enabled() and (x.requires_grad or prompt_emb.requires_grad or
                                                     (self.ffn_lora is not None and any(p.requires_grad for p in self.ffn_lora.parameters())))
        if want != (None, False) and training_pass:
            # TRAINING through the adapters: un-merged base weights + the DoRA branch with gradients (modules/dora.py)
            if want[1]:
                raise RuntimeError("use_attn_lora in a training pass needs module-held adapters (UNetWrapper.set_up_attn_loras); "
                                   "adapters loaded from a state dict are inference-only (merged)")
            if want[0] is not None and self.ffn_lora is None:
                raise RuntimeError("use_ffn_lora requested in a training pass but no trainable adapters exist (UNetWrapper.set_up_ffn_loras)")
            self._set_loras((None, False))
            extra_info["_ffn_lora_adapters"] = self.ffn_lora.active(want[0])
            try:
                out = self.diffusion_model(x, t, prompt_emb, extra_info=extra_info)
            finally:
                extra_info.pop("_ffn_lora_adapters", None)
            return out.to(out_dtype)
        if want != (None, False) and self.ffn_lora is not None and self.unet_lora_state_dict is None:
            key = tuple(ops.param_key(p) for p in self.ffn_lora.parameters())        # inference with the module-held adapters: merge them
            if key != getattr(self, "_merged_from_key", None):
                self._set_loras((None, False))
                self._module_sd = self.ffn_lora.peft_state_dict()
                self._merged_from_key = key
            self.unet_lora_state_dict, keep = self._module_sd, True
            try:
                self._set_loras(want)
            finally:
                self.unet_lora_state_dict = None
        else:
            self._set_loras(want)
        out = self.diffusion_model(x, t, prompt_emb, extra_info=extra_info)
        return out.to(out_dtype)


class LatentDiffusion(CompReconLossesMixin, nn.Module):
    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=0.00085, linear_end=0.012,
                 cosine_s=8e-3, parameterization="eps"):
        super().__init__()
        assert parameterization == "eps"
        self.parameterization = parameterization
        self.model = UNetWrapper(unet_config)
        self.uncond_context = None   # (uncond_emb [1,L,768], [""], {}) when guided_denoise runs with cfg_scale > 1
        self.first_stage_model = None   # AutoencoderKLDecoder (instantiate_first_stage); VAE scale factor of SD-1.5:
        self.scale_factor = 0.18215
        self.unet_teacher = None     # adaface.unet_teachers.UNetTeacher (frozen)
        self.comp_distill_priming_unet = None   # UNetTeacher with CFG on: primes the Stage-2 latents (ddpm.py:582-610)
        self.cond_stage_model = None    # FrozenCLIPEmbedder (instantiate_cond_stage)
        self.embedding_manager = None   # EmbeddingManager (instantiate_embedding_manager)
        self.arcface = None          # modules/arcface_wrapper.ArcFaceWrapper (face crops + ResNetFace-18) for the face-gated loss terms
        self.flow_model = None       # ddpm.py:652-662: the GMA flow network only exists under use_face_flow_for_sc_matching_loss (default False)
        self.iter_flags = {"do_comp_feat_distill": False, "do_unet_distill": False}
        self.num_id_vecs, self.num_static_img_suffix_embs = 16, 0
        self.register_schedule(beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start,
                               linear_end=linear_end, cosine_s=cosine_s)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        to_torch = partial(torch.tensor, dtype=torch.float32)
        reg = lambda n, v: self.register_buffer(n, to_torch(v))
        reg("betas", betas)
        reg("alphas_cumprod", alphas_cumprod)
        reg("alphas_cumprod_prev", alphas_cumprod_prev)
        reg("sqrt_alphas_cumprod", np.sqrt(alphas_cumprod))
        reg("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - alphas_cumprod))
        reg("log_one_minus_alphas_cumprod", np.log(1.0 - alphas_cumprod))
        reg("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / alphas_cumprod))
        reg("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / alphas_cumprod - 1))
        posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
        reg("posterior_variance", posterior_variance)
        reg("posterior_log_variance_clipped", np.log(np.maximum(posterior_variance, 1e-20)))
        reg("posterior_mean_coef1", betas * np.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod))
        reg("posterior_mean_coef2", (1.0 - alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - alphas_cumprod))

    # ------------------------------------------------------------------ Stable Diffusion checkpoints
    def init_from_ckpt(self, path, ignore_keys=(), only_model=False):
        """Load an LDM-format Stable Diffusion checkpoint (``v1-5-pruned-emaonly.safetensors`` / ``.ckpt``; reference
        ``DDPM.init_from_ckpt`` ddpm.py:347-387).  The key prefixes of such a file are this class's own module paths --
        ``model.diffusion_model.*`` (U-Net), ``first_stage_model.*`` (VAE, if ``instantiate_first_stage`` was called),
        ``cond_stage_model.transformer.text_model.*`` (CLIP text encoder, if ``instantiate_cond_stage`` was called) and the schedule
        buffers -- so it is one non-strict ``load_state_dict``.  Keys starting with an entry of ``ignore_keys`` are dropped first.  When
        the text encoder's position table was extended beyond 77 rows (``max_length`` 97), the checkpoint's 77 rows are loaded and the
        extension is re-applied, as the reference's constructor order achieves (:373-382, :555).  Returns (missing, unexpected)."""
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path, device="cpu")
        elif path.endswith(".ckpt") or path.endswith(".pt"):
            sd = torch.load(path, map_location="cpu", weights_only=False)
            sd = sd.get("state_dict", sd)
        else:
            raise ValueError(f"Unknown checkpoint format: {path}")
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        self.model._set_loras((None, False))                  # never load into merged weights
        pos_key = "cond_stage_model.transformer.text_model.embeddings.position_embedding.weight"
        extended = None
        if self.cond_stage_model is not None and pos_key in sd and not only_model:
            emb = self.cond_stage_model.transformer.text_model.embeddings.position_embedding
            if emb.num_embeddings > sd[pos_key].shape[0]:
                extended = emb.num_embeddings
                pos = sd.pop(pos_key)
                el = extended - pos.shape[0]
                with torch.no_grad():
                    emb.weight.copy_(torch.cat([pos, pos[-el:]], dim=0).to(emb.weight))
        if only_model:
            res = self.model.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=False)
        else:
            res = self.load_state_dict(sd, strict=False)
        missing = [k for k in res.missing_keys if not (extended and k == pos_key)]
        return missing, list(res.unexpected_keys)

    # ------------------------------------------------------------------ text conditioning
    def instantiate_cond_stage(self, config=None):
        """The hooked, frozen CLIP text encoder (reference ddpm.py:705-710, 612-617).  ``config``: a ``FrozenCLIPEmbedder`` or its
        constructor kwargs (the yaml's ``cond_stage_config.params``)."""
        from ...modules.encoders.modules import FrozenCLIPEmbedder
        m = config if isinstance(config, nn.Module) else FrozenCLIPEmbedder(**(config or {}))
        m.initialize_hooks()
        m.eval()
        for p in m.parameters():
            p.requires_grad_(False)
        self.cond_stage_model = m
        return m

    def instantiate_embedding_manager(self, config=None, text_embedder=None):
        """Reference ddpm.py:712-733: the manager shares the U-Net's LoRA modules for checkpointing / parameter groups."""
        from ...modules.embedding_manager import EmbeddingManager
        text_embedder = text_embedder or self.cond_stage_model
        if text_embedder is None:
            raise RuntimeError("instantiate_cond_stage() first: the embedding manager needs the text encoder's tokenizer")
        if isinstance(config, nn.Module):
            m = config
        else:
            kw = dict(config or {})
            kw.setdefault("subject_strings", ["z"])
            kw.setdefault("unet_lora_modules", self.model.unet_lora_modules if len(self.model.unet_lora_modules) > 0 else None)
            m = EmbeddingManager(text_embedder, **kw)
        self.embedding_manager = m
        self.num_id_vecs = m.id2ada_prompt_encoder.num_id_vecs
        self.num_static_img_suffix_embs = m.id2ada_prompt_encoder.num_static_img_suffix_embs
        return m

    def get_text_conditioning(self, cond_in, subj_id2img_prompt_embs=None, clip_bg_features=None, randomize_clip_weights=False,
                              return_prompt_embs_type="text", text_conditioning_iter_type=None, real_batch_size=-1):
        """list of prompts -> cond_context ``(prompt_embeddings [B,T,768], cond_in, extra_info)`` (reference ddpm.py:739-853): the
        embedding manager patches the subject tokens inside the text encoder's embedding step; in training, class strings found
        in class prompts are merged into one token so that they line up with the subject token of the paired prompt."""
        from ...util import merge_cls_token_embeddings
        if self.cond_stage_model is None or self.embedding_manager is None:
            raise RuntimeError("instantiate_cond_stage() and instantiate_embedding_manager() first")
        self.cond_stage_model.device = self.device
        if randomize_clip_weights:
            self.cond_stage_model.sample_last_layers_skip_weights()
        if text_conditioning_iter_type is None:
            text_conditioning_iter_type = ("compos_distill_iter" if self.iter_flags["do_comp_feat_distill"] else
                                           "unet_distill_iter" if self.iter_flags["do_unet_distill"] else "recon_iter")
        em = self.embedding_manager
        em.set_image_prompts_and_iter_type(subj_id2img_prompt_embs, clip_bg_features, text_conditioning_iter_type, real_batch_size)
        prompt_embeddings = self.cond_stage_model.encode(cond_in, embedding_manager=em)
        if self.training:
            prompt_embeddings = merge_cls_token_embeddings(prompt_embeddings, em.cls_delta_string_indices)
        if return_prompt_embs_type in ("id", "text_id"):
            if text_conditioning_iter_type == "plain_text_iter" and subj_id2img_prompt_embs is None:
                subj_id2img_prompt_embs = (prompt_embeddings[:, :self.num_id_vecs] if return_prompt_embs_type == "id"
                                           else prompt_embeddings[:, -self.num_id_vecs:])
            elif subj_id2img_prompt_embs is not None:
                assert subj_id2img_prompt_embs.shape[1] == self.num_id_vecs + self.num_static_img_suffix_embs
                subj_id2img_prompt_embs = subj_id2img_prompt_embs.repeat(len(cond_in) // subj_id2img_prompt_embs.shape[0], 1, 1)
            if return_prompt_embs_type == "id":
                prompt_embeddings = subj_id2img_prompt_embs
            else:
                prompt_embeddings = torch.cat([prompt_embeddings, subj_id2img_prompt_embs.to(prompt_embeddings.dtype)], dim=1)
        extra_info = {"placeholder2indices": copy.copy(em.placeholder2indices), "prompt_emb_mask": copy.copy(em.prompt_emb_mask),
                      "prompt_pad_mask": copy.copy(em.prompt_pad_mask), "capture_ca_activations": False, "use_attn_lora": False,
                      "use_ffn_lora": False}
        return (prompt_embeddings, cond_in, extra_info)

    def instantiate_first_stage(self, ddconfig=None, embed_dim=4):
        """The first-stage VAE (reference instantiate_first_stage ddpm.py:698-704; frozen, eval): encoder + decoder."""
        from ...modules.diffusionmodules.model import AutoencoderKL
        self.first_stage_model = AutoencoderKL(ddconfig, embed_dim).eval()
        for p in self.first_stage_model.parameters():
            p.requires_grad_(False)
        return self.first_stage_model

    @torch.no_grad()
    def decode_first_stage(self, z):
        """latent -> image, roughly [-1, 1] (reference ddpm.py:889-896: z / scaling_factor, then vae.decode)."""
        if self.first_stage_model is None:
            raise RuntimeError("no first-stage decoder: call instantiate_first_stage() and load its weights")
        return self.first_stage_model.decode(z / self.scale_factor)

    @torch.no_grad()
    def encode_first_stage(self, x, mask=None):
        """image [B,3,H,W] in [-1, 1] -> posterior parameters (mean, logvar) of the latent (reference ddpm.py:875-887 / autoencoder.py:30-34)."""
        if self.first_stage_model is None:
            raise RuntimeError("no first-stage model: call instantiate_first_stage() and load its weights")
        return self.first_stage_model.encode(x, mask)

    def get_first_stage_encoding(self, encoder_posterior, generator=None):
        """sample the DiagonalGaussian posterior and apply the latent scale factor (reference ddpm.py:718-727)."""
        mean, logvar = encoder_posterior
        z = mean + torch.exp(0.5 * logvar) * torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        return self.scale_factor * z

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        sa = self.sqrt_alphas_cumprod[t]
        xt = ops.q_sample(x_start.detach(), noise, sa, self.sqrt_one_minus_alphas_cumprod[t])
        if torch.is_grad_enabled() and x_start.requires_grad:
            # recon on pure noise chains the denoising steps WITH gradients (ddpm.py:1815-1823: the x0 prediction of one step is the
            # x_start of the next): d x_t / d x_start = sqrt(alpha_bar_t)
            xt = _QSampleGrad.apply(x_start, xt, sa)
        return xt

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def apply_model(self, x_noisy, t, cond_context, use_attn_lora=False, use_ffn_lora=False, ffn_lora_adapter_name=None):
        cond_context[2]["use_attn_lora"] = use_attn_lora
        cond_context[2]["use_ffn_lora"] = use_ffn_lora
        cond_context[2]["ffn_lora_adapter_name"] = ffn_lora_adapter_name
        return self.model(x_noisy, t, cond_context)

    # ------------------------------------------------------------------ Stage-1 distillation (ddpm.py:1597-1750, 2984-3184)
    unet_distill_weight = 8          # ddpm.py:2367
    batch_student_steps = True       # one student U-Net call for all denoising steps of a micro-batch
    cache_uncond_in_step = os.environ.get("AF_CACHE_UNCOND", "1") != "0"       # recon steps: the null-prompt pass once per step, not once per guided pass
    skip_unread_cls_priming = os.environ.get("AF_SKIP_CLS_PRIMING", "1") != "0"   # recon steps: no class-prompt pass on priming steps (nothing reads it); its own switch
    batch_no_grad_instances = os.environ.get("AF_BATCH_NO_GRAD", "1") != "0"   # subject-compos steps: SS + SR as one no-grad pass where their flags agree
    share_no_grad_trunk = os.environ.get("AF_SHARE_TRUNK", "1") != "0"        # subject-compos steps: ONE batched trunk pass for every gradient-free instance pass of the step
    batch_cond_with_uncond = os.environ.get("AF_BATCH_UNCOND", "1") != "0"    # gradient-free guided passes: the prompt rows and the null-prompt rows as ONE U-Net call where their flags agree
    res_hidden_states_gradscale = 0.5   # reference ctor default (ddpm.py:140): gradient scale of the decoder's skip inputs

    def guided_denoise(self, x_start, noise, t, cond_context, uncond_emb=None, img_mask=None, subj_indices=None,
                       normalize_cross_attn=False, mix_sc_mc_attn=False, batch_part_has_grad="all", do_pixel_recon=False,
                       cfg_scale=-1, capture_ca_activations=False, res_hidden_states_gradscale=1, use_attn_lora=False,
                       use_ffn_lora=False, ffn_lora_adapter_name=None, uncond_cache=None):
        """q_sample -> U-Net (gradient mode 'all' / 'none' / 'subject-compos') -> optional no-grad unconditional pass and CFG combine
        eps = eps_c * s - eps_u * (s - 1) -> optional x0 (reference ddpm.py:1597-1750).

        'subject-compos' (Stage 2, :1635-1707): the batch is four one-instance blocks [SS, SC, SR (= sc_comp_rep), MC].  SS and SR run
        without gradient (SR keeps the caller's ``normalize_cross_attn``); then either SC and MC TOGETHER as one batch of two with
        ``mix_attn_mats_in_batch`` (their cross-attention scores are averaged; no attention LoRA; MC's eps and activations detached)
        or SC alone with gradient and MC alone without gradient and without any LoRA.  The FFN LoRA stays on with probability 0.5."""
        from ...util import collate_dicts, recursive_detach, split_dict
        x_noisy = self.q_sample(x_start, t, noise)
        extra_info = cond_context[2]
        extra_info["capture_ca_activations"] = capture_ca_activations
        extra_info["res_hidden_states_gradscale"] = res_hidden_states_gradscale
        extra_info["img_mask"] = img_mask
        extra_info["normalize_cross_attn"] = normalize_cross_attn
        extra_info["subj_indices"] = subj_indices
        ca_layers_activations = None
        lora = dict(use_attn_lora=use_attn_lora, use_ffn_lora=use_ffn_lora, ffn_lora_adapter_name=ffn_lora_adapter_name)
        fused_uncond = None
        if batch_part_has_grad == "none":
            # Round 5: a gradient-free GUIDED pass is two U-Net calls on the same (x_noisy, t) rows that differ in the prompt rows only -- when
            # nothing else distinguishes them (no image mask, no capture, no attention adapters, no score rewrite: the priming steps and the
            # class-prompt passes of a recon iteration), they are ONE call on 2 B rows: the kernels see twice the rows per launch instead of
            # twice the launches.  The null half also fills the step's uncond_cache, so the step's OTHER guided pass finds it there.
            fuse = (self.batch_cond_with_uncond and cfg_scale > 1 and img_mask is None and not capture_ca_activations and not use_attn_lora
                    and not normalize_cross_attn and not use_ffn_lora and self.uncond_context is not None
                    and not (uncond_cache is not None and uncond_cache.get("x") is x_start and uncond_cache.get("n") is noise and uncond_cache.get("t") is t
                             and uncond_cache.get("u") is (uncond_emb if uncond_emb is not None else self.uncond_context[0])
                             and uncond_cache.get("ck") == (False, None)))
            un2 = None
            if fuse:
                un2 = uncond_emb if uncond_emb is not None else self.uncond_context[0].repeat(x_noisy.shape[0], 1, 1)
                fuse = un2.shape == cond_context[0].shape
            with torch.no_grad():
                if fuse:
                    both = self.apply_model(torch.cat([x_noisy, x_noisy], dim=0), torch.cat([t, t], dim=0),
                                            (torch.cat([cond_context[0], un2.to(cond_context[0].dtype)], dim=0),
                                             list(cond_context[1]) + list(self.uncond_context[1]) * x_noisy.shape[0], extra_info), **lora)
                    noise_pred, fused_uncond = both.chunk(2, dim=0)
                else:
                    noise_pred = self.apply_model(x_noisy, t, cond_context, **lora)
            if capture_ca_activations:
                ca_layers_activations = extra_info["ca_layers_activations"]
        elif batch_part_has_grad == "all":
            noise_pred = self.apply_model(x_noisy, t, cond_context, **lora)
            if capture_ca_activations:
                ca_layers_activations = extra_info["ca_layers_activations"]
        elif batch_part_has_grad == "subject-compos":
            use_ffn_lora = use_ffn_lora and bool(torch.rand(1) < 0.5)
            lora["use_ffn_lora"] = use_ffn_lora

            def context_with(**flags):
                ei = copy.copy(extra_info)
                ei.update(flags)
                return (cond_context[0], cond_context[1], ei)
            # Round 5: the gradient-free passes of this step -- SS, SR, MC when it runs alone, and the four rows of the unconditional pass
            # below -- differ only in the LAST THREE decoder blocks (score rewrites, attention / FFN adapters, capture): everything below
            # them is a function of (x_noisy row, t, prompt row) alone.  That trunk runs ONCE here for all of those rows as one batch
            # (UNetWrapper.precompute_trunks); the separate apply_model calls below keep the reference's call structure and pick their rows'
            # trunk outputs up from extra_info['_trunk_cache'] (modules/diffusionmodules/capture_graph.cached_trunk).  SC (and MC while the
            # scores are mixed) needs gradients through the trunk and runs as before.
            trunk_cache = None
            if self.share_no_grad_trunk and img_mask is None and hasattr(self.model, "precompute_trunks") and x_noisy.shape[0] == 4:
                rows = [0, 2] if mix_sc_mc_attn else [0, 2, 3]
                entries = [(r, x_noisy[r:r + 1], t[r:r + 1], cond_context[0][r:r + 1]) for r in rows]
                if cfg_scale > 1:
                    un = uncond_emb if uncond_emb is not None else self.uncond_context[0].repeat(x_noisy.shape[0], 1, 1)
                    if un.shape[1:] == cond_context[0].shape[1:]:
                        entries += [(("u", r), x_noisy[r:r + 1], t[r:r + 1], un[r:r + 1]) for r in range(4)]
                trunk_cache = self.model.precompute_trunks(entries)
                extra_info["_trunk_cache"] = trunk_cache
            ctx_ss = context_with(normalize_cross_attn=False, mix_attn_mats_in_batch=False)
            ctx_sr = context_with(normalize_cross_attn=normalize_cross_attn, mix_attn_mats_in_batch=False)
            if self.batch_no_grad_instances and not normalize_cross_attn:
                # SS and SR run without gradient under the SAME flags whenever SR's scores are not normalised (always while SC / MC scores
                # are mixed): one pass over the two instances instead of two passes over one -- half the launches, twice the rows per launch
                noise_pred_2 = self.sliced_apply_model(x_noisy, t, ctx_ss, slice_indices=[0, 2], enable_grad=False, **lora)
                noise_pred_ss, noise_pred_sr = noise_pred_2.chunk(2, dim=0)
                if "ca_layers_activations" in ctx_ss[2]:
                    ss_acts, sr_acts = split_dict(ctx_ss[2]["ca_layers_activations"], 2)
                    ctx_ss[2]["ca_layers_activations"], ctx_sr[2]["ca_layers_activations"] = ss_acts, sr_acts
            else:
                noise_pred_ss = self.sliced_apply_model(x_noisy, t, ctx_ss, slice_indices=[0], enable_grad=False, **lora)
                noise_pred_sr = self.sliced_apply_model(x_noisy, t, ctx_sr, slice_indices=[2], enable_grad=False, **lora)
            if mix_sc_mc_attn:
                ctx_sm = context_with(normalize_cross_attn=False, mix_attn_mats_in_batch=True)
                noise_pred_sm = self.sliced_apply_model(x_noisy, t, ctx_sm, slice_indices=[1, 3], enable_grad=True, use_attn_lora=False,
                                                        use_ffn_lora=use_ffn_lora, ffn_lora_adapter_name=ffn_lora_adapter_name)
                noise_pred_sc, noise_pred_mc = noise_pred_sm.chunk(2, dim=0)
                noise_pred_mc = noise_pred_mc.detach()
                sc_acts, mc_acts = split_dict(ctx_sm[2]["ca_layers_activations"], 2)
                mc_acts = recursive_detach(mc_acts)
            else:
                ctx_sc = context_with(normalize_cross_attn=normalize_cross_attn, mix_attn_mats_in_batch=False)
                noise_pred_sc = self.sliced_apply_model(x_noisy, t, ctx_sc, slice_indices=[1], enable_grad=True, **lora)
                sc_acts = ctx_sc[2]["ca_layers_activations"]
                ctx_mc = context_with(normalize_cross_attn=False)
                noise_pred_mc = self.sliced_apply_model(x_noisy, t, ctx_mc, slice_indices=[3], enable_grad=False, use_attn_lora=False,
                                                        use_ffn_lora=False, ffn_lora_adapter_name=ffn_lora_adapter_name)
                mc_acts = ctx_mc[2]["ca_layers_activations"]
            noise_pred = torch.cat([noise_pred_ss, noise_pred_sc, noise_pred_sr, noise_pred_mc], dim=0)
            if capture_ca_activations:
                ca_layers_activations = collate_dicts([ctx_ss[2]["ca_layers_activations"], sc_acts, ctx_sr[2]["ca_layers_activations"], mc_acts])
        else:
            raise ValueError(f"batch_part_has_grad={batch_part_has_grad!r}: 'all', 'none' or 'subject-compos'")
        if cfg_scale > 1:
            uncond_key = uncond_emb if uncond_emb is not None else self.uncond_context[0]      # identity of the null embedding BEFORE the per-call repeat
            if uncond_emb is None:
                uncond_emb = self.uncond_context[0].repeat(x_noisy.shape[0], 1, 1)
            uncond_context = (uncond_emb, self.uncond_context[1] * x_noisy.shape[0], copy.copy(self.uncond_context[2]))
            if extra_info.get("_trunk_cache") is not None and all(("u", r) in extra_info["_trunk_cache"] for r in range(x_noisy.shape[0])):
                uncond_context[2]["_trunk_cache"] = extra_info["_trunk_cache"]                     # this step's shared trunk holds the null-prompt rows too
                uncond_context[2]["_trunk_rows"] = tuple(("u", r) for r in range(x_noisy.shape[0]))
            # ``uncond_cache`` (a dict the caller keeps for ONE denoising step): the unconditional prediction is a function of (x_start, noise, t,
            # the null embedding, the FFN-adapter state) only -- a recon step asks for it twice with the same arguments (after the subject pass
            # and after the class-prompt pass, ddpm.py:1800-1860): the second request takes the first one's tensor (same kernels, same bits)
            # With the FFN adapters live the reference's two null passes are NOT the same computation in training mode (peft draws a fresh
            # lora_dropout mask per pass): the cache is bypassed then and both passes run, as the reference's do.
            ck = (use_ffn_lora, ffn_lora_adapter_name if use_ffn_lora else None)
            if use_ffn_lora:
                uncond_cache = None
            hit = (uncond_cache is not None and uncond_cache.get("ck") == ck and uncond_cache.get("x") is x_start and uncond_cache.get("n") is noise
                   and uncond_cache.get("t") is t and uncond_cache.get("u") is uncond_key)
            if fused_uncond is not None:
                noise_pred_uncond = fused_uncond                # came out of the prompt rows' own call (above)
                if uncond_cache is not None:
                    uncond_cache.update(ck=ck, x=x_start, n=noise, t=t, u=uncond_key, eps=noise_pred_uncond)
            elif hit:
                noise_pred_uncond = uncond_cache["eps"]
            else:
                with torch.no_grad():
                    noise_pred_uncond = self.apply_model(x_noisy, t, uncond_context, use_attn_lora=False, use_ffn_lora=use_ffn_lora,
                                                         ffn_lora_adapter_name=ffn_lora_adapter_name)
                if uncond_cache is not None:
                    uncond_cache.update(ck=ck, x=x_start, n=noise, t=t, u=uncond_key, eps=noise_pred_uncond)
            noise_pred = noise_pred * cfg_scale - noise_pred_uncond * (cfg_scale - 1)
        extra_info.pop("_trunk_cache", None)                                                    # the shared trunk lives for this step only
        x_recon = self.predict_start_from_noise(x_noisy, t=t, noise=noise_pred) if do_pixel_recon else None
        return noise_pred, x_recon, ca_layers_activations

    def prime_x_start_for_comp_prompts(self, subj_context, x_start, noise, num_comp_priming_denoising_steps, cls_subj_mix_ratio, BLOCK_SIZE=1):
        """Compositional priming (reference ddpm.py:1923-1985): pure noise is denoised for a few steps by the PRIMING U-Net
        (``self.comp_distill_priming_unet``, a second frozen U-Net teacher -- the "dual U-Net" of a Stage-2 iteration) with
        classifier-free guidance, one trajectory under the subject-single prompt and one under a mix of the subject-comp and class-comp
        prompt embeddings, sharing timesteps and noise.  Returns the two primed latents [2 * BLOCK_SIZE, 4, h, w]."""
        prompt_emb = subj_context[0]
        x_start_2 = torch.randn_like(x_start)[:BLOCK_SIZE].repeat(2, 1, 1, 1)
        noise_2 = noise[:BLOCK_SIZE].repeat(2, 1, 1, 1)
        t_rear = torch.randint(int(self.num_timesteps * 0.7), int(self.num_timesteps * 0.9), (BLOCK_SIZE,), device=x_start.device)
        t_2 = t_rear.repeat(2)
        subj_single_emb, subj_comp_emb, _, cls_comp_emb = prompt_emb.chunk(4)
        cls_comp_emb_mix = subj_comp_emb * (1 - cls_subj_mix_ratio) + cls_comp_emb * cls_subj_mix_ratio
        with torch.no_grad():
            _, primed_x_starts, _, _ = self.comp_distill_priming_unet(
                self, x_start_2, noise_2, t_2, teacher_context=torch.cat([subj_single_emb, cls_comp_emb_mix], dim=0).detach(),
                negative_context=self.uncond_context[0], num_denoising_steps=num_comp_priming_denoising_steps,
                same_t_noise_across_instances=True)
        return primed_x_starts[-1].to(dtype=x_start.dtype)

    def comp_distill_multistep_denoise(self, x_starts, noises, ts, subj_context, uncond_emb, all_subj_indices_1b=None,
                                       normalize_cross_attn=False, mix_sc_mc_attn=False, cfg_scale=2.5, num_denoising_steps=4,
                                       old_x_starts_mix_ratio=0.3, use_attn_lora=False, use_ffn_lora=False, ffn_lora_adapter_name=None,
                                       BLKS=4, batch_part_has_grad="subject-compos"):
        """The denoising chain of a compositional-distillation iteration (reference ddpm.py:1997-2086): ``num_denoising_steps``
        guided_denoise passes over the four-block batch with activation capture; each step's x0 prediction (detached) seeds the next
        step -- mixed with the caller's earlier x_start when one is given -- at an earlier timestep drawn in
        [t * 0.5^p, t * 0.7^p], p = (steps - 1)^-0.3, with ONE noise / timestep shared by the four blocks.  ``x_starts``, ``noises``,
        ``ts`` are lists that are extended in place, as in the reference.  LoRAs are off on every instance while SC / MC attention
        is mixed."""
        assert num_denoising_steps <= 10
        use_attn_lora = use_attn_lora and (not mix_sc_mc_attn)
        use_ffn_lora = use_ffn_lora and (not mix_sc_mc_attn)
        noise_preds, x_recons, acts_list = [], [], []
        for i in range(num_denoising_steps):
            x_start, t, noise = x_starts[i], ts[i], noises[i]
            noise_pred, x_recon, acts = self.guided_denoise(
                x_start, noise, t, subj_context, uncond_emb, img_mask=None, subj_indices=all_subj_indices_1b,
                normalize_cross_attn=normalize_cross_attn, mix_sc_mc_attn=mix_sc_mc_attn, batch_part_has_grad=batch_part_has_grad,
                do_pixel_recon=True, cfg_scale=cfg_scale, capture_ca_activations=True,
                res_hidden_states_gradscale=self.res_hidden_states_gradscale, use_attn_lora=use_attn_lora, use_ffn_lora=use_ffn_lora,
                ffn_lora_adapter_name=ffn_lora_adapter_name)
            noise_preds.append(noise_pred)
            x_recons.append(x_recon)
            acts_list.append(acts)
            if i < num_denoising_steps - 1:
                if len(noises) <= i + 1:
                    noise = torch.randn_like(x_start.chunk(BLKS)[0]).repeat(BLKS, 1, 1, 1)
                    t0 = t.chunk(BLKS)[0]
                    rand_ts = torch.rand_like(t0.float())
                    p = np.power(num_denoising_steps - 1, -0.3)
                    t_lb, t_ub = t0 * np.power(0.5, p), t0 * np.power(0.7, p)
                    ts.append(((t_ub - t_lb) * rand_ts + t_lb).long().repeat(BLKS))
                    noises.append(noise)
                if len(x_starts) <= i + 1:
                    x_starts.append(x_recon.detach())
                else:
                    x_starts[i + 1] = x_starts[i + 1] * old_x_starts_mix_ratio + x_recon.detach() * (1 - old_x_starts_mix_ratio)
        return noise_preds, x_starts, x_recons, noises, ts, acts_list

    def sliced_apply_model(self, x_noisy, t, cond_context, slice_indices, enable_grad, use_attn_lora=False, use_ffn_lora=False,
                           ffn_lora_adapter_name=None):
        """apply_model on the instances ``slice_indices`` of the batch, with or without gradient (reference ddpm.py:1572-1587)."""
        prompt_emb, prompt_in, extra_info = cond_context
        # an index LIST becomes a device index tensor through a blocking host-to-device copy (the host then waits for everything it has
        # queued, three times per call); the lists used here are arithmetic progressions, which a slice takes without any copy
        idx = slice_indices
        if len(idx) == 1:
            idx = slice(idx[0], idx[0] + 1)
        elif len(idx) > 1 and len({b - a for a, b in zip(idx, idx[1:])}) == 1 and idx[1] > idx[0]:
            idx = slice(idx[0], idx[-1] + 1, idx[1] - idx[0])
        ctx = (prompt_emb[idx], [prompt_in[i] for i in slice_indices], extra_info)
        if extra_info is not None and extra_info.get("_trunk_cache") is not None:
            extra_info["_trunk_rows"] = tuple(slice_indices)                                    # which rows of the step's shared trunk this call is (guided_denoise)
        with torch.set_grad_enabled(enable_grad):
            return self.apply_model(x_noisy[idx], t[idx], ctx, use_attn_lora=use_attn_lora, use_ffn_lora=use_ffn_lora,
                                    ffn_lora_adapter_name=ffn_lora_adapter_name)

    def prepare_unet_teacher_context(self, subj_context, uncond_context, BLOCK_SIZE, id2img_prompt_embs, id2img_neg_prompt_embs,
                                     img_prompt_prefix_embs, unet_teacher_types, encoders_num_id_vecs, p_unet_teacher_uses_cfg,
                                     unet_distill_uses_comp_prompt):
        """The teacher's prompt embeddings (reference ddpm.py:2885-2980) for the Arc2Face teacher, the only teacher of the Stage-1
        yaml: ["photo of a" prefix ++ ID image-prompt embeddings] per instance, [BS, 4 + 16, 768]; when the teacher may use
        classifier-free guidance the unconditional embedding cut to the same length is stacked below it on dim 0."""
        if list(unet_teacher_types) != ["arc2face"] or encoders_num_id_vecs is not None:
            raise NotImplementedError(f"unet_teacher_types={unet_teacher_types}: only the single Arc2Face teacher is built "
                                      "(consistentID / unet_ensemble need the external ConsistentID package)")
        teacher_context = torch.cat([img_prompt_prefix_embs.repeat(BLOCK_SIZE, 1, 1), id2img_prompt_embs], dim=1)
        if p_unet_teacher_uses_cfg > 0:
            neg = uncond_context[0][:, :teacher_context.shape[1]].repeat(BLOCK_SIZE, 1, 1)
            teacher_context = torch.cat([teacher_context, neg.to(teacher_context.dtype)], dim=0)
        return teacher_context

    def calc_unet_distill_loss(self, x_start, noise, subj_context, teacher_context, img_mask, fg_mask,
                               num_unet_denoising_steps, t=None, recon_bg_pixel_weight=0, presampled=None):
        """Teacher multi-step epsilon targets, student epsilon per step, fg-masked MSE, sum / sqrt(steps)
        (reference ddpm.py:2984-3184 without the logging decodes / pure-noise priming, which SURVEY.md section 3.1
        identifies as pure overhead for throughput).  FFN-LoRA 'unet_distill' is SURVEY.md 8f rank 1."""
        if t is None:
            t = torch.randint(int(self.num_timesteps * 0.7), int(self.num_timesteps * 0.9), (x_start.shape[0],),
                              device=x_start.device).long()
        with torch.no_grad():
            t_preds, t_x_starts, t_noises, all_t = self.unet_teacher(self, x_start, noise, t, teacher_context,
                                                                     num_denoising_steps=num_unet_denoising_steps,
                                                                     force_uses_cfg=False, presampled=presampled)
        losses = []
        n, bs = num_unet_denoising_steps, x_start.shape[0]
        if self.batch_student_steps and n > 1:
            # The student's per-step passes are independent once the teacher's trajectory exists (the reference loops over
            # them, ddpm.py:3103-3176): run them as ONE U-Net call on the steps*BS batch -- same arithmetic per sample
            # (GroupNorm, attention and the masked MSE are per instance), larger GEMM M, one backward node.
            ctx, prompts, extra = subj_context
            cat_context = (ctx.repeat(n, 1, 1), list(prompts) * n, extra)
            pred_all, _, _ = self.guided_denoise(torch.cat([x.to(x_start.dtype) for x in t_x_starts[:n]]),
                                                 torch.cat([z.to(x_start.dtype) for z in t_noises[:n]]), torch.cat(list(all_t[:n])),
                                                 cat_context, img_mask=None, batch_part_has_grad="all", do_pixel_recon=True,
                                                 cfg_scale=self.unet_teacher.cfg_scale,
                                                 res_hidden_states_gradscale=self.res_hidden_states_gradscale,
                                                 # ** Always enable ffn LoRAs on unet distillation (ddpm.py:3130-3134) when they exist
                                                 use_ffn_lora=self.model.ffn_lora is not None, ffn_lora_adapter_name="unet_distill")
            preds = pred_all.split(bs)
        else:
            preds = []
            for s in range(n):
                noise_pred_s, _, _ = self.guided_denoise(t_x_starts[s].to(x_start.dtype), t_noises[s].to(x_start.dtype), all_t[s],
                                                         subj_context, img_mask=None, batch_part_has_grad="all",
                                                         do_pixel_recon=True, cfg_scale=self.unet_teacher.cfg_scale,
                                                         res_hidden_states_gradscale=self.res_hidden_states_gradscale,
                                                 # ** Always enable ffn LoRAs on unet distillation (ddpm.py:3130-3134) when they exist
                                                 use_ffn_lora=self.model.ffn_lora is not None, ffn_lora_adapter_name="unet_distill")
                preds.append(noise_pred_s)
        for s in range(n):
            loss_s, _ = calc_recon_loss(F.mse_loss, preds[s], t_preds[s].to(preds[s].dtype), img_mask, fg_mask,
                                        instance_weights=None, fg_pixel_weight=1, bg_pixel_weight=recon_bg_pixel_weight)
            losses.append(loss_s)
        return sum(losses) / np.sqrt(num_unet_denoising_steps)
