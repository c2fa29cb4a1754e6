"""Stage-1 (Arc2Face -> AdaFace) U-Net distillation training step, the data-parallel hot loop of BASELINE configs 3/4
(reference ``DDPM.training_step`` ddpm.py:434-503, ``shared_step``/``forward`` :936-1355 for the ``do_unet_distill``
iteration type, ``p_losses`` :2355-2368, ``configure_optimizers`` :3855-4020; Lightning ``strategy="ddp"``,
``accumulate_grad_batches: 2`` main.py:618,911-915).

What one micro-batch does (everything below the Python control flow is a HIP kernel launch through the C ABI):

  face IDs [B,512] --Arc2Face CLIP (frozen, no grad)--> image-prompt embs [B,16,768]                     (F2)
      --SubjBasisGenerator CLIP (TRAINABLE, autograd nodes)--> AdaFace token embs [B,16,768]             (F3-F6)
      --patched into the subject slots of the prompt's token embeddings--> frozen CLIP text encoder
        (activation-gradient only) --> subj_context [B,T,768]                                           (8f rank 2, minimal)
  teacher: "photo of a" prefix (4 tokens) ++ image-prompt embs -> multi-step Arc2Face U-Net (no grad)    (D5)
  student: per step eps-prediction of the frozen SD-1.5 U-Net under subj_context, fg-masked MSE x 8      (D4, D6)
  backward: whole-U-Net activation-gradient node -> text encoder dgrad -> SubjBasisGenerator weights
  every `accumulate_grad_batches`-th micro-batch: bucketed RCCL all-reduce overlapped with the backward
  (GradReducer), gradient clipping by value (+-0.01, the yaml's Lightning setting; once per optimizer step on the
  accumulated rank-averaged gradients, as Lightning's automatic optimization does), unscale, cautious AdamW on the
  flat fp32 arena, warm-up/cosine LR.

Like the reference's do_unet_distill iterations the batch is cut to HALF_BS = ceil(BS / steps) instances when
`num_unet_denoising_steps` > 1 (ddpm.py:1283-1311) and the step count cycles 2,3,4 deterministically so every rank runs
the same graph (ddpm.py:1266-1270).

fp16 activation gradients need loss scaling (the reference trains with fp16 autocast + GradScaler through Lightning
`precision: 16`): `LossScaler` is the usual dynamic scheme (x2 every `growth_interval` clean steps, /2 and skip on
inf/nan), with the overflow flag all-reduced so ranks skip together."""
import contextlib
import math

import torch
import torch.distributed as dist

from .. import ops
from ..adaface.subj_basis_generator import template_ids
from ..distributed import GradReducer
from .c_adamw import AdamW as CAdamW
from .modules.arcface_wrapper import to_device_async
from .modules.lr_scheduler import LambdaWarmUpCosineScheduler
from .util import set_seed_per_rank_and_batch


class LossScaler:
    def __init__(self, init_scale=2.0 ** 14, growth_interval=200, min_scale=1.0, max_scale=2.0 ** 24):
        self.scale = float(init_scale)
        self.growth_interval = growth_interval
        self.min_scale, self.max_scale = min_scale, max_scale
        self._good = 0

    def update(self, overflow: bool):
        if overflow:
            self.scale = max(self.min_scale, self.scale / 2)
            self._good = 0
        else:
            self._good += 1
            if self._good % self.growth_interval == 0:
                self.scale = min(self.max_scale, self.scale * 2)


class DistillTrainer:
    """Owns the optimizer state, the gradient reducer and the iteration bookkeeping; the models are passed in.

    ldm            LatentDiffusion with `.unet_teacher` set (frozen student U-Net + Arc2Face teacher U-Net)
    id2ada         Arc2Face_ID2AdaPrompt (frozen Arc2Face encoder + trainable SubjBasisGenerator)
    text_encoder   CLIPTextModelWrapper holding the frozen SD-1.5 text encoder weights
    """

    unet_distill_weight = 8                                                     # ddpm.py:2367

    def __init__(self, ldm, id2ada, text_encoder, base_lr=2e-6, batch_size=4, accumulate_grad_batches=2,
                 betas=(0.9, 0.995), eps=1e-6, weight_decay=0.0, lora_weight_decay=0.02, warm_up_steps=500, max_decay_steps=60000,
                 bucket_bytes=32 << 20, loss_scaler=None, prompt_len=77, subj_slot=4, process_group=None,
                 gradient_clip_val=0.01, gradient_clip_algorithm="value", p_gen_rand_id_for_id2img=0.0, p_perturb_face_id_embs=0.0,
                 perturb_face_id_embs_std_range=(0.3, 0.6), stage=1, use_graphs=False):
        self.ldm, self.id2ada, self.text_encoder = ldm, id2ada, text_encoder
        self.iter_type = "comp_distill" if stage == 2 else "unet_distill"
        self.graph_segments = []
        if use_graphs:
            # hipGraph replay of the fixed-shape segments of the micro-batch (graphs.py): the teacher's multi-step forward and the
            # student U-Net's forward / backward walks -- ~2,400 of the ~4,100 Python-issued launches of a Stage-1 micro-batch
            from ..graphs import GraphedSegment
            unet = ldm.model.diffusion_model
            unet.train_graphs = (GraphedSegment("student.forward"), GraphedSegment("student.backward"))
            self.graph_segments += list(unet.train_graphs)
            if getattr(ldm, "unet_teacher", None) is not None:
                ldm.unet_teacher.graphs = GraphedSegment("teacher.multistep")
                self.graph_segments.append(ldm.unet_teacher.graphs)
            if getattr(ldm, "comp_distill_priming_unet", None) is not None:      # Stage 2: the priming U-Net's guided multi-step forward
                ldm.comp_distill_priming_unet.graphs = GraphedSegment("priming.multistep")
                self.graph_segments.append(ldm.comp_distill_priming_unet.graphs)
        for p in text_encoder.parameters():
            p.requires_grad_(False)
        self.accum = accumulate_grad_batches
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.pg = process_group
        # lr = accumulate_grad_batches * ngpu * bs * base_lr (main.py:911-915)
        self.learning_rate = accumulate_grad_batches * self.world * batch_size * base_lr
        params = [p for p in id2ada.subj_basis_generator.parameters() if p.requires_grad]
        groups = [{"params": params, "weight_decay": weight_decay}]
        # second group: the U-Net's `unet_distill` FFN DoRA adapters when they exist (embedding_manager.optimized_parameters gives
        # them their own weight decay, ddpm.py:143 lora_weight_decay = 0.02); the other adapter names are not touched in Stage 1
        self.ffn_lora = getattr(ldm.model, "ffn_lora", None)
        lora_params = []
        if self.ffn_lora is not None:
            adapter = "comp_distill" if stage == 2 else "unet_distill"      # the adapter each stage trains (ddpm.py:3130-3134, 2167-2173)
            for n, p in self.ffn_lora.named_parameters():
                p.requires_grad_(n.startswith(f"adapters.{adapter}."))
            lora_params += [p for p in self.ffn_lora.parameters() if p.requires_grad]
        if stage == 2:
            # Stage 2 also trains the attention DoRA adapters of the three captured layers and their score scale factors
            # (diffusers_attn_lora_capture.py:511-524 puts both into the optimised unet_lora_modules)
            if getattr(ldm.model, "attn_lora", None) is not None:
                lora_params += list(ldm.model.attn_lora.parameters())
            lora_params += list(ldm.model.cross_attn_scale_factors.values())
        else:
            for p in ldm.model.cross_attn_scale_factors.values():
                p.requires_grad_(False)
        if lora_params:
            groups.append({"params": lora_params, "weight_decay": lora_weight_decay})
        self.optimizer = CAdamW(groups, lr=self.learning_rate, betas=betas, eps=eps, weight_decay=weight_decay)
        self.arenas = [self.optimizer.arena(i) for i in range(len(groups))]
        self.arena = self.arenas[0]
        self.reducer = GradReducer(self.arenas, bucket_bytes=bucket_bytes, process_group=process_group)
        self.lr_lambda = LambdaWarmUpCosineScheduler(warm_up_steps, 0.1, 1.0, 0.01, max_decay_steps)
        self.scaler = loss_scaler or LossScaler()
        self.global_step = 0                                                   # optimizer steps
        self.unet_distill_iters_count = 0
        self.skipped_steps = 0
        self.prompt_len, self.subj_slot = prompt_len, subj_slot
        # Lightning trainer settings of the reference yaml (v1-distill-arc2face-ada.yaml:150-152), applied once per optimizer step
        # (automatic optimization)
        if gradient_clip_algorithm != "value":
            raise NotImplementedError("only gradient_clip_algorithm='value' (the reference's setting) is built")
        self.gradient_clip_val = gradient_clip_val
        # per-iteration input variants of a U-Net distillation iteration (ddpm.py:1131-1137, 1152-1169, 1222-1261); the reference's yaml
        # uses 0 / 0.2 / [0.3, 0.6] (v1-distill-arc2face-ada.yaml:36-42).  Off (0, 0) by default here so that a batch is used as given.
        self.p_gen_rand_id_for_id2img = p_gen_rand_id_for_id2img
        self.p_perturb_face_id_embs = p_perturb_face_id_embs
        self.perturb_face_id_embs_std_range = tuple(perturb_face_id_embs_std_range)
        self.iter_flags = {}
        self._bad = None                                                        # device flag: a non-finite gradient was seen

    # ------------------------------------------------------------------ conditioning
    def prompt_ids(self, bs, device):
        """"a photo of z_0_0 ... z_0_15" with the 16 subject tokens at slots subj_slot.. (adaface_wrapper.py:491-532
        rewrites the prompt the same way); the placeholder ids are irrelevant because their embeddings are replaced."""
        n_id = self.id2ada.subj_basis_generator.N_ID
        return template_ids(["a", "photo", "of"] + [","] * n_id, self.prompt_len, device).repeat(bs, 1)

    def subject_prompt(self, subject_string="z", filler=","):
        """"a photo of z, , , ..." -- the subject token followed by K - 1 filler tokens, as the reference's dataset writes its
        captions (ldm/data/personalized.py) so that the embedding manager finds K slots."""
        n_id = self.id2ada.subj_basis_generator.N_ID
        return "a photo of " + subject_string + (filler + " ") * (n_id - 1)

    def get_text_conditioning(self, adaface_embs, input_ids=None):
        """AdaFace token embeddings -> prompt embeddings [B, T, 768] through the frozen text encoder (the minimal form of
        reference get_text_conditioning ddpm.py:739-853 + EmbeddingManager token patching embedding_manager.py:236-421)."""
        bs = adaface_embs.shape[0]
        te = self.text_encoder
        ids = self.prompt_ids(bs, adaface_embs.device) if input_ids is None else input_ids
        s, n = self.subj_slot, adaface_embs.shape[1]
        with torch.no_grad():
            tok = te(input_ids=ids, return_token_embs=True).to(adaface_embs.dtype)
        tok = torch.cat([tok[:, :s], adaface_embs, tok[:, s + n:]], dim=1)
        return te(input_ids=ids, input_token_embs=tok)[0]

    def teacher_context(self, id2img_prompt_embs):
        """[BS, 4 + 16, 768]: "photo of a" prefix embeddings ++ image-prompt embeddings (ddpm.py:2925-2931)."""
        enc = self.id2ada.text_to_image_prompt_encoder
        bs = id2img_prompt_embs.shape[0]
        if getattr(self, "_prefix", None) is None:
            with torch.no_grad():
                ids = template_ids(["photo", "of", "a"], 22, id2img_prompt_embs.device)
                self._prefix = enc(input_ids=ids)[0][:, :4].to(id2img_prompt_embs.dtype)
        return torch.cat([self._prefix.repeat(bs, 1, 1), id2img_prompt_embs], dim=1)

    # ------------------------------------------------------------------ per-iteration input variants
    def select_iteration_inputs(self, batch):
        """The data-dependent head of a U-Net distillation iteration (reference ddpm.py:1131-1169, 1222-1245), consuming the global
        torch RNG in the reference's order.  With probability ``p_gen_rand_id_for_id2img`` the iteration distils on RANDOM face IDs
        (``randn(BS, 512)``), a random ``x_start`` and no masks; with probability ``p_perturb_face_id_embs`` the batch becomes BS copies
        of its first instance whose image-prompt embeddings [1:] are perturbed later (``perturb_img_prompt_embs``).  Returns a new
        batch dict; ``self.iter_flags`` records the draws."""
        from .util import select_and_repeat_instances
        out = dict(batch)
        BS = batch["x_start"].shape[0]
        flags = {"gen_rand_id_for_id2img": False, "same_subject_in_batch": False, "perturb_face_id_embs": False}
        if self.p_gen_rand_id_for_id2img > 0 and bool(torch.rand(1) < self.p_gen_rand_id_for_id2img):
            flags["gen_rand_id_for_id2img"] = True
            x = batch["x_start"]
            out["face_id_embs"] = torch.randn(BS, batch["face_id_embs"].shape[-1], device=x.device)
            out["fg_mask"], out["img_mask"] = None, None
            out["x_start"] = torch.randn_like(x)
        flags["perturb_face_id_embs"] = bool(torch.rand(1) < self.p_perturb_face_id_embs)
        if flags["perturb_face_id_embs"]:
            flags["same_subject_in_batch"] = True
            keys = [k for k in ("x_start", "fg_mask", "img_mask", "face_id_embs") if out.get(k) is not None]
            reps = select_and_repeat_instances(slice(0, 1), BS, *[out[k] for k in keys])
            out.update(dict(zip(keys, reps)))
        self.iter_flags = flags
        return out

    def perturb_img_prompt_embs(self, id2img_prompt_embs):
        """Keep the first instance's image-prompt embeddings, add norm-preserving Gaussian noise of relative std ~ U(range) to the
        others (reference ddpm.py:1247-1261): neighbours of the subject in embedding space as extra distillation targets."""
        from .util import anneal_perturb_embedding
        if not self.iter_flags.get("perturb_face_id_embs") or id2img_prompt_embs.shape[0] < 2:
            return id2img_prompt_embs
        rest = anneal_perturb_embedding(id2img_prompt_embs[1:], 0, self.perturb_face_id_embs_std_range, None, perturb_prob=1,
                                        perturb_std_is_relative=True, keep_norm=True)
        return torch.cat([id2img_prompt_embs[:1], rest], dim=0)

    # ------------------------------------------------------------------ one micro-batch
    def shared_step(self, batch, num_unet_denoising_steps=None, t=None, presampled=None):
        """batch: dict with 'x_start' [B,4,h,w] latents, 'face_id_embs' [B,512], optional 'fg_mask'/'img_mask' [B,1,h,w]
        and 'noise'.  Returns the (unscaled) loss tensor."""
        if self.p_gen_rand_id_for_id2img > 0 or self.p_perturb_face_id_embs > 0:
            batch = self.select_iteration_inputs(batch)
        x_start = batch["x_start"]
        BS = x_start.shape[0]
        steps = num_unet_denoising_steps or (self.unet_distill_iters_count % 3 + 2)      # ddpm.py:1270
        half = math.ceil(BS / steps) if steps > 1 else BS                                # == arange(BS).chunk(steps)[0]
        sel = slice(0, half)
        x_start = x_start[sel]
        fg_mask = batch.get("fg_mask")
        img_mask = batch.get("img_mask")
        fg_mask = None if fg_mask is None else fg_mask[sel]
        img_mask = None if img_mask is None else img_mask[sel]
        noise = batch["noise"][sel] if "noise" in batch else torch.randn_like(x_start)
        with torch.no_grad():
            _, _, id2img = self.id2ada.get_img_prompt_embs(batch["face_id_embs"][sel], id_batch_size=half)[:3]
        id2img = self.perturb_img_prompt_embs(id2img.float())
        em = getattr(self.ldm, "embedding_manager", None)
        if em is not None:
            # the reference's conditioning path (ddpm.py:739-853): prompts -> hooked text encoder, the embedding manager generates
            # the ada embeddings inside the embedding step and patches them over "z, , , ..." (last-2-layers skip weights included)
            if em.id2ada_prompt_encoder is not self.id2ada:
                raise RuntimeError("the embedding manager must wrap the trainer's ID->prompt encoder (its parameters are the ones optimised)")
            prompts = batch.get("caption")
            prompts = list(prompts[sel]) if prompts is not None else [self.subject_prompt()] * half
            self.ldm.iter_flags["do_unet_distill"] = True
            cond = self.ldm.get_text_conditioning(prompts, subj_id2img_prompt_embs=id2img, text_conditioning_iter_type="unet_distill_iter",
                                                  real_batch_size=half)
        else:
            ada = self.id2ada.subj_basis_generator(id2img, out_id_embs_cfg_scale=self.id2ada.out_id_embs_cfg_scale, is_face=True)
            ctx = self.get_text_conditioning(ada.float())
            cond = (ctx, ["a photo of z"] * half, {})
        loss = self.ldm.calc_unet_distill_loss(x_start, noise, cond, self.teacher_context(id2img), img_mask, fg_mask, steps,
                                               t=t, presampled=presampled)
        return loss * self.unet_distill_weight

    # ------------------------------------------------------------------ Stage 2: one compositional-distillation micro-batch
    num_comp_distill_denoising_steps = 4          # reference ctor defaults (ddpm.py:105-107, 84)
    max_num_comp_priming_denoising_steps = 4
    cls_subj_mix_ratio = 0.6
    comp_iters_count = 0

    def comp_prompt_context(self, ada_embs, comp_words=("of", "a", "person")):
        """The four prompts of a compositional iteration for ONE subject -- subject-single "a photo of z", subject-comp
        "a photo of z <comp>", subject-comp-rep (the compositional part repeated) and class-comp "a photo of person <comp>" -- as
        [4, T, 768] prompt embeddings with the AdaFace embeddings patched into the subject slots of the first three, plus the
        subject-token indices of block 0 and the [4, T, 1] embedding / padding masks the Stage-2 losses consume (the reference
        builds these through its dataset's prompt lists and the EmbeddingManager, ddpm.py:1392-1558)."""
        dev = ada_embs.device
        n_id, T = self.id2ada.subj_basis_generator.N_ID, self.prompt_len
        subj = ["a", "photo", "of"] + [","] * n_id
        word_lists = [subj, subj + list(comp_words), subj + list(comp_words) * 2, ["a", "photo", "of", "person"] + [","] * (n_id - 1) + list(comp_words)]
        ids = torch.cat([template_ids(w, T, dev) for w in word_lists])
        s = self.subj_slot
        with torch.no_grad():
            tok = self.text_encoder(input_ids=ids, return_token_embs=True).float()
        ada = ada_embs.float()
        tok = torch.cat([torch.cat([tok[:3, :s], ada.expand(3, -1, -1), tok[:3, s + n_id:]], dim=1), tok[3:]], dim=0)
        ctx = self.text_encoder(input_ids=ids, input_token_embs=tok)[0]
        lens = to_device_async(torch.tensor([len(w) + 2 for w in word_lists]), dev)
        pos = torch.arange(T, device=dev)[None, :]
        emb_mask = ((pos >= 1) & (pos < (lens - 1)[:, None])).float().unsqueeze(2)          # real tokens (no BOS / EOS / padding)
        pad_mask = (pos >= lens[:, None]).float().unsqueeze(2)
        subj_indices_1b = (torch.zeros(n_id, dtype=torch.long, device=dev), torch.arange(s, s + n_id, device=dev))
        return ctx, ["ss", "sc", "sc-rep", "mc"], subj_indices_1b, emb_mask, pad_mask

    def comp_distill_step(self, batch, attn_aug=None):
        """do_comp_feat_distill iteration (reference ddpm.py:2371-2483): BLOCK_SIZE 1; four prompts; latents primed from pure noise by
        the priming U-Net; ``num_comp_distill_denoising_steps`` subject-compos passes of the student with activation capture (only
        the subject-comp instance carries gradients; SC / MC scores mixed or subject scores normalised, p = 0.5 each, :940-952);
        the x0 predictions of every step decoded for the face detector; ``LatentDiffusion.calc_comp_feat_distill_loss`` on all of it.
        Faces come from ``ldm.arcface`` (modules/arcface_wrapper.ArcFaceWrapper around the caller's detector); with
        ``ldm.arcface_align_loss_weight = 0`` no face is looked for and, as in the reference, every face-gated term is zero."""
        ldm = self.ldm
        x_start = batch["x_start"][:1]
        with torch.no_grad():
            _, _, id2img = self.id2ada.get_img_prompt_embs(batch["face_id_embs"][:1], id_batch_size=1)[:3]
        ada = self.id2ada.subj_basis_generator(id2img.float(), out_id_embs_cfg_scale=self.id2ada.out_id_embs_cfg_scale, is_face=True)
        ctx, prompts, subj_1b, emb_mask, pad_mask = self.comp_prompt_context(ada)
        if attn_aug is None:
            attn_aug = ("normalize_cross_attn", "mix_sc_mc_attn")[int(torch.multinomial(torch.tensor([0.5, 0.5]), 1))]
        steps_prime = self.comp_iters_count % 2 - 1 + self.max_num_comp_priming_denoising_steps
        self.comp_iters_count += 1
        ldm.comp_iters_count = self.comp_iters_count
        noise = torch.randn_like(x_start)
        primed = ldm.prime_x_start_for_comp_prompts((ctx, prompts, {}), x_start, noise, steps_prime, 0.5 + self.cls_subj_mix_ratio / 2)
        noise = torch.randn_like(x_start).repeat(4, 1, 1, 1)
        xs, xc = primed.chunk(2)
        x_start_primed = torch.cat([xs, xc, xc, xc], dim=0)
        uncond_emb = ldm.uncond_context[0].repeat(4, 1, 1)
        t = torch.randint(int(ldm.num_timesteps * 0.45), int(ldm.num_timesteps * 0.65), (1,), device=x_start.device).repeat(4)
        has_attn_lora, has_ffn_lora = ldm.model.attn_lora is not None, ldm.model.ffn_lora is not None
        S = self.num_comp_distill_denoising_steps
        ldm.num_comp_distill_denoising_steps = S
        noise_preds, x_starts, x_recons, noises, ts, acts = ldm.comp_distill_multistep_denoise(
            [x_start_primed], [noise], [t], (ctx, prompts, {}), uncond_emb=uncond_emb, all_subj_indices_1b=subj_1b,
            normalize_cross_attn=attn_aug == "normalize_cross_attn", mix_sc_mc_attn=attn_aug == "mix_sc_mc_attn", cfg_scale=2.5,
            num_denoising_steps=S, old_x_starts_mix_ratio=0, use_attn_lora=has_attn_lora,
            use_ffn_lora=has_ffn_lora, ffn_lora_adapter_name="comp_distill", batch_part_has_grad="subject-compos")
        pixels = None
        if ldm.arcface_align_loss_weight > 0:
            if ldm.arcface is None or ldm.first_stage_model is None:
                raise RuntimeError("a compositional-distillation iteration with arcface_align_loss_weight > 0 looks for faces in the decoded x0 "
                                   "predictions: set ldm.arcface (modules/arcface_wrapper.ArcFaceWrapper around your face detector) and "
                                   "instantiate the first-stage decoder, or set ldm.arcface_align_loss_weight = 0")
            # ddpm.py:2454-2457 decodes all four blocks of every step -- for its image logger (:2459-2465); the loss reads the
            # subject-single block only (ddpm_losses.calc_comp_feat_distill_loss), and this trainer logs no images: decode what is consumed
            # (4 images instead of 16) unless a logger asks for the rest
            blocks = x_recons if self.decode_all_blocks_for_logging else [x.chunk(4)[0] for x in x_recons]
            pixels = ldm.decode_first_stage(torch.cat(blocks, dim=0).detach()).chunk(S)
        ss_context = (ctx.chunk(4)[0], prompts[:1], {})
        self.mon_loss_dict = {}
        return ldm.calc_comp_feat_distill_loss(self.mon_loss_dict, "train", x_start, x_starts, x_recons, pixels, noise_preds, noises, ts, acts, subj_1b,
                                               ss_context, ldm.uncond_context[0], emb_mask, pad_mask, 1, ldm.sc_fg_face_suppress_mask_shrink_ratio,
                                               use_attn_lora=has_attn_lora, use_ffn_lora=has_ffn_lora)

    decode_all_blocks_for_logging = False        # comp_distill_step: also decode the SC / SR / MC x0 predictions of every step (nothing here reads them)

    # ------------------------------------------------------------------ do_normal_recon iteration (ddpm.py:2296-2352, 2593-2883)
    p_normal_recon_on_pure_noise = 0.4            # reference ctor defaults (ddpm.py:116, 126-130)
    unet_uses_attn_lora = True
    recon_uses_ffn_lora = False
    comp_uses_ffn_lora = True

    def recon_prompt_context(self, ada_embs, bs):
        """(subject-single context "a photo of z", class-single context "a photo of person") for ``bs`` instances and the subject-token
        indices of the whole batch (the reference reads them from the EmbeddingManager's placeholder2indices, ddpm.py:2302-2303)."""
        dev = ada_embs.device
        n_id, T, s = self.id2ada.subj_basis_generator.N_ID, self.prompt_len, self.subj_slot
        ctx = self.get_text_conditioning(ada_embs.float())
        cls_ids = template_ids(["a", "photo", "of", "person"] + [","] * (n_id - 1), T, dev).repeat(bs, 1)
        with torch.no_grad():
            cls_ctx = self.text_encoder(input_ids=cls_ids)[0]
        subj = (torch.arange(bs, device=dev).repeat_interleave(n_id), torch.arange(s, s + n_id, device=dev).repeat(bs))
        return ctx, cls_ctx.to(ctx.dtype), subj

    def normal_recon_step(self, batch, on_pure_noise=None):
        """do_normal_recon iteration: the subject prompt must reconstruct the input images' noise (``calc_normal_recon_loss``), on the
        images themselves or -- with probability ``p_normal_recon_on_pure_noise`` (ddpm.py:1166-1169) -- from pure noise after four
        priming steps; attention LoRAs on half of the time, the FFN adapter per ``recon_uses_ffn_lora`` (:2305-2326)."""
        ldm = self.ldm
        # the recon iteration's losses are all face-gated (ddpm.py:2702-2790): without the face pipeline, or with its weight at 0, the reference
        # reaches a stack of an empty list -- refuse with a message instead
        if getattr(ldm, "arcface", None) is None or getattr(ldm, "first_stage_model", None) is None:
            raise RuntimeError("normal_recon_step: the do_normal_recon iteration needs ldm.arcface (ArcFaceWrapper) and ldm.first_stage_model (the VAE)")
        if not ldm.arcface_align_loss_weight > 0:
            raise RuntimeError("normal_recon_step: arcface_align_loss_weight must be > 0 for do_normal_recon iterations (every recon term is gated on detected faces)")
        x_start = batch["x_start"]
        BS = x_start.shape[0]
        fg_mask = batch.get("fg_mask")
        fg_mask = torch.ones(BS, 1, *x_start.shape[2:], device=x_start.device) if fg_mask is None else fg_mask
        img_mask = batch.get("img_mask")
        with torch.no_grad():
            _, _, id2img = self.id2ada.get_img_prompt_embs(batch["face_id_embs"], id_batch_size=BS)[:3]
        ada = self.id2ada.subj_basis_generator(id2img.float(), out_id_embs_cfg_scale=self.id2ada.out_id_embs_cfg_scale, is_face=True)
        ctx, cls_ctx, subj = self.recon_prompt_context(ada, BS)
        if on_pure_noise is None:
            on_pure_noise = bool(torch.rand(1) < self.p_normal_recon_on_pure_noise)
        extra = {}
        prompts = ["a photo of z"] * BS
        noise = batch["noise"] if "noise" in batch else torch.randn_like(x_start)
        if on_pure_noise:
            attn_lora, ffn_lora, adapter, priming = False, False, "recon_loss", 4
        else:
            attn_lora = self.unet_uses_attn_lora and ldm.model.attn_lora is not None and torch.rand(1).item() < 0.5
            ffn_lora = self.recon_uses_ffn_lora and ldm.model.ffn_lora is not None
            adapter = "comp_distill" if (self.comp_uses_ffn_lora and torch.randn(1).item() < 0.25) else "recon_loss"
            priming = 0
        do_adv = bool(torch.rand(1) < ldm.p_do_adv_attack_when_recon_on_images) and not on_pure_noise
        self.mon_loss_dict = {}
        return ldm.calc_normal_recon_loss(self.mon_loss_dict, "train", ldm.num_recon_denoising_steps, priming, x_start, noise, (ctx, prompts, extra),
                                          (cls_ctx, prompts, extra), img_mask, fg_mask, subj, ldm.recon_bg_pixel_weight, on_pure_noise, attn_lora,
                                          ffn_lora, adapter, do_adv, min(BS, 2))

    # ------------------------------------------------------------------ which iteration a micro-batch is (ddpm.py:451-470)
    comp_distill_iter_gap = 0          # reference default 5 (ddpm.py:82) when Stage 2 is on; 0 = never, as DDPM treats <= 0
    unet_distill_iter_gap = 0          # v1-distill-arc2face-ada.yaml:28 sets 2 (every 2nd non-comp micro-batch distils, the others reconstruct);
                                       # 0 here = this trainer's historical behaviour: the ``stage`` argument fixes one iteration type
    non_comp_iters_count = 0
    normal_recon_iters_count = 0

    def schedule_iteration(self):
        """The reference's iteration typing (ddpm.py:451-470): a micro-batch whose OPTIMIZER step is a multiple of
        ``comp_distill_iter_gap`` is compositional distillation (so both micro-batches of such an accumulation window are); every other
        micro-batch bumps ``non_comp_iters_count``, and every ``unet_distill_iter_gap``-th of THOSE is U-Net distillation, the rest
        normal recon -- with the yaml's gap of 2 recon and distillation micro-batches alternate, one of each per accumulation window.
        With both gaps 0 the constructor's ``stage`` decides (iter_type)."""
        if self.comp_distill_iter_gap <= 0 and self.unet_distill_iter_gap <= 0:
            return self.iter_type
        if self.comp_distill_iter_gap > 0 and self.global_step % self.comp_distill_iter_gap == 0:
            return "comp_distill"
        self.non_comp_iters_count += 1
        if self.unet_distill_iter_gap > 0 and self.non_comp_iters_count % self.unet_distill_iter_gap == 0:
            return "unet_distill"
        self.normal_recon_iters_count += 1
        return "normal_recon"

    iter_type = "unet_distill"        # or "comp_distill" (Stage 2)

    def set_graphs_enabled(self, on: bool):
        """Switch the captured segments between replay and eager launches (eager is needed when launches are to be timed one by one)."""
        for g in self.graph_segments:
            g.enabled = bool(on)

    def training_step(self, batch, batch_idx, epoch=0, **kw):
        """One micro-batch: forward, scaled backward (with the gradient exchange overlapped on the last micro-batch of an
        accumulation window), and on window end: unscale, optimizer step, LR schedule.  Returns the loss (detached)."""
        set_seed_per_rank_and_batch(self.rank, epoch, batch_idx)                        # ddpm.py:442
        self.unet_distill_iters_count += 1
        last = (batch_idx + 1) % self.accum == 0
        sync = contextlib.nullcontext() if last else self.reducer.no_sync()
        kind = self.schedule_iteration()
        self.last_iter_type = kind
        loss = {"comp_distill": self.comp_distill_step, "normal_recon": self.normal_recon_step, "unet_distill": self.shared_step}[kind](batch, **kw)
        with sync:
            (loss * (self.scaler.scale / self.accum)).backward()
        if last:
            self.reducer.finish()                                 # wait for the in-flight buckets; sums -> means
        # overflow check BEFORE the clip (a clamp would hide an inf), accumulated on the device across the window
        bad = sum((~torch.isfinite(a.flat_g.sum())).float() for a in self.arenas)       # inf/nan anywhere poisons the sum
        self._bad = bad if self._bad is None else self._bad + bad
        if last:
            # Lightning automatic optimization (the reference's default, lightning_auto_optimization=True): the loss is divided by
            # accumulate_grad_batches, and the value clip runs ONCE per optimizer step, on the accumulated, rank-averaged,
            # unscaled gradients, right before optimizer.step() -- clamp((g1 + g2) / 2), not clamp(clamp(g1 / 2) + g2 / 2)
            # (the manual-optimization path ddpm.py:494-497 clips after every backward instead; it is not the mode mirrored here)
            if self.gradient_clip_val:
                lim = self.gradient_clip_val * self.scaler.scale                        # gradients are still loss-scaled
                for a in self.arenas:
                    ops.clamp_f32_(a.flat_g, -lim, lim)
            self.optimizer_step()
        return loss.detach()

    def optimizer_step(self):
        bad, self._bad = self._bad, None
        if self.world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.pg)
        overflow = bool(bad.item() > 0)
        if not overflow:
            for a in self.arenas:
                ops.scale_f32_(a.flat_g, 1.0 / self.scaler.scale)
            for group in self.optimizer.param_groups:
                group["lr"] = self.learning_rate * self.lr_lambda(self.global_step)
            self.optimizer.step()
            self.global_step += 1
        else:
            self.skipped_steps += 1
        self.scaler.update(overflow)
        self.optimizer.zero_grad()
