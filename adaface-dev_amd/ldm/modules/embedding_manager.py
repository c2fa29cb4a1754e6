"""Host-side mirror of the reference's ``ldm/modules/embedding_manager.py`` ``EmbeddingManager``: called
inside the text encoder's embedding step (``FrozenCLIPEmbedder.forward``), it replaces the token embeddings of the subject
placeholder ("z" followed by "," filler tokens) with the K ada embeddings produced on the fly by the ID -> prompt encoder
(``Arc2Face_ID2AdaPrompt.generate_adaface_embeddings``, the F2-F6 stack on the gfx950 kernels), records where they went
(``placeholder2indices``), which prompt tokens are real (``prompt_emb_mask`` / ``prompt_pad_mask``), finds class strings in
class prompts (``cls_delta_string_indices``), owns the trainable parameter groups and reads / writes ``embeddings_gs-N.pt``.

Reference lines: ``__init__`` :36-200, ``init_cls_delta_tokens`` :202-233, ``forward`` :236-252, ``update_text_embeddings`` :254-421,
``update_prompt_masks`` :425-430, ``clear_prompt_adhoc_info`` :433-436, ``set_curr_batch_subject_names`` :440-464,
``update_placeholder_indices`` :466-489, ``set_image_prompts_and_iter_type`` :502-511, ``save`` :514-524, ``load`` :527-662,
``optimized_parameters`` :666-693.

This is index bookkeeping on [B, 77] integer tensors plus one scatter of [B, K, 768] embeddings: torch indexing on the device,
no kernel of its own.  Differences from the reference, all at its ``breakpoint()`` guards: they raise here.  Only the Arc2Face
encoder exists (``adaface_encoder_types`` must be ``None`` / ``["arc2face"]``); an encoder object can be injected."""
from collections import OrderedDict
from functools import partial

import torch
import torch.distributed as dist
from torch import nn

from ..util import (anneal_perturb_embedding, extract_first_index_in_each_instance, get_clip_tokens_for_string,
                    get_embeddings_for_clip_tokens, scan_cls_delta_strings)

FFN_ADAPTER_NAMES = ("recon_loss", "unet_distill", "comp_distill")


class EmbeddingManager(nn.Module):
    def __init__(self, text_embedder, subject_strings, subj_name_to_cls_delta_string=None, out_emb_dim=768, num_unet_ca_layers=16,
                 layer_idx2ca_layer_idx=None, training_perturb_std_range=None, training_perturb_prob=None, cls_delta_string="person",
                 cls_delta_token_weights=None, prompt2token_proj_ext_attention_perturb_ratio=0, adaface_ckpt_paths=None,
                 adaface_encoder_types=None, enabled_encoders=None, extend_prompt2token_proj_attention_multiplier=1,
                 num_static_img_suffix_embs=0, p_encoder_dropout=0, multi_token_filler=",", unet_lora_modules=None,
                 load_unet_attn_lora_from_ckpt=True, unet_ffn_adapters_to_load=("recon_loss", "unet_distill"),
                 id2ada_prompt_encoder=None):
        super().__init__()
        if adaface_encoder_types is not None and list(adaface_encoder_types) != ["arc2face"]:
            raise NotImplementedError("only the Arc2Face ID encoder is in scope (ConsistentID needs an external package)")
        self.rank = -1
        self.string_to_token_dict = OrderedDict()
        self.string_to_subj_basis_generator_dict = nn.ModuleDict()
        self.placeholder_to_emb_cache = nn.ParameterDict()
        self.num_unet_ca_layers = num_unet_ca_layers
        self.subject_strings = list(subject_strings)
        self.subject_string_dict = {s: True for s in self.subject_strings}
        self.placeholder_strings = list(subject_strings)
        self.set_training_perturb_specs(training_perturb_std_range, training_perturb_prob)
        self.layer_idx2ca_layer_idx = layer_idx2ca_layer_idx or {1: 0, 2: 1, 4: 2, 5: 3, 7: 4, 8: 5, 12: 6, 16: 7, 17: 8, 18: 9, 19: 10,
                                                                 20: 11, 21: 12, 22: 13, 23: 14, 24: 15}
        self.ca_layer_idx2layer_idx = {v: k for k, v in self.layer_idx2ca_layer_idx.items()}
        self.ca_infeat_dims = [320, 320, 640, 640, 1280, 1280, 1280, 1280, 1280, 1280, 640, 640, 640, 320, 320, 320]
        self.token2num_vectors = {}
        self.out_emb_dim = out_emb_dim
        self.p_encoder_dropout = p_encoder_dropout
        self.get_tokens_for_string = partial(get_clip_tokens_for_string, text_embedder.tokenizer)
        self.get_embeddings_for_tokens = partial(get_embeddings_for_clip_tokens, text_embedder.transformer.text_model.embeddings)
        self.cls_delta_string = cls_delta_string
        self.prompt2token_proj_ext_attention_perturb_ratio = prompt2token_proj_ext_attention_perturb_ratio
        self.adaface_encoder_types = adaface_encoder_types
        self.enabled_encoders = enabled_encoders
        if id2ada_prompt_encoder is None:
            from ...adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
            id2ada_prompt_encoder = Arc2Face_ID2AdaPrompt(
                num_static_img_suffix_embs=num_static_img_suffix_embs,
                extend_prompt2token_proj_attention_multiplier=extend_prompt2token_proj_attention_multiplier,
                prompt2token_proj_ext_attention_perturb_ratio=prompt2token_proj_ext_attention_perturb_ratio)
        self.id2ada_prompt_encoder = id2ada_prompt_encoder

        if self.cls_delta_string is not None:
            self.cls_delta_tokens = self.get_tokens_for_string(cls_delta_string)
            if cls_delta_token_weights is None:
                w = torch.ones(len(self.cls_delta_tokens))
                w[-1] = 2
            else:
                w = torch.tensor(cls_delta_token_weights, dtype=float)
            w = w ** 2
            self.cls_delta_token_weights = w / w.max()          # the main (last) word gets 1, the words before it 0.25
        else:
            self.cls_delta_tokens = None
            self.cls_delta_token_weights = None

        for placeholder_string in self.placeholder_strings:
            self.string_to_token_dict[placeholder_string] = self.get_tokens_for_string(placeholder_string, force_single_token=True)[0].item()
            self.string_to_subj_basis_generator_dict[placeholder_string] = self.id2ada_prompt_encoder.subj_basis_generator
            self.token2num_vectors[placeholder_string] = self.id2ada_prompt_encoder.num_id_vecs
            if num_static_img_suffix_embs > 0:
                self.token2num_vectors[placeholder_string] += self.id2ada_prompt_encoder.num_static_img_suffix_embs

        self.multi_token_filler = multi_token_filler
        self.string_to_token_dict[multi_token_filler] = self.get_tokens_for_string(multi_token_filler, force_single_token=True)[0].item()
        self.unet_lora_modules = unet_lora_modules
        if adaface_ckpt_paths is not None:
            self.load(adaface_ckpt_paths, load_unet_attn_lora_from_ckpt, unet_ffn_adapters_to_load)
        self.init_cls_delta_tokens(self.get_tokens_for_string, subj_name_to_cls_delta_string, cls_delta_string)
        self.layer_idx = -1
        self.clear_prompt_adhoc_info()
        self.cls_delta_string_indices = []
        self.iter_type = None          # 'recon_iter', 'unet_distill_iter', 'compos_distill_iter', 'plain_text_iter'
        self.set_curr_batch_subject_names(["default"])
        self.set_image_prompts_and_iter_type(None, None, "plain_text_iter", real_batch_size=10000)
        self.loss_call_count = 0
        self.CLS_DELTA_STRING_MAX_SEARCH_SPAN += 1               # "just to be safe" (reference :199)

    def init_cls_delta_tokens(self, get_tokens_for_string, subj_name_to_cls_delta_string, cls_delta_string=None):
        m = dict(subj_name_to_cls_delta_string or {})
        if cls_delta_string is not None:
            m["default"] = cls_delta_string
        m["rand_id_to_img_prompt"] = "person"
        self.subj_name_to_cls_delta_string = m
        self.subj_name_to_cls_delta_tokens = {name: get_tokens_for_string(s) for name, s in m.items()}
        self.CLS_DELTA_STRING_MAX_SEARCH_SPAN = max([0] + [len(t) - 1 for t in self.subj_name_to_cls_delta_tokens.values()])

    # ------------------------------------------------------------------ the hook the text encoder calls
    def forward(self, tokenized_text, embedded_text):
        """tokenized_text [B, N] ids, embedded_text [B, N, 768] token-table embeddings -> patched copy."""
        self.clear_prompt_adhoc_info()
        patched = self.update_text_embeddings(tokenized_text, embedded_text.clone())
        self.update_prompt_masks(tokenized_text)
        return patched

    def update_text_embeddings(self, tokenized_text, embedded_text):
        BS, N = tokenized_text.shape
        if self.rank == -1:
            self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        filler = self.string_to_token_dict[self.multi_token_filler]
        self.cls_delta_string_indices = []
        for placeholder_string, placeholder_token in self.string_to_token_dict.items():
            if placeholder_string == self.multi_token_filler:
                continue
            hits = torch.where(tokenized_text == placeholder_token)
            if hits[0].numel() == 0:
                continue
            first_b, first_n = extract_first_index_in_each_instance(hits)          # later occurrences belong to the background prompt
            occurs = first_b.numel()
            if occurs < BS and self.CLS_DELTA_STRING_MAX_SEARCH_SPAN > 0 and len(self.current_subj_name_to_cls_delta_tokens) > 0:
                self.cls_delta_string_indices += scan_cls_delta_strings(tokenized_text, (first_b, first_n),
                                                                        self.current_subj_name_to_cls_delta_tokens,
                                                                        self.CLS_DELTA_STRING_MAX_SEARCH_SPAN)
            id2img_prompt_embs = self.image_prompt_dict["subj"] if self.curr_subj_is_face else None
            if self.iter_type == "compos_distill_iter":
                id2img_prompt_embs = id2img_prompt_embs[:1]                         # the whole batch is one subject
            adaface_subj_embs, _, _lens = self.id2ada_prompt_encoder.generate_adaface_embeddings(
                image_paths=None, face_id_embs=None, img_prompt_embs=id2img_prompt_embs,
                p_dropout=self.p_encoder_dropout if self.training else 0, return_zero_embs_for_dropped_encoders=False,
                avg_at_stage=None, enable_static_img_suffix_embs=(self.iter_type == "unet_distill_iter"))
            if adaface_subj_embs is None:
                raise RuntimeError("generate_adaface_embeddings returned no embeddings")
            if adaface_subj_embs.shape[0] < occurs:
                if adaface_subj_embs.shape[0] < self.real_batch_size and self.iter_type != "compos_distill_iter":
                    raise ValueError(f"{adaface_subj_embs.shape[0]} subject embeddings for a real batch of {self.real_batch_size}")
                adaface_subj_embs = adaface_subj_embs.repeat(occurs // adaface_subj_embs.shape[0], 1, 1)
            adaface_subj_embs = adaface_subj_embs.to(embedded_text.dtype)
            K = adaface_subj_embs.shape[1]
            if K > self.token2num_vectors[placeholder_string]:
                raise ValueError(f"{K} subject embeddings but only {self.token2num_vectors[placeholder_string]} slots")
            k2 = 0
            for k in range(K):
                emb_k = adaface_subj_embs[:, k]
                if self.training and self.training_perturb_std_range is not None:
                    emb_k = anneal_perturb_embedding(emb_k, 0, self.training_perturb_std_range, None,
                                                     self.training_perturb_prob[self.iter_type], perturb_std_is_relative=True,
                                                     keep_norm=False, verbose=False)
                if emb_k.shape[0] != occurs:
                    raise ValueError(f"{emb_k.shape[0]} embeddings for {occurs} prompts holding the subject token")
                # the k-th slot: the next position (same offset in every instance) holding the placeholder or a filler token
                while True:
                    if (first_n + k2 >= N).any():
                        raise ValueError("ran out of prompt tokens while placing the subject embeddings")
                    tok = tokenized_text[first_b, first_n + k2]
                    if not ((tok != placeholder_token) & (tok != filler)).any():
                        break
                    k2 += 1
                embedded_text[first_b, first_n + k2] = emb_k
                k2 += 1
            self.update_placeholder_indices(tokenized_text, placeholder_string, placeholder_token, K)
        return embedded_text

    def update_prompt_masks(self, tokenized_text):
        """BOS / EOS(=padding) excluded; NOTE the pad mask tests id 49047, as the reference does (:429)."""
        self.prompt_emb_mask = ((tokenized_text != 49406) & (tokenized_text != 49407)).unsqueeze(2)
        self.prompt_pad_mask = (tokenized_text == 49047).unsqueeze(2)

    def clear_prompt_adhoc_info(self):
        self.placeholder2indices = {}
        self.prompt_emb_mask = None
        self.prompt_pad_mask = None

    def set_curr_batch_subject_names(self, subj_names):
        self.curr_batch_subj_names = subj_names
        self.current_subj_name_to_cls_delta_tokens = {n: self.subj_name_to_cls_delta_tokens[n] for n in subj_names}
        if len(subj_names) > 0:
            self.curr_subj_is_face = True
        if len(self.current_subj_name_to_cls_delta_tokens) > 0:
            self.cls_delta_strings = [self.subj_name_to_cls_delta_string[n] for n in subj_names]
        else:
            self.cls_delta_strings = None

    def update_placeholder_indices(self, tokenized_text, placeholder_string, placeholder_token, num_vectors_per_subj_token):
        b, n = torch.where(tokenized_text == placeholder_token)
        if len(b) == 0:
            self.placeholder2indices[placeholder_string] = None
            return
        b, n = extract_first_index_in_each_instance((b, n))
        K = num_vectors_per_subj_token
        if K > 1:                                                                  # [b1_v1 .. b1_vK, b2_v1 ..]: contiguous slots
            b = b.repeat_interleave(K)
            n = n.repeat_interleave(K) + torch.arange(K, device=tokenized_text.device).repeat(len(n))
        self.placeholder2indices[placeholder_string] = (b, n)

    def set_training_perturb_specs(self, training_perturb_std_range, training_perturb_prob):
        self.training_perturb_std_range = training_perturb_std_range
        self.training_perturb_prob = training_perturb_prob

    def set_image_prompts_and_iter_type(self, id2img_prompt_embs, clip_bg_features, iter_type, real_batch_size):
        self.image_prompt_dict = {"subj": id2img_prompt_embs, "bg": clip_bg_features}
        self.iter_type = iter_type
        self.real_batch_size = real_batch_size
        if self.cls_delta_strings is not None and iter_type == "compos_distill_iter":
            self.cls_delta_strings = self.cls_delta_strings[:1]

    # ------------------------------------------------------------------ embeddings_gs-N.pt
    def save(self, adaface_ckpt_path):
        """Same dict layout as the reference, embedding_manager.py:513-524 (the generator modules are pickled whole).  Written
        through ``adaface.ckpt.save_adaface_ckpt_file``: reference class paths, CPU tensors, no derived fp16 weight packs."""
        from ...adaface.ckpt import save_adaface_ckpt_file
        saved = {"string_to_subj_basis_generator_dict": self.string_to_subj_basis_generator_dict,
                 "placeholder_strings": self.placeholder_strings, "subject_strings": self.subject_strings}
        if self.unet_lora_modules is not None:
            saved["unet_lora_modules"] = self.unet_lora_modules.state_dict()
        save_adaface_ckpt_file(saved, adaface_ckpt_path)

    def load(self, adaface_ckpt_paths, load_unet_attn_lora_from_ckpt=True, unet_ffn_adapters_to_load=("recon_loss", "unet_distill")):
        from ...adaface.ckpt import load_adaface_ckpt_file
        if isinstance(adaface_ckpt_paths, str):
            adaface_ckpt_paths = [adaface_ckpt_paths]
        self.string_to_token_dict = OrderedDict()
        self.subject_strings = []
        adaface_ckpt_path = adaface_ckpt_paths[0]
        parts = adaface_ckpt_path.split(":")
        ckpt = load_adaface_ckpt_file(parts[0])
        lora_path = adaface_ckpt_paths[1] if len(adaface_ckpt_paths) == 2 else adaface_ckpt_path
        lora_ckpt = load_adaface_ckpt_file(lora_path.split(":")[0]) if len(adaface_ckpt_paths) == 2 else ckpt
        self.id2ada_prompt_encoder.load_adaface_ckpt(adaface_ckpt_path)
        mapper = dict(m.split("-") for m in parts[1].split(",")) if len(parts) == 2 else None      # "path:from-to,from2-to2"
        for km in ckpt.get("placeholder_strings", []):
            km2 = mapper[km] if mapper is not None and km in mapper else km
            token = self.get_tokens_for_string(km2, force_single_token=True)[0]
            if km2 in self.string_to_token_dict:
                continue
            if km2 not in self.subject_strings:
                self.subject_strings.append(km2)
            self.string_to_token_dict[km2] = token.item()

        if self.unet_lora_modules is not None and "unet_lora_modules" in lora_ckpt \
                and (load_unet_attn_lora_from_ckpt or len(unet_ffn_adapters_to_load) > 0):
            sd = dict(lora_ckpt["unet_lora_modules"])
            if not load_unet_attn_lora_from_ckpt:
                sd = {k: v for k, v in sd.items() if "attn2_processor" not in k}
            attn_sd = {k: v for k, v in sd.items() if "resnets" not in k}
            ffn_sd = {k: v for k, v in sd.items() if "resnets" in k}
            if "all" not in unet_ffn_adapters_to_load:
                ffn_sd = {k: v for k, v in ffn_sd.items() if any(a in k for a in unet_ffn_adapters_to_load)}
            sd = {**attn_sd, **ffn_sd}
            for key in list(sd.keys()):
                if key.startswith("base_model_model_"):
                    sd[key.replace("base_model_model_", "")] = sd.pop(key)
            own = self.unet_lora_modules.state_dict().keys()
            for key in list(sd.keys()):                                            # old checkpoints: one 'default' adapter
                renamed = False
                for adapter in FFN_ADAPTER_NAMES:
                    akey = key.replace("default.weight", f"{adapter}.weight")
                    if akey in own and akey not in sd:
                        sd[akey] = sd[key]
                        renamed = True
                if renamed:
                    del sd[key]
            self.unet_lora_load_result = self.unet_lora_modules.load_state_dict(sd, strict=False)

        self.string_to_token_dict[self.multi_token_filler] = self.get_tokens_for_string(self.multi_token_filler, force_single_token=True)[0].item()
        self.placeholder_strings = self.subject_strings
        self.subject_string_dict = {s: True for s in self.subject_strings}
        for s in self.placeholder_strings:
            self.string_to_subj_basis_generator_dict[s] = self.id2ada_prompt_encoder.subj_basis_generator
            self.token2num_vectors.setdefault(s, self.id2ada_prompt_encoder.num_id_vecs)

    def optimized_parameters(self, lr, weight_decay, lora_lr, lora_weight_decay):
        sbg = [p for p in self.string_to_subj_basis_generator_dict.parameters() if p.requires_grad]
        loras = list(self.unet_lora_modules.parameters()) if self.unet_lora_modules is not None else []
        for p in loras:
            p.requires_grad = True
        return [{"params": sbg, "lr": lr, "weight_decay": weight_decay},
                {"params": loras, "lr": lora_lr, "weight_decay": lora_weight_decay}]
