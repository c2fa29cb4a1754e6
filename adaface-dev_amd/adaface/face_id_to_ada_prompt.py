"""Host-side mirror of the Arc2Face branch of the reference's ``adaface/face_id_to_ada_prompt.py``:
``get_img_prompt_embs`` (:368-470), ``generate_adaface_embeddings`` (:503-578) and
``Arc2Face_ID2AdaPrompt.map_init_id_to_img_prompt_embs`` (:680-724).

The 512-d face-ID vector comes from insightface's ONNX detector/recogniser in the reference (third-party, CPU
round-trips); here it is an input (or, as the reference itself does when no image is given,
``torch.randn(bs, 512)``, :384).  ConsistentID / Joint encoders are out of scope (external package not in the tree)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .arc2face_models import CLIPTextModelWrapper, clip_text_config
from .subj_basis_generator import SubjBasisGenerator, template_ids
from .util import perturb_tensor


class Arc2Face_ID2AdaPrompt(nn.Module):
    name = "arc2face"

    def __init__(self, subj_basis_generator=None, text_to_image_prompt_encoder=None, out_id_embs_cfg_scale=1.0,
                 num_static_img_suffix_embs=0, clip_config=None, tokenizer=None, subject_string="z", adaface_ckpt_path=None,
                 extend_prompt2token_proj_attention_multiplier=1, prompt2token_proj_ext_attention_perturb_ratio=0.1):
        super().__init__()
        self.subject_string = subject_string
        self.output_dim = 768
        self.extend_prompt2token_proj_attention_multiplier = extend_prompt2token_proj_attention_multiplier
        self.prompt2token_proj_ext_attention_perturb_ratio = prompt2token_proj_ext_attention_perturb_ratio
        self.num_id_vecs = self.num_id_vecs0 = 16
        self.id_img_prompt_max_length = 22
        self.num_static_img_suffix_embs = num_static_img_suffix_embs
        self.default_enable_static_img_suffix_embs = False
        self.out_id_embs_cfg_scale = out_id_embs_cfg_scale
        self.gen_neg_img_prompt = False
        self.tokenizer = tokenizer
        cfg = clip_config or clip_text_config()
        self.text_to_image_prompt_encoder = text_to_image_prompt_encoder or CLIPTextModelWrapper(cfg)   # Arc2Face-finetuned encoder
        for p in self.text_to_image_prompt_encoder.parameters():
            p.requires_grad_(False)
        self.subj_basis_generator = subj_basis_generator or SubjBasisGenerator(
            num_id_vecs=self.num_id_vecs, num_static_img_suffix_embs=num_static_img_suffix_embs, clip_config=cfg, tokenizer=tokenizer)

        if adaface_ckpt_path is not None:
            self.load_adaface_ckpt(adaface_ckpt_path)

    def load_adaface_ckpt(self, adaface_ckpt_path):
        """Load ``subj_basis_generator`` from an ``embeddings_gs-N.pt`` checkpoint (reference :109-162): the K/V widths of the
        checkpointed prompt2token_proj are reproduced first, then the state dict is copied, then the configured extra widening."""
        from .ckpt import load_adaface_ckpt_file, module_state_dict
        if isinstance(adaface_ckpt_path, (list, tuple)):
            adaface_ckpt_path = adaface_ckpt_path[0]
        ckpt = load_adaface_ckpt_file(adaface_ckpt_path.split(":")[0])
        gens = ckpt["string_to_subj_basis_generator_dict"]
        if self.subject_string not in gens:
            raise KeyError(f"Subject '{self.subject_string}' not found in {adaface_ckpt_path}")
        g = gens[self.subject_string]
        if isinstance(g, (nn.ModuleList, list, tuple)):
            g = g[{"consistentID": 0, "arc2face": 1}[self.name]]
        sd = dict(module_state_dict(g))
        sbg = self.subj_basis_generator
        n_layers = len(sbg.prompt2token_proj.text_model.encoder.layers)
        mults = getattr(g, "prompt2token_proj_attention_multipliers", None)
        if mults is None:                                   # plain state dict: read the widths off the k_proj shapes
            e = sbg.prompt2token_proj.config.hidden_size
            mults = [sd[f"prompt2token_proj.text_model.encoder.layers.{i}.self_attn.k_proj.weight"].shape[0] // e
                     for i in range(n_layers)]
        rel = [int(m) // int(c) for m, c in zip(mults, sbg.prompt2token_proj_attention_multipliers)]
        sbg.extend_prompt2token_proj_attention(rel, -1, -1, 1, perturb_std=0)
        if sbg.N_SFX == 0:
            sd.pop("static_img_suffix_embs", None)
        elif "static_img_suffix_embs" in sd and sd["static_img_suffix_embs"].shape[1] != sbg.N_SFX:
            n = min(sd["static_img_suffix_embs"].shape[1], sbg.N_SFX)       # reference initialize_static_img_suffix_embs: keep the overlap
            new = sbg.static_img_suffix_embs.detach().clone()
            new[:, :n] = sd["static_img_suffix_embs"][:, :n]
            sd["static_img_suffix_embs"] = new
        ret = sbg.load_state_dict(sd, strict=False)
        if self.extend_prompt2token_proj_attention_multiplier > 1:
            sbg.extend_prompt2token_proj_attention(None, -1, -1, self.extend_prompt2token_proj_attention_multiplier,
                                                   perturb_std=self.prompt2token_proj_ext_attention_perturb_ratio)
        sbg.freeze_prompt2token_proj()
        return ret

    @property
    def dtype(self):
        return torch.float16

    @torch.no_grad()
    def map_init_id_to_img_prompt_embs(self, init_id_embs, clip_features=None, called_for_neg_img_prompt=False):
        """[N,512] unit-norm ID -> [N,16,768]: the ID, zero-padded to 768, replaces the 'id' token embedding of
        "photo of a id person" (22 tokens); CLIP text forward; tokens 4:20."""
        enc = self.text_to_image_prompt_encoder
        n = len(init_id_embs)
        input_ids = template_ids(["photo", "of", "a", "id", "person"], self.id_img_prompt_max_length, init_id_embs.device).repeat(n, 1)
        id_pos = 4
        emb = F.pad(init_id_embs.to(self.dtype), (0, enc.config.hidden_size - init_id_embs.shape[-1]), "constant", 0)
        token_embs = enc(input_ids=input_ids, return_token_embs=True).to(self.dtype)
        token_embs = torch.cat([token_embs[:, :id_pos], emb[:, None], token_embs[:, id_pos + 1:]], dim=1)
        prompt_embeds = enc(input_ids=input_ids, input_token_embs=token_embs)[0].to(self.dtype)
        return prompt_embeds[:, 4:20]

    def get_img_prompt_embs(self, init_id_embs, pre_clip_features=None, image_paths=None, image_objs=None, id_batch_size=1,
                            skip_non_faces=True, avg_at_stage=None, perturb_at_stage=None, perturb_std=0.0, verbose=False):
        if image_paths is not None or image_objs is not None:
            raise NotImplementedError("face detection / ID extraction from images uses insightface ONNX (third-party, absent)")
        dev = next(self.text_to_image_prompt_encoder.parameters()).device
        if init_id_embs is None:
            faceid_embeds = torch.randn(id_batch_size, 512).to(device=dev, dtype=torch.float16)        # reference :384
        else:
            faceid_embeds = init_id_embs
            if faceid_embeds.shape[0] == 1:
                faceid_embeds = faceid_embeds.repeat(id_batch_size, 1)
        if perturb_at_stage == "id_emb" and perturb_std > 0:
            faceid_embeds = perturb_tensor(faceid_embeds, perturb_std, perturb_std_is_relative=True, keep_norm=True)
        faceid_embeds = F.normalize(faceid_embeds, p=2, dim=-1)
        pos_prompt_embs = self.map_init_id_to_img_prompt_embs(faceid_embeds)
        if avg_at_stage == "img_prompt_emb":
            pos_prompt_embs = pos_prompt_embs.mean(dim=0, keepdim=True)
            faceid_embeds = faceid_embeds.mean(dim=0, keepdim=True)
        if perturb_at_stage == "img_prompt_emb" and perturb_std > 0:
            pos_prompt_embs = perturb_tensor(pos_prompt_embs, perturb_std, perturb_std_is_relative=True, keep_norm=True)
        return 0, faceid_embeds, pos_prompt_embs, None

    @staticmethod
    def average_id_embs(face_id_embs):
        """[N, 512] IDs of N images of one subject -> [1, 512]: mean, then L2 normalisation -- what the reference's
        ``extract_init_id_embeds_from_images(calc_avg=True)`` does with the IDs it extracts (face_id_to_ada_prompt.py:347-350)."""
        return F.normalize(face_id_embs.mean(dim=0, keepdim=True), p=2, dim=-1)

    def generate_adaface_embeddings(self, image_paths=None, face_id_embs=None, img_prompt_embs=None, p_dropout=0,
                                    return_zero_embs_for_dropped_encoders=True, avg_at_stage="id_emb", perturb_at_stage=None,
                                    perturb_std=0, enable_static_img_suffix_embs=None):
        if enable_static_img_suffix_embs is None:
            enable_static_img_suffix_embs = self.default_enable_static_img_suffix_embs
        lens = [self.num_id_vecs + enable_static_img_suffix_embs * self.num_static_img_suffix_embs]
        avg = None if (avg_at_stage is None or str(avg_at_stage).lower() == "none") else avg_at_stage
        if img_prompt_embs is None:
            # as the reference (:529-538): with an averaging stage the ID batch size is 1.  IDs handed in directly are NOT averaged at
            # the 'id_emb' stage -- the reference averages there only while it extracts IDs from images (calc_avg, :325-350;
            # ``average_id_embs`` below is that step for callers that bring pre-extracted IDs of several images)
            bs = 1 if avg is not None else (face_id_embs.shape[0] if face_id_embs is not None else 1)
            _, _, img_prompt_embs, _ = self.get_img_prompt_embs(face_id_embs, None, None, None, bs, perturb_at_stage=perturb_at_stage,
                                                                perturb_std=perturb_std, avg_at_stage=avg)
        elif avg is not None:
            img_prompt_embs = img_prompt_embs.mean(dim=0, keepdim=True)
        embs = self.subj_basis_generator(img_prompt_embs, clip_features=None, raw_id_embs=None,
                                         out_id_embs_cfg_scale=self.out_id_embs_cfg_scale, is_face=True,
                                         enable_static_img_suffix_embs=enable_static_img_suffix_embs)
        if avg is not None:
            embs = embs.squeeze(0)
        return embs, img_prompt_embs, lens
