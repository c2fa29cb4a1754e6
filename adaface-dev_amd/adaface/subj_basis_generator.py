"""Host-side mirror of the hot-path slice of the reference's ``adaface/subj_basis_generator.py``:
``ImgPrompt2TextPrompt.inverse_img_prompt_embs`` (:443-562) and ``SubjBasisGenerator.forward`` (:692-770, face branch) --
the inverse CLIP-text projection that maps 16 image-prompt-space ID embeddings to 16 text-token-space embeddings.

Out of scope: the background branch / DINO object branch (``obj_proj_in``, ``prompt_translator``: unused for faces),
``LayerwiseMLPProjWithSkip`` (``use_layerwise_proj`` defaults to False).

Tokenisation: the templates are constant, so their CLIP BPE ids are constants (no tokenizer files exist offline):
"photo of a " + ", " * (N_ID + 2) and "photo of a id person".  A tokenizer can be passed to override them."""
import torch
import torch.nn as nn

from .arc2face_models import CLIPTextModelWrapper, clip_text_config

# openai/clip-vit-large-patch14 BPE vocabulary ids of the words the two templates use
CLIP_BOS, CLIP_EOS = 49406, 49407
CLIP_IDS = {"photo": 1125, "of": 539, "a": 320, ",": 267, "id": 1014, "person": 2533}


_TEMPLATE_IDS = {}


def template_ids(words, max_length, device=None):
    """Token ids [1, max_length] of a template prompt.  Kept per (words, length, device): a training iteration asks for the same few
    templates every time, and making a device tensor from a Python list is a blocking copy."""
    key = (tuple(words), max_length, str(device))
    t = _TEMPLATE_IDS.get(key)
    if t is None:
        ids = [CLIP_BOS] + [CLIP_IDS[w] for w in words] + [CLIP_EOS]
        ids = ids[:max_length] + [CLIP_EOS] * (max_length - len(ids))          # pad token of the CLIP tokenizer is EOS
        t = _TEMPLATE_IDS[key] = torch.tensor([ids], dtype=torch.long, device=device)
    return t.clone()


class ScaleGrad(torch.autograd.Function):
    """Identity forward, gradient x alpha (reference gen_gradient_scaler / adaface/util.py:97)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.alpha, None


class SubjBasisGenerator(nn.Module):
    def __init__(self, dtype=torch.float16, num_id_vecs=16, num_static_img_suffix_embs=0, output_dim=768,
                 learnable_hidden_state_weights_scheme="per-layer", placeholder_is_bg=False, clip_config=None, tokenizer=None):
        super().__init__()
        if placeholder_is_bg:
            raise NotImplementedError("the background translator branch is unused for faces (SURVEY.md section 2 #9)")
        self.dtype = dtype
        self.N_ID = num_id_vecs
        self.N_SFX = num_static_img_suffix_embs
        self.num_out_embs = self.N_ID + self.N_SFX
        self.output_dim = output_dim
        self.max_prompt_length = 77
        self.tokenizer = tokenizer
        self.prompt2token_proj = CLIPTextModelWrapper(clip_config or clip_text_config())
        self.layerwise_proj = nn.Identity()
        self.prompt2token_proj_attention_multipliers = [1] * len(self.prompt2token_proj.text_model.encoder.layers)
        if self.N_SFX > 0:
            self.static_img_suffix_embs = nn.Parameter(torch.randn(1, self.N_SFX, output_dim))
        else:
            self.static_img_suffix_embs = None
        if learnable_hidden_state_weights_scheme == "per-layer":
            self.hidden_state_layer_weights = nn.Parameter(torch.tensor([[1.0], [2.0], [4.0]]))     # :783-786
            self.hidden_state_layer_weights_grad_scale = 5.0
        else:
            self.hidden_state_layer_weights = None
            self.hidden_state_layer_weights_grad_scale = 1.0
        self.register_buffer("pad_embeddings", torch.zeros(77, output_dim), persistent=False)
        self.freeze_prompt2token_proj()

    def freeze_prompt2token_proj(self):
        """Token / position embeddings stay frozen (reference :841-853); the transformer layers train."""
        for p in self.prompt2token_proj.text_model.embeddings.parameters():
            p.requires_grad_(False)

    def extend_prompt2token_proj_attention(self, prompt2token_proj_attention_multipliers=None, begin_layer_idx=-1, end_layer_idx=-1,
                                           multiplier=1, perturb_std=0.1):
        """Widen the K/V projections of prompt2token_proj layers (reference :791-815); multipliers are relative to the current state."""
        n = len(self.prompt2token_proj.text_model.encoder.layers)
        b = 0 if begin_layer_idx == -1 else begin_layer_idx
        e = n - 1 if end_layer_idx == -1 else end_layer_idx
        if prompt2token_proj_attention_multipliers is None:
            if multiplier == 1:
                return 0
            prompt2token_proj_attention_multipliers = [multiplier if b <= i <= e else 1 for i in range(n)]
        done = self.prompt2token_proj.extend_clip_attention_MKV_multiplier(list(prompt2token_proj_attention_multipliers), perturb_std)
        for i in range(b, e + 1):
            self.prompt2token_proj_attention_multipliers[i] *= prompt2token_proj_attention_multipliers[i]
        return done

    def squeeze_prompt2token_proj_attention(self, prompt2token_proj_attention_divisors=None, begin_layer_idx=-1, end_layer_idx=-1,
                                            divisor=1):
        """Reference :817-841."""
        n = len(self.prompt2token_proj.text_model.encoder.layers)
        b = 0 if begin_layer_idx == -1 else begin_layer_idx
        e = n - 1 if end_layer_idx == -1 else end_layer_idx
        if prompt2token_proj_attention_divisors is None:
            if divisor == 1:
                return 0
            prompt2token_proj_attention_divisors = [divisor if b <= i <= e else 1 for i in range(n)]
        done = self.prompt2token_proj.squeeze_clip_attention_MKV_divisor(list(prompt2token_proj_attention_divisors))
        for i in range(b, e + 1):
            self.prompt2token_proj_attention_multipliers[i] //= prompt2token_proj_attention_divisors[i]
        return done

    def _template(self, bs, device):
        if self.tokenizer is not None:
            ids = self.tokenizer(["photo of a " + ", " * (self.N_ID + 2)] * bs, truncation=True, padding="max_length",
                                 max_length=self.max_prompt_length, return_tensors="pt").input_ids.to(device)
            return ids
        return template_ids(["photo", "of", "a"] + [","] * (self.N_ID + 2), self.max_prompt_length, device).repeat(bs, 1)

    def inverse_img_prompt_embs(self, face_prompt_embs, list_extra_words=None, return_emb_types=("core",),
                                hidden_state_layer_weights=None, enable_static_img_suffix_embs=False):
        if list_extra_words is not None:
            raise NotImplementedError("extra words need the CLIP tokenizer (vocabulary files are not available offline)")
        bs = face_prompt_embs.shape[0]
        input_ids = self._template(bs, face_prompt_embs.device)
        orig_dtype = face_prompt_embs.dtype
        ID_END = 4 + self.N_ID
        token_embs = self.prompt2token_proj(input_ids=input_ids, return_token_embs=True).to(face_prompt_embs.dtype)
        token_embs = torch.cat([token_embs[:, :4], face_prompt_embs, token_embs[:, ID_END:]], dim=1)   # slots 4:ID_END (:497)
        if enable_static_img_suffix_embs and self.N_SFX > 0:
            token_embs = torch.cat([token_embs[:, :ID_END], self.static_img_suffix_embs.expand(bs, -1, -1).to(token_embs.dtype),
                                    token_embs[:, ID_END + self.N_SFX:]], dim=1)
        prompt_embeds = self.prompt2token_proj(input_ids=input_ids, input_token_embs=token_embs,
                                               hidden_state_layer_weights=hidden_state_layer_weights)[0].to(orig_dtype)
        core = prompt_embeds[:, 4:ID_END + (self.N_SFX if enable_static_img_suffix_embs else 0)]
        out = []
        for t in return_emb_types:
            if t == "core":
                out.append(core)
            elif t == "full":
                out.append(prompt_embeds)
            else:
                raise NotImplementedError(f"return_emb_type {t!r}")
        return out

    def forward(self, faceid2img_prompt_embs, clip_features=None, raw_id_embs=None, out_id_embs_cfg_scale=1.0, is_face=True,
                enable_static_img_suffix_embs=False):
        if not is_face:
            raise NotImplementedError("the DINO object branch (obj_proj_in) is unused for faces")
        w = self.hidden_state_layer_weights
        if w is not None and self.hidden_state_layer_weights_grad_scale != 1.0 and w.requires_grad:
            w = ScaleGrad.apply(w, self.hidden_state_layer_weights_grad_scale)
        ada_id_embs, = self.inverse_img_prompt_embs(faceid2img_prompt_embs, None, ["core"], w, enable_static_img_suffix_embs)
        out = self.layerwise_proj(ada_id_embs)
        if out_id_embs_cfg_scale != 1:
            pad = self.pad_embeddings[4:4 + self.N_ID].unsqueeze(0).to(out.device, out.dtype)
            out = torch.cat([out[:, :self.N_ID] * out_id_embs_cfg_scale + pad * (1 - out_id_embs_cfg_scale), out[:, self.N_ID:]], dim=1)
        return out
