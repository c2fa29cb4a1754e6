"""Host-side mirror of the reference's ``adaface/arc2face_models.py``: ``CLIPTextModelWrapper`` (CLIP ViT-L/14 text
transformer that accepts pre-computed token embeddings and mixes the last-k hidden states, :233-338) and
``CLIPAttentionMKV`` (K/V projections widened x m, :51-231), executed with this package's gfx950 kernels.

The reference subclasses ``transformers.CLIPTextModel`` and depends on transformers-4 internals; here the module tree
is built directly with the transformers-4 parameter names (``text_model.embeddings.token_embedding.weight``,
``text_model.encoder.layers.N.self_attn.{q,k,v,out}_proj``, ``layer_norm1/2``, ``mlp.fc1/fc2``,
``text_model.final_layer_norm``) so Arc2Face / openai-CLIP checkpoints load with ``load_state_dict``.

Execution: token-major fp16 ``[B*T, 768]``; LayerNorm kernel; q / k / v projections and the MLP through ``af_gemm``
(bias, quick-GELU and the residual adds fused in the epilogues); causal attention -- including the m-keys-per-token
layout of CLIPAttentionMKV, whose ``[B, T, m*768]`` K/V projection output *is* a ``[B, T*m, 768]`` key matrix -- in
the fused attention kernel (``causal_m``)."""
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import autograd_ops as ag
from .. import ops
from ..ldm.modules.diffusionmodules.util import LayerNorm, Linear
from ..ops import AF_ACT_QUICKGELU, F16
from .util import perturb_tensor


def clip_text_config(hidden_size=768, num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072, vocab_size=49408,
                     max_position_embeddings=77, attention_dropout=0.0, eos_token_id=2):
    """openai/clip-vit-large-patch14 text config by default."""
    return SimpleNamespace(hidden_size=hidden_size, num_attention_heads=num_attention_heads, num_hidden_layers=num_hidden_layers,
                           intermediate_size=intermediate_size, vocab_size=vocab_size, max_position_embeddings=max_position_embeddings,
                           attention_dropout=attention_dropout, eos_token_id=eos_token_id, hidden_act="quick_gelu")


class CLIPAttentionMKV(nn.Module):
    def __init__(self, config, multiplier=1):
        super().__init__()
        self.config = config
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        if self.head_dim * self.num_heads != self.embed_dim:
            raise ValueError(f"embed_dim must be divisible by num_heads (got {self.embed_dim} and {self.num_heads}).")
        self.scale = self.head_dim ** -0.5
        self.dropout = config.attention_dropout
        self.multiplier = multiplier
        self.k_proj = Linear(self.embed_dim, self.embed_dim * multiplier)
        self.v_proj = Linear(self.embed_dim, self.embed_dim * multiplier)
        self.q_proj = Linear(self.embed_dim, self.embed_dim)
        self.out_proj = Linear(self.embed_dim, self.embed_dim)

    def extend_weights(self, clip_attn_layer, layer_idx, multiplier, perturb_std=0.2, perturb_std_is_relative=True,
                       perturb_keep_norm=False, verbose=False):
        """Widen K/V x `multiplier` by repeating the weights and perturbing the extra copies (reference :82-127)."""
        e0 = clip_attn_layer.k_proj.weight.shape[0]
        self.multiplier *= multiplier
        with torch.no_grad():
            for n in ("q_proj", "out_proj"):
                getattr(self, n).weight.data = getattr(clip_attn_layer, n).weight.data.clone()
                getattr(self, n).bias.data = getattr(clip_attn_layer, n).bias.data.clone()
            for n in ("k_proj", "v_proj"):
                src, dst = getattr(clip_attn_layer, n), getattr(self, n)
                dst.bias.data = src.bias.data.repeat(multiplier)
                dst.weight.data = src.weight.data.repeat(multiplier, 1)
                dst.out_features = dst.weight.shape[0]
                if perturb_std > 0:
                    dst.weight.data[e0:] = perturb_tensor(dst.weight.data[e0:], perturb_std, perturb_std_is_relative, perturb_keep_norm)

    def squeeze_weights(self, clip_attn_layer, divisor):
        assert self.multiplier % divisor == 0
        self.multiplier //= divisor
        with torch.no_grad():
            for n in ("k_proj", "v_proj"):
                src, dst = getattr(clip_attn_layer, n), getattr(self, n)
                dst.bias.data = src.bias.data.reshape(divisor, -1).mean(dim=0)
                dst.weight.data = src.weight.data.reshape(divisor, -1, src.weight.shape[1]).mean(dim=0)
                dst.out_features = dst.weight.shape[0]

    # ---- multiplier-1 layers run q | k | v as ONE GEMM on the concatenated weights (packs rebuilt when a parameter changes)
    def _qkv_params(self):
        return (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias)

    def qkv_packed(self):
        from ..ldm.modules.diffusionmodules.util import _PackCache
        if not hasattr(self, "_cache"):
            self._cache = _PackCache()
        ps = self._qkv_params()
        return self._cache.get(ps, lambda: ops.pack_matrix(torch.cat([p.detach() for p in ps[:3]], 0), torch.cat([p.detach() for p in ps[3:]], 0),
                                                           ps[0].device))

    def qkv_packed_bwd(self):
        from ..ldm.modules.diffusionmodules.util import _PackCache
        if not hasattr(self, "_cache_bwd"):
            self._cache_bwd = _PackCache()
        ps = self._qkv_params()[:3]
        return self._cache_bwd.get(ps, lambda: ops.pack_matrix(torch.cat([p.detach() for p in ps], 0).t().contiguous(), None, ps[0].device))

    def hip(self, x2d, B, T, residual=None, causal=True):
        """x2d [B*T, E] fp16 -> [B*T, E] (+ residual)."""
        E, m = self.embed_dim, self.multiplier
        if m == 1:
            qkv = ops.gemm(x2d, self.qkv_packed())
            vt = ops.transpose_tokens(qkv[:, 2 * E:], B, T, E, 3 * E)
            o = ops.attention(qkv[:, :E], qkv[:, E:2 * E], vt, B=B, Nq=T, L=T, heads=self.num_heads, d=self.head_dim, ldq=3 * E, ldk=3 * E,
                              scale=self.scale, causal_m=1 if causal else 0)
            return self.out_proj.hip(o, residual=residual)
        q = self.q_proj.hip(x2d)
        k = self.k_proj.hip(x2d).reshape(B * T * m, E)          # [B, T, m*E] rows ARE [B, T*m, E] key rows
        v = self.v_proj.hip(x2d).reshape(B * T * m, E)
        vt = ops.transpose_tokens(v, B, T * m, E, E)
        o = ops.attention(q, k, vt, B=B, Nq=T, L=T * m, heads=self.num_heads, d=self.head_dim, ldq=E, ldk=E, scale=self.scale,
                          causal_m=m if causal else 0)
        return self.out_proj.hip(o, residual=residual)

    def hip_autograd(self, x2d, B, T, residual=None, causal=True):
        """Same as `hip`, built from autograd nodes so the projection weights receive gradients."""
        E, m = self.embed_dim, self.multiplier
        if m == 1:
            qkv = ag.QKVLinearFn.apply(x2d, *self._qkv_params(), self)
            o = ag.AttentionQKVFn.apply(qkv, B, T, self.num_heads, self.scale, causal)
            return ag.linear(self.out_proj, o, residual=residual)
        q = ag.linear(self.q_proj, x2d)
        k = ag.linear(self.k_proj, x2d).reshape(B * T * m, E)
        v = ag.linear(self.v_proj, x2d).reshape(B * T * m, E)
        o = ag.AttentionFn.apply(q, k, v, B, T, m, self.num_heads, self.scale, causal)
        return ag.linear(self.out_proj, o, residual=residual)

    def forward(self, hidden_states, attention_mask=None, causal_attention_mask=None, output_attentions=False):
        if attention_mask is not None or output_attentions:
            raise NotImplementedError("padding masks / attention outputs are not used on the AdaFace path")
        B, T, E = hidden_states.shape
        y = self.hip(hidden_states.reshape(B * T, E).to(F16).contiguous(), B, T, causal=causal_attention_mask is not None)
        y = y.reshape(B, T, E)
        return (y if hidden_states.dtype == F16 else y.to(hidden_states.dtype)), None


class CLIPMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = Linear(config.intermediate_size, config.hidden_size)

    def hip(self, x2d, residual=None):
        return self.fc2.hip(self.fc1.hip(x2d, act=AF_ACT_QUICKGELU), residual=residual)

    def hip_autograd(self, x2d, residual=None):
        return ag.linear(self.fc2, ag.QuickGeluFn.apply(ag.linear(self.fc1, x2d)), residual=residual)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = CLIPAttentionMKV(config, multiplier=1)
        self.layer_norm1 = LayerNorm(config.hidden_size, eps=1e-5)
        self.mlp = CLIPMLP(config)
        self.layer_norm2 = LayerNorm(config.hidden_size, eps=1e-5)

    def hip(self, h, B, T):
        h = self.self_attn.hip(self.layer_norm1.hip(h), B, T, residual=h)
        return self.mlp.hip(self.layer_norm2.hip(h), residual=h)

    def hip_autograd(self, h, B, T):
        h = self.self_attn.hip_autograd(ag.layer_norm(self.layer_norm1, h), B, T, residual=h)
        return self.mlp.hip_autograd(ag.layer_norm(self.layer_norm2, h), residual=h)


class CLIPTextEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.token_embedding = nn.Embedding(config.vocab_size, config.hidden_size)
        self.position_embedding = nn.Embedding(config.max_position_embeddings, config.hidden_size)

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None):
        T = input_ids.shape[-1] if input_ids is not None else inputs_embeds.shape[-2]
        if inputs_embeds is None:
            inputs_embeds = self.token_embedding(input_ids)
        pos = self.position_embedding.weight[:T] if position_ids is None else self.position_embedding(position_ids)
        return inputs_embeds + pos


class CLIPEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(config) for _ in range(config.num_hidden_layers)])


class CLIPTextTransformer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = CLIPTextEmbeddings(config)
        self.encoder = CLIPEncoder(config)
        self.final_layer_norm = LayerNorm(config.hidden_size, eps=1e-5)
        self.eos_token_id = config.eos_token_id


class CLIPTextModelWrapper(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.config = config or clip_text_config()
        self.text_model = CLIPTextTransformer(self.config)

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, input_token_embs=None, hidden_state_layer_weights=None, return_token_embs=False):
        """Reference contract (arc2face_models.py:236-338).  Returns (last_hidden_state, pooled_output[, hidden_states])
        as a tuple (return_dict is not supported: the AdaFace callers index [0])."""
        tm = self.text_model
        if return_token_embs:
            return tm.embeddings.token_embedding(input_ids)
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed on the AdaFace path")
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        B, T = input_ids.shape
        E = self.config.hidden_size
        want_hidden = bool(output_hidden_states) or hidden_state_layer_weights is not None
        h0 = tm.embeddings(input_ids=input_ids, position_ids=position_ids, inputs_embeds=input_token_embs)
        out_dtype = h0.dtype
        h = h0.reshape(B * T, E).to(F16).contiguous()
        hs = [h]
        # training: any encoder weight (or the input embeddings) wants a gradient -> autograd-node execution
        train = torch.is_grad_enabled() and (h0.requires_grad or any(p.requires_grad for p in tm.encoder.parameters())
                                             or tm.final_layer_norm.weight.requires_grad)
        for layer in tm.encoder.layers:
            h = layer.hip_autograd(h, B, T) if train else layer.hip(h, B, T)
            if want_hidden:
                hs.append(h)
        if hidden_state_layer_weights is None:
            last = h
        else:
            k = len(hidden_state_layer_weights)
            w = hidden_state_layer_weights.to(torch.float32)
            w = (w / w.sum(dim=0, keepdim=True)).unsqueeze(1)                       # [k, 1, 1 | E]  (reference :296-300)
            last = (torch.stack([t.float() for t in hs[-k:]], dim=0) * w).sum(dim=0).to(F16).contiguous()
        last = (ag.layer_norm(tm.final_layer_norm, last) if train else tm.final_layer_norm.hip(last)).reshape(B, T, E)
        last = last if out_dtype == F16 else last.to(out_dtype)
        if tm.eos_token_id == 2:
            pos = input_ids.to(torch.int).argmax(dim=-1)
        else:
            pos = (input_ids.to(torch.int) == tm.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=last.device), pos]
        hidden = tuple(t.reshape(B, T, E) for t in hs) if want_hidden else None
        return (last, pooled) + ((hidden,) if hidden is not None else ())

    @torch.no_grad()
    def extend_position_embeddings(self, max_length):
        """[77, E] -> [max_length, E] by appending copies of the last (max_length - 77) rows, as the reference's hooked text encoder
        does for `--clip_prompt_max_length 97` (ldm/modules/encoders/modules.py:373-382, main.py:272)."""
        emb = self.text_model.embeddings.position_embedding
        old = emb.num_embeddings
        if max_length <= old:
            return
        el = max_length - old
        new = nn.Embedding(max_length, emb.embedding_dim).to(device=emb.weight.device, dtype=emb.weight.dtype)
        new.weight[:old] = emb.weight
        new.weight[old:] = emb.weight[-el:]
        new.weight.requires_grad_(emb.weight.requires_grad)
        self.text_model.embeddings.position_embedding = new
        self.config.max_position_embeddings = max_length

    def extend_clip_attention_MKV_multiplier(self, prompt2token_proj_attention_multipliers=None, perturb_std=0.1,
                                             perturb_std_is_relative=True, perturb_keep_norm=False, verbose=False,
                                             begin_layer_idx=None, end_layer_idx=None, multiplier=None):
        """Widen K/V of every encoder layer whose entry in the per-layer list is not 1 (reference arc2face_models.py:343-360).
        ``begin_layer_idx / end_layer_idx / multiplier`` build that list for an inclusive layer range."""
        layers = self.text_model.encoder.layers
        n = len(layers)
        mults = prompt2token_proj_attention_multipliers
        if mults is None:
            b = 0 if begin_layer_idx in (None, -1) else begin_layer_idx % n
            e = n - 1 if end_layer_idx in (None, -1) else end_layer_idx % n
            mults = [multiplier if b <= i <= e else 1 for i in range(n)]
        extended = 0
        for i, m in enumerate(mults):
            if m == 1:
                continue
            old = layers[i].self_attn
            new = CLIPAttentionMKV(self.config, multiplier=old.multiplier).to(device=old.q_proj.weight.device)
            new.extend_weights(old, i, m, perturb_std, perturb_std_is_relative, perturb_keep_norm, verbose)
            layers[i].self_attn = new
            extended += 1
        return extended

    def squeeze_clip_attention_MKV_divisor(self, prompt2token_proj_attention_divisors):
        """Inverse of the extension: average groups of K/V copies (reference arc2face_models.py:365-382)."""
        layers = self.text_model.encoder.layers
        squeezed = 0
        for i, dv in enumerate(prompt2token_proj_attention_divisors):
            if dv == 1:
                continue
            old = layers[i].self_attn
            new = CLIPAttentionMKV(self.config, multiplier=old.multiplier).to(device=old.q_proj.weight.device)
            with torch.no_grad():
                for nm in ("q_proj", "out_proj"):
                    getattr(new, nm).weight.data = getattr(old, nm).weight.data.clone()
                    getattr(new, nm).bias.data = getattr(old, nm).bias.data.clone()
            new.squeeze_weights(old, dv)
            layers[i].self_attn = new
            squeezed += 1
        return squeezed
