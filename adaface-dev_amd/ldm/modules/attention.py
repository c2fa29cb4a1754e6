"""Host-side mirror of the reference's ``ldm/modules/attention.py`` (CrossAttention,
BasicTransformerBlock, SpatialTransformer, FeedForward / GEGLU, Normalize) on MI355X kernels.

Constructor arguments, attribute names (``save_cross_attn_vars``, ``cached_activations``,
``infeat_size``), parameter names and ``forward`` signatures follow the reference
(attention.py:31-58, 146-304).  What differs is the execution plan:

* activations stay token-major fp16 ``[B*N, C]`` from ``proj_in`` to ``proj_out`` -- the NHWC
  feature map *is* the token matrix, so the two ``rearrange(...).contiguous()`` copies of
  attention.py:293,301 disappear;
* self-attention uses ONE projection GEMM for q, k and v (weights concatenated at pack time);
  cross-attention one for q and one for k|v; the v columns are written transposed by the GEMM
  epilogue, which is the layout the fused attention kernel consumes;
* softmax(QK^T)V is one flash-style kernel (no [b*h, N, L] score tensor, attention.py:181-202);
* the residual adds of BasicTransformerBlock (attention.py:244-250) and SpatialTransformer
  (attention.py:304) are fused into the epilogues of to_out / ff.net.2 / proj_out, and
  GEGLU's ``x * gelu(gate)`` into the epilogue of its projection.
"""
import math
import os as _os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...ops import AF_ACT_GEGLU, F16
from .diffusionmodules.util import (Conv2d, GroupNorm32, LayerNorm, Linear, _PackCache, checkpoint, from_nhwc_f16,
                                    to_nhwc_f16, zero_module)


# LayerNorm -> Linear pairs of the inference pass run as ONE GEMM (ops.pack_matrix_ln); AF_FOLD_LAYERNORM=0 keeps the separate kernels (A/B runs)
FOLD_LAYERNORM = _os.environ.get("AF_FOLD_LAYERNORM", "1") != "0"
FUSE_GN_PROJ = _os.environ.get("AF_FUSE_GN_PROJ", "1") != "0"      # SpatialTransformer: GroupNorm's normalising pass + proj_in in one launch at C = 320 (af_gn_proj_fused)
CHAIN_XATTN = _os.environ.get("AF_CHAIN_XATTN", "1") != "0"    # round 6: the self-attention's to_out + residual as phase 0 of the one-launch C = 320 cross-attention block (af_xattn_chain)
XATTN_FUSE_MIN_TOKENS = int(_os.environ.get("AF_XATTN_FUSE_MIN_TOKENS", "8192"))   # U-Net batch x tokens from which the one-launch block is used
CHAIN_PROJ_OUT = _os.environ.get("AF_CHAIN_PROJ_OUT", "1") != "0"   # the SpatialTransformer's proj_out + residual as the tail of the one-launch feed-forward at C = 320 (af_ff_chain, round 6)
FUSE_XATTN640 = _os.environ.get("AF_FUSE_XATTN640", "0") != "0"  # the C = 640 blocks as one launch too (af_xattn640t_kernel, round 6): measured, see DESIGN.md 8.0
XATTN640_FUSE_MIN_TOKENS = 4096                                   # 64-token workgroups: 64 of them at least
FUSE_XATTN = _os.environ.get("AF_FUSE_XATTN", "1") != "0"      # the C = 320 cross-attention block as ONE launch (af_xattn_fused): on par with the
                                                                # three launches alone, -0.03 ms per denoise step (csrc/af_xattn_fused.hip); 0 = three launches
FUSE_FF = _os.environ.get("AF_FUSE_FF", "1") != "0"          # the C = 320 feed-forward as one launch (af_ff_fused); 0 = the two GEMMs (A/B runs)


def exists(val):
    return val is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def Normalize(in_channels):
    """GroupNorm(32, eps=1e-6, affine) -- reference attention.py:70-71."""
    return GroupNorm32(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class GEGLU(nn.Module):
    """proj: Linear(dim_in, 2*dim_out); forward = value * gelu(gate) (attention.py:31-38)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)
        self._cache = _PackCache()

    def packed(self):
        def build():
            wi, bi = ops.interleave_geglu(self.proj.weight.detach(), self.proj.bias.detach())
            return ops.pack_matrix(wi, bi, self.proj.weight.device)

        return self._cache.get((self.proj.weight, self.proj.bias), build)

    def packed_ln(self, ln):
        """The projection with the LayerNorm in front of it folded in (ops.pack_matrix_ln): takes the un-normalised rows."""
        def build():
            w = self.proj.weight.detach().float()
            b = self.proj.bias.detach().float() + (w * ln.bias.detach().float()[None, :]).sum(dim=1)
            wi, bi = ops.interleave_geglu(w * ln.weight.detach().float()[None, :], b)
            pw = ops.pack_matrix(wi, bi, self.proj.weight.device)
            pw.ln_cs, pw.ln_eps = pw.wt.float().sum(dim=1).contiguous(), float(ln.eps)
            return pw

        if not hasattr(self, "_cache_ln"):
            self._cache_ln = _PackCache()
        return self._cache_ln.get((self.proj.weight, self.proj.bias, ln.weight, ln.bias), build)

    def hip(self, x2d, ln=None):
        return ops.gemm(x2d, self.packed() if ln is None else self.packed_ln(ln), act=AF_ACT_GEGLU)

    def hip_train(self, x2d, ln=None):
        """Un-fused: keeps the (interleaved) pre-activation for the backward.  -> (out, hp).  ``ln``: as in hip()."""
        hp = ops.gemm(x2d, self.packed() if ln is None else self.packed_ln(ln))
        return ops.geglu_fwd(hp), hp

    def packed_bwd(self):
        def build():
            wi, _ = ops.interleave_geglu(self.proj.weight.detach(), self.proj.bias.detach())
            return ops.pack_matrix(wi.t().contiguous(), None, self.proj.weight.device)

        if not hasattr(self, "_cache_bwd"):
            self._cache_bwd = _PackCache()
        return self._cache_bwd.get((self.proj.weight,), build)

    def hip_bwd(self, hp, dout):
        return ops.gemm(ops.geglu_bwd(hp, dout), self.packed_bwd())

    def forward(self, x):
        y = self.hip(x.reshape(-1, x.shape[-1]).to(F16).contiguous())
        y = y.reshape(*x.shape[:-1], y.shape[-1])
        return y if x.dtype == F16 else y.to(x.dtype)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        if not glu:
            raise NotImplementedError("FeedForward(glu=False) is not on the SD-1.5 path")
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is not supported (SD-1.5 uses 0)")
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), Linear(inner_dim, dim_out))

    def ff_fusable(self, M, C, ln) -> bool:
        """Whether this feed-forward runs as the one-launch C = 320 kernel (af_ff_fused / af_ff_chain)."""
        return bool(FUSE_FF and ln is not None and C == 320 and M >= 24576)

    def hip(self, x2d, residual=None, ln=None, post=None):
        """post = (proj_out pack, x_in 2-D, rows per image, gn_cpg): the SpatialTransformer's proj_out + residual chained behind (af_ff_chain; the caller checked ff_fusable)."""
        if post is not None:
            pw_p, x_in, rpb, cpg = post
            return ops.ff_chain(x2d, self.net[0].packed_ln(ln), self.net[2].packed(), residual, pw_p, x_in, rows_per_batch=rpb, gn_cpg=cpg)
        if FUSE_FF and ln is not None and x2d.shape[1] == 320 and x2d.shape[0] >= 24576:
            # the 64 x 64 level: LayerNorm, GEGLU projection, output projection and residual as ONE launch (af_ff_fused): the
            # [tokens, 1280] intermediate never leaves the compute unit.  One workgroup per 128 tokens: worth it once they fill the chip
            # (U-Net batch >= 6: 121.9 vs 166.5 us at batch 8, 105.1 vs 98.4 at batch 4 -- profiles/r03p_ff_fused.txt)
            return ops.ff_fused(x2d, self.net[0].packed_ln(ln), self.net[2].packed(), residual=residual)
        return self.net[2].hip(self.net[0].hip(x2d, ln=ln), residual=residual)

    def hip_train(self, x2d, residual=None, ln=None):
        g, hp = self.net[0].hip_train(x2d, ln=ln)
        return self.net[2].hip(g, residual=residual), hp

    def hip_bwd(self, hp, dy):
        return self.net[0].hip_bwd(hp, self.net[2].hip_dgrad(dy))

    def forward(self, x):
        y = self.hip(x.reshape(-1, x.shape[-1]).to(F16).contiguous()).reshape(x.shape[:-1] + (-1,))
        return y if x.dtype == F16 else y.to(x.dtype)


class CrossAttention(nn.Module):
    """All SD-1.5 attention layers: 8 heads, head dim C/8 (attention.py:146-222)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is not supported (SD-1.5 uses 0)")
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.inner_dim = inner_dim
        self.to_q = Linear(query_dim, inner_dim, bias=False)
        self.to_k = Linear(context_dim, inner_dim, bias=False)
        self.to_v = Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self.save_cross_attn_vars = False
        self.cached_activations = None
        self._kv_pre = None
        self._qkv_cache = _PackCache()
        self._kv_cache = _PackCache()

    # fused projection weights (built once per device / weight version)
    def _packed_qkv(self):
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        return self._qkv_cache.get(ws, lambda: ops.pack_matrix(torch.cat([w.detach() for w in ws], 0), None, ws[0].device))

    def _packed_kv(self):
        ws = (self.to_k.weight, self.to_v.weight)
        return self._kv_cache.get(ws, lambda: ops.pack_matrix(torch.cat([w.detach() for w in ws], 0), None, ws[0].device))

    def _packed_qkv_ln(self, ln):
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        if not hasattr(self, "_qkv_cache_ln"):
            self._qkv_cache_ln = _PackCache()
        return self._qkv_cache_ln.get(ws + (ln.weight, ln.bias), lambda: ops.pack_matrix_ln(
            torch.cat([w.detach() for w in ws], 0), None, ln.weight, ln.bias, ln.eps, ws[0].device))

    def _packed_q_ln(self, ln):
        if not hasattr(self, "_q_cache_ln"):
            self._q_cache_ln = _PackCache()
        return self._q_cache_ln.get((self.to_q.weight, ln.weight, ln.bias), lambda: ops.pack_matrix_ln(
            self.to_q.weight, None, ln.weight, ln.bias, ln.eps, self.to_q.weight.device))

    def xattn_fusable(self, C, B, N, L, ln, chain=False) -> bool:
        """Whether this (cross-attention) layer runs as the one-launch block (af_xattn_fused; ``chain``: af_xattn_chain, C = 320 only)."""
        if not (FUSE_XATTN and ln is not None and self.heads == 8 and not self.save_cross_attn_vars and L <= 80 and self.inner_dim == C):
            return False
        if C == 320:
            return N % 128 == 0 and B * N >= XATTN_FUSE_MIN_TOKENS
        return bool(FUSE_XATTN640 and not chain and C == 640 and N % 64 == 0 and B * N >= XATTN640_FUSE_MIN_TOKENS)

    def hip(self, x2d, B, N, context=None, keybias=None, residual=None, ln=None, defer_out=False, pre=None):
        """x2d [B*N, C] fp16; context [B, L, Cc] fp16 or None (self-attention); keybias fp32
        [B, roundup(L,64)] or None; residual [B*N, C] added after to_out.  ``ln``: the LayerNorm whose output this layer's query
        (and, for self-attention, key / value) projection consumes -- x2d is then the UN-normalised input and the normalisation is
        folded into the projection GEMM.  Returns [B*N, C]."""
        Ci, h, d = self.inner_dim, self.heads, self.dim_head
        if context is None:
            if self.to_k.in_features != self.to_q.in_features:
                raise RuntimeError("CrossAttention: context is required (context_dim != query_dim)")
            L = N
            qk, vt = ops.gemm(x2d, self._packed_qkv() if ln is None else self._packed_qkv_ln(ln), rows_per_batch=N, split_col=2 * Ci)
            q, k, ldq, ldk = qk, qk[:, Ci:], 2 * Ci, 2 * Ci
        else:
            L = context.shape[1]
            ldk = Ci
            if self._kv_pre is not None:                     # projected for all layers at once by UNetModel._project_context_all
                k, vt, ldk = self._kv_pre
            else:
                k, vt = ops.gemm(context.reshape(B * L, context.shape[-1]), self._packed_kv(), rows_per_batch=L, split_col=Ci)
            if pre is not None:
                # round 6: the self-attention's output projection + residual chained in front (BasicTransformerBlock.hip decided with xattn_fusable):
                # pre = (self-attention core output, attn1.to_out pack, the block's input)
                ao, pw_o1, x0 = pre
                return ops.xattn_chain(ao, pw_o1, x0, self._packed_q_ln(ln), k, vt, self.to_out[0].packed(), B=B, N=N, L=L, heads=h, scale=self.scale, ldk=ldk)
            if keybias is None and self.xattn_fusable(x2d.shape[1], B, N, L, ln):
                # the 64 x 64 level: q projection (norm2 folded in), the 77-key core, to_out and the residual as ONE launch over 128-token tiles
                # that stay in LDS (af_xattn_fused, tiled form); worth it once the 128-token workgroups cover enough of the chip
                return ops.xattn_fused(x2d, self._packed_q_ln(ln), k, vt, self.to_out[0].packed(), B=B, N=N, L=L, heads=h, scale=self.scale,
                                       ldk=ldk, residual=residual)
            q = self.to_q.hip(x2d) if ln is None else ops.gemm(x2d, self._packed_q_ln(ln))
            ldq = Ci
        o = ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=h, d=d, ldq=ldq, ldk=ldk, keybias=keybias, scale=self.scale)
        if self.save_cross_attn_vars:
            # attention.py:207-220 -- explicit score / prob only on the (rare) capture path
            out_plain = self.to_out[0].hip(o)
            qc = q if context is not None else qk[:, :Ci].contiguous()
            kc = k.contiguous() if context is not None else qk[:, Ci:].contiguous()
            score, prob = ops.attention_scores(qc, kc, B=B, Nq=N, L=L, heads=h, d=d, scale=self.scale)
            rs = math.sqrt(self.scale)
            qcap = (qc.reshape(B, N, Ci).permute(0, 2, 1).float() * rs).contiguous()
            # v^T comes out of the projection already as [B, C, ld] (ld = L rounded up): the captured layout 'b (h d) n'
            self.cached_activations = {
                "q": qcap,
                "q2": qcap,                                   # query2 = query without a q LoRA (diffusers_attn_lora_capture.py:250, 347-354)
                "k": (kc.reshape(B, L, Ci).permute(0, 2, 1).float() * rs).contiguous(),
                "v": (vt[:, :Ci, :L].float() * rs).contiguous(),
                "attn": prob,
                "attnscore": score,
                "attn_out": out_plain.reshape(B, N, -1).permute(0, 2, 1).float().contiguous(),
            }
            if residual is None:
                return out_plain
        if defer_out and not self.save_cross_attn_vars:
            return o, self.to_out[0].packed(), residual          # the consumer (the chained cross-attention block) runs to_out + residual itself
        return self.to_out[0].hip(o, residual=residual)

    # ---- training mode: row-major q|k|v kept for the flash backward
    def _packed_qkv_bwd(self):
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        if not hasattr(self, "_qkv_cache_bwd"):
            self._qkv_cache_bwd = _PackCache()
        return self._qkv_cache_bwd.get(ws, lambda: ops.pack_matrix(torch.cat([w.detach() for w in ws], 0).t().contiguous(), None, ws[0].device))

    def _packed_kv_bwd(self):
        ws = (self.to_k.weight, self.to_v.weight)
        if not hasattr(self, "_kv_cache_bwd"):
            self._kv_cache_bwd = _PackCache()
        return self._kv_cache_bwd.get(ws, lambda: ops.pack_matrix(torch.cat([w.detach() for w in ws], 0).t().contiguous(), None, ws[0].device))

    def hip_train(self, x2d, B, N, context=None, keybias=None, residual=None, ln=None):
        """As hip(), returning (out, saved) where `saved` feeds hip_bwd.  With ``ln`` the projections take the un-normalised rows (the
        backward is unchanged: it differentiates the projection w.r.t. LN's output and the caller's LayerNorm backward does the rest)."""
        Ci, h, d = self.inner_dim, self.heads, self.dim_head
        if context is None:
            qkv = ops.gemm(x2d, self._packed_qkv() if ln is None else self._packed_qkv_ln(ln))       # [M, 3C] row-major
            vt = ops.transpose_tokens(qkv[:, 2 * Ci:], B, N, Ci, 3 * Ci)
            o, lse = ops.attention(qkv, qkv[:, Ci:], vt, B=B, Nq=N, L=N, heads=h, d=d, ldq=3 * Ci, ldk=3 * Ci, keybias=keybias,
                                   scale=self.scale, want_lse=True)
            saved = ("self", qkv, o, lse, keybias, B, N, N)
        else:
            L = context.shape[1]
            q = self.to_q.hip(x2d) if ln is None else ops.gemm(x2d, self._packed_q_ln(ln))
            kv = ops.gemm(context.reshape(B * L, context.shape[-1]), self._packed_kv())  # [B*L, 2C]
            vt = ops.transpose_tokens(kv[:, Ci:], B, L, Ci, 2 * Ci)
            o, lse = ops.attention(q, kv, vt, B=B, Nq=N, L=L, heads=h, d=d, ldq=Ci, ldk=2 * Ci, scale=self.scale, want_lse=True)
            saved = ("cross", q, kv, o, lse, B, N, L)
        return self.to_out[0].hip(o, residual=residual), saved

    def hip_bwd(self, saved, dy, dctx=None):
        """dy [M, C] -> (dx [M, C], dctx [B*L, Cc] accumulated) -- activation gradients only."""
        Ci, h, d = self.inner_dim, self.heads, self.dim_head
        do = self.to_out[0].hip_dgrad(dy)
        if saved[0] == "self":
            _, qkv, o, lse, keybias, B, N, L = saved
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd(qkv, qkv[:, Ci:], qkv[:, 2 * Ci:], o, do, lse, B=B, Nq=N, L=L, heads=h, d=d, ldq=3 * Ci, ldk=3 * Ci,
                              ldv=3 * Ci, dq=dqkv, dk=dqkv[:, Ci:], dv=dqkv[:, 2 * Ci:], lddq=3 * Ci, lddk=3 * Ci, lddv=3 * Ci,
                              keybias=keybias, scale=self.scale)
            return ops.gemm(dqkv, self._packed_qkv_bwd()), dctx
        _, q, kv, o, lse, B, N, L = saved
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attention_bwd(q, kv, kv[:, Ci:], o, do, lse, B=B, Nq=N, L=L, heads=h, d=d, ldq=Ci, ldk=2 * Ci, ldv=2 * Ci, dq=dq,
                          dk=dkv, dv=dkv[:, Ci:], lddq=Ci, lddk=2 * Ci, lddv=2 * Ci, scale=self.scale)
        dctx = ops.gemm(dkv, self._packed_kv_bwd(), residual=dctx)
        return self.to_q.hip_dgrad(dq), dctx

    def forward(self, x, context=None, mask=None):
        """x [b, n, C]; context [b, l, Cc] or None; mask [b, ...] (nonzero = keep) over keys."""
        B, N, _ = x.shape
        x2d = x.reshape(B * N, -1).to(F16).contiguous()
        ctx = None if context is None else context.to(F16).contiguous()
        L = N if ctx is None else ctx.shape[1]
        kb = None if mask is None else ops.make_keybias(mask.reshape(B, -1), L)
        y = self.hip(x2d, B, N, ctx, kb).reshape(B, N, -1)
        return y if x.dtype == F16 else y.to(x.dtype)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)
        self.checkpoint = checkpoint

    def hip(self, x2d, B, N, context=None, keybias=None, post=None):
        """attention.py:242-252 with the three residual adds fused into GEMM epilogues.  post (SpatialTransformer.hip, its LAST block only) = (proj_out pack, x_in 2-D,
        rows per image, gn_cpg): proj_out + residual run as the tail of the one-launch feed-forward where that runs (af_ff_chain) -- the return value is then the
        SpatialTransformer's output and ``post_done(...)`` says so."""
        if FOLD_LAYERNORM and x2d.shape[1] % 64 == 0:
            # the LayerNorms do not run as kernels: each is folded into the projection GEMM that consumes it.  Below ~1,000 rows (the
            # 8 x 8 level) the q | k | v and to_q GEMMs sit on the register-staged 64 x 64 tile, which has no fold, and a 3.8 us LayerNorm
            # is cheaper than moving them to a whole-line tile (profiles/r03d_lnfold_runtime_flag.txt: +2.5 .. +4.0 us folded)
            small = x2d.shape[0] < 1024
            if (CHAIN_XATTN and not small and context is not None and not self.attn1.save_cross_attn_vars 
                    and self.attn2.xattn_fusable(x2d.shape[1], B, N, context.shape[1], self.norm2, chain=True)):
                # the 64 x 64 level: attn1's to_out + residual run as phase 0 of the one-launch cross-attention block (af_xattn_chain)
                pre = self.attn1.hip(x2d, B, N, None, keybias, residual=x2d, ln=self.norm1, defer_out=True)
                x2 = self.attn2.hip(None, B, N, context, None, residual=None, ln=self.norm2, pre=pre)
                return self.ff.hip(x2, residual=x2, ln=self.norm3, post=post if self.post_done(x2d.shape[0], x2d.shape[1], post) else None)
            x1 = self.attn1.hip(self.norm1.hip(x2d) if small else x2d, B, N, None, keybias, residual=x2d, ln=None if small else self.norm1)
            x2 = self.attn2.hip(self.norm2.hip(x1) if small else x1, B, N, context, None, residual=x1, ln=None if small else self.norm2)
            return self.ff.hip(x2, residual=x2, ln=self.norm3, post=post if self.post_done(x2d.shape[0], x2d.shape[1], post) else None)
        x1 = self.attn1.hip(self.norm1.hip(x2d), B, N, None, keybias, residual=x2d)
        x2 = self.attn2.hip(self.norm2.hip(x1), B, N, context, None, residual=x1)
        return self.ff.hip(self.norm3.hip(x2), residual=x2)

    def post_done(self, M, C, post) -> bool:
        """Whether ``hip(..., post=post)`` on [M, C] rows runs the caller's proj_out + residual itself (the one-launch feed-forward with its tail, af_ff_chain)."""
        return bool(post is not None and CHAIN_PROJ_OUT and FOLD_LAYERNORM and C % 64 == 0 and self.ff.ff_fusable(M, C, self.norm3))

    def hip_train(self, x2d, B, N, context=None, keybias=None):
        if FOLD_LAYERNORM and x2d.shape[1] % 64 == 0 and x2d.shape[0] >= 1024:
            # the LayerNorm outputs are not needed by the backward (it re-normalises from the saved inputs): fold them as in hip()
            x1, s1 = self.attn1.hip_train(x2d, B, N, None, keybias, residual=x2d, ln=self.norm1)
            x2, s2 = self.attn2.hip_train(x1, B, N, context, None, residual=x1, ln=self.norm2)
            x3, hp = self.ff.hip_train(x2, residual=x2, ln=self.norm3)
            return x3, (x2d, x1, x2, s1, s2, hp)
        x1, s1 = self.attn1.hip_train(self.norm1.hip(x2d), B, N, None, keybias, residual=x2d)
        x2, s2 = self.attn2.hip_train(self.norm2.hip(x1), B, N, context, None, residual=x1)
        x3, hp = self.ff.hip_train(self.norm3.hip(x2), residual=x2)
        return x3, (x2d, x1, x2, s1, s2, hp)

    def hip_bwd(self, saved, dy, dctx=None):
        """Reverse of attention.py:242-252; every residual add of the backward is fused into the LayerNorm
        backward kernel's `add` input."""
        x, x1, x2, s1, s2, hp = saved
        dx2 = self.norm3.hip_bwd(x2, self.ff.hip_bwd(hp, dy), add=dy)
        dn2, dctx = self.attn2.hip_bwd(s2, dx2, dctx)
        dx1 = self.norm2.hip_bwd(x1, dn2, add=dx2)
        dn1, _ = self.attn1.hip_bwd(s1, dx1, None)
        return self.norm1.hip_bwd(x, dn1, add=dx1), dctx

    def forward(self, x, context=None, mask=None):
        return checkpoint(self._forward, (x, context, mask), self.parameters(), self.checkpoint)

    def _forward(self, x, context=None, mask=None):
        B, N, _ = x.shape
        ctx = None if context is None else context.to(F16).contiguous()
        kb = None if mask is None else ops.make_keybias(mask.reshape(B, -1), N)
        y = self.hip(x.reshape(B * N, -1).to(F16).contiguous(), B, N, ctx, kb).reshape(B, N, -1)
        return y if x.dtype == F16 else y.to(x.dtype)


class SpatialTransformer(nn.Module):
    """GroupNorm -> 1x1 conv -> transformer block(s) over the H*W tokens -> 1x1 conv -> + input
    (attention.py:254-304)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim) for _ in range(depth)]
        )
        self.proj_out = zero_module(Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))

    # The LIVE attention processor (adaface/diffusers_attn_lora_capture.py:254-260) drops the self-attention key mask for the whole batch
    # when, at this layer's resolution, ANY instance's mask is empty; the in-tree LDM U-Net (attention.py:188-194) keeps it and such
    # an instance attends uniformly.  UNetWrapper (the live path's seam) switches this on; a bare UNetModel keeps LDM semantics.
    live_mask_rule = False

    def _keybias(self, mask, B, H, W):
        m2 = F.interpolate(mask.float(), size=(H, W), mode="nearest")       # attention.py:298 / diffusers_attn_lora_capture.py:256-258
        kb = ops.make_keybias(m2.reshape(B, H * W), H * W)
        if self.live_mask_rule:
            any_empty = (m2.sum(dim=(2, 3)) == 0).any()                      # stays on the device: no host decision, graph-capturable
            kb = torch.where(any_empty, torch.zeros_like(kb), kb)
        return kb

    def hip(self, x, context=None, mask=None):
        """x [B,H,W,C] fp16; context [B,L,Cc] fp16; mask [B,1,h0,w0] (nonzero = keep) or None."""
        B, H, W, Cn = x.shape
        N = H * W
        # norm + proj_in as ONE launch where the producer of x left its GroupNorm partials (C = 320: ops.gn_proj_fused); else two launches
        y = ops.gn_proj_fused(x, self.norm.weight, self.norm.bias, self.norm.eps, self.proj_in.packed(), self.norm.num_groups) if FUSE_GN_PROJ else None
        if y is None:
            y = self.norm.hip(x)
            y = self.proj_in.hip(y).reshape(B * N, -1)
        kb = None
        if mask is not None:
            kb = self._keybias(mask, B, H, W)
        # the block's output feeds the next GroupNorm(32) (a ResBlock's first norm or the output norm) when it is not concatenated first: leave
        # the partial statistics with it (ops.GnPartials)
        cpg = Cn // 32 if Cn % 32 == 0 else 0
        last = len(self.transformer_blocks) - 1
        post = (self.proj_out.packed(), x.reshape(B * N, Cn), N, cpg)
        chained = False
        for i, block in enumerate(self.transformer_blocks):
            block.attn2.infeat_size = (H, W)
            if i == last and block.post_done(B * N, y.shape[1], post):
                # round 6: proj_out + residual as the tail of the last block's one-launch feed-forward (C = 320, the 64 x 64 level: af_ff_chain)
                y, chained = block.hip(y, B, N, context, kb, post=post), True
            else:
                y = block.hip(y, B, N, context, kb)
        out = y if chained else ops.gemm(y, self.proj_out.packed(), residual=x.reshape(B * N, Cn), rows_per_batch=N, gn_cpg=cpg)
        out4 = out.reshape(B, H, W, Cn)
        if hasattr(out, "_gn_partials"):
            out4._gn_partials = out._gn_partials      # a view: same storage address and version counter (ops.partials_of)
        return out4

    def hip_train(self, x, context=None, mask=None):
        B, H, W, Cn = x.shape
        N = H * W
        y, st = self.norm.hip_train(x)
        y = self.proj_in.hip(y).reshape(B * N, -1)
        kb = None
        if mask is not None:
            kb = self._keybias(mask, B, H, W)
        bs = []
        for block in self.transformer_blocks:
            block.attn2.infeat_size = (H, W)
            y, s = block.hip_train(y, B, N, context, kb)
            bs.append(s)
        out = ops.gemm(y, self.proj_out.packed(), residual=x.reshape(B * N, Cn))
        return out.reshape(B, H, W, Cn), (x, st, bs)

    def hip_bwd(self, saved, dy, dctx=None):
        x, st, bs = saved
        B, H, W, Cn = x.shape
        d = ops.gemm(dy.reshape(B * H * W, Cn), self.proj_out.packed_bwd())
        for block, s in zip(reversed(self.transformer_blocks), reversed(bs)):
            d, dctx = block.hip_bwd(s, d, dctx)
        dn = ops.gemm(d, self.proj_in.packed_bwd()).reshape(B, H, W, Cn)
        return self.norm.hip_bwd(x, st, dn, silu=False, add=dy), dctx

    def forward(self, x, context=None, mask=None):
        ctx = None if context is None else context.to(F16).contiguous()
        return from_nhwc_f16(self.hip(to_nhwc_f16(x), ctx, mask), x.dtype)
