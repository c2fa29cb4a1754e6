"""The U-Net pass of the Stage-2 (compositional distillation) iterations: activations of the last three cross-attention layers
captured WITH gradients, their scores optionally rewritten, attention LoRAs optionally trainable.

Reference: ``DiffusersUNetWrapper.forward`` (ldm/models/diffusion/ddpm.py:4187-4252) with ``capture_ca_activations`` under autograd,
``AttnProcessor_LoRA_Capture.__call__`` and ``scaled_dot_product_attention`` (adaface/diffusers_attn_lora_capture.py:79-139, 192-364),
``CrossAttnUpBlock2D_forward_capture`` (:366-446).  The captured layers are the cross-attention layers of diffusers ``up_blocks.3``
= LDM ``output_blocks`` 9, 10, 11 = layer indices 22, 23, 24 (openaimodel.py:842-857).

The Stage-1 training pass is ONE autograd node over the whole U-Net (``openaimodel._UNetFunction``).  Losses on captured activations
need gradient entry points inside the last three decoder blocks, so this pass is a short CHAIN of nodes instead, each still a manual
forward / backward over the same HIP kernels:

    _TrunkFn            time embedding, encoder, middle block, decoder blocks 0 .. 8        -> h, and the three skips the tail consumes
    per tail block      _ResBlockFn (skip gradient scaled, FFN DoRA adapters as in Stage 1)
                        _STPreFn     GroupNorm, proj_in, self-attention sub-block, norm2    -> x1, LN2(x1)
                        to_q / to_k / to_v  (autograd_ops.LinearFn, or the DoRA linear when the attention LoRAs are on)
                        _ScoresFn    score = scale q k^T                                     (csrc/af_xattn_explicit.hip)
                        rewrite of the scores: SC/MC mixing or subject-token normalisation   (index bookkeeping, torch)
                        _SoftmaxPVFn prob, o = softmax(score) v
                        to_out, + x1
                        _STPostFn    norm3, GEGLU feed-forward, proj_out, + block input
    _HeadFn             GroupNorm + SiLU + the 320 -> 4 convolution

Activations between nodes are fp16 (NHWC / token-major), their gradients too; as in the reference's fp16 autocast run the loss is
scaled by the trainer's LossScaler.  Captured tensors are returned in the reference's layouts ([B, C, N] for q / q2 / k / v / attn_out,
[B, heads, N, L] for attn / attnscore, [B, C, H, W] for outfeat) as ordinary autograd tensors."""
import math
import os

import torch
import torch.nn.functional as F

from .... import ops
from ....ops import F16
from .util import from_nhwc_f16, to_nhwc_f16

CAPTURE_KEYS = ("outfeat", "attn", "attnscore", "q", "q2", "k", "v", "attn_out")
INFER_TRUNK = os.environ.get("AF_INFER_TRUNK", "1") != "0"      # A/B switch: 0 = the activation-saving trunk for every captured pass


class ScaleGrad(torch.autograd.Function):
    """Identity forward, gradient times alpha (diffusers_attn_lora_capture.py:23-42)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = float(alpha)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.alpha, None


def gen_gradient_scaler(alpha):
    """diffusers_attn_lora_capture.py:62-70."""
    if alpha == 1:
        return lambda x: x
    if alpha == 0:
        return torch.detach
    return lambda x: ScaleGrad.apply(x, alpha)


def rewrite_scores(score, cross_attn_scale_factor, subj_indices=None, normalize_cross_attn=False, mix_attn_mats_in_batch=False):
    """The two score rewrites of the explicit attention (diffusers_attn_lora_capture.py:108-133), on score fp32 [B, heads, N, L].
    mix_attn_mats_in_batch: the batch is [SC ..., MC ...]; both halves get (SC + MC.detach()) / 2.
    normalize_cross_attn: the columns of the subject tokens ``subj_indices = (batch indices, token indices)`` are centred over the N
    pixels (mean detached) and multiplied by the learnable factor, whose gradient is scaled x10."""
    if mix_attn_mats_in_batch:
        if score.shape[0] % 2 != 0:
            raise ValueError("mix_attn_mats_in_batch needs an even batch [SC..., MC...]")
        sc, mc = score.chunk(2, dim=0)
        return ((sc + mc.detach()) / 2).repeat(2, 1, 1, 1)
    if normalize_cross_attn:
        if subj_indices is None:
            raise ValueError("normalize_cross_attn needs subj_indices")
        b, n = subj_indices
        sub = score[b, :, :, n]
        sub = sub - sub.mean(dim=2, keepdim=True).detach()
        sub = sub * gen_gradient_scaler(10)(cross_attn_scale_factor.to(sub.dtype))
        out = score.clone()
        out[b, :, :, n] = sub
        return out
    return score


# ----------------------------------------------------------------------------- explicit attention nodes
class _ScoresFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, B, N, L, heads, d, scale):
        ctx.save_for_backward(q, k)
        ctx.cfg = (B, N, L, heads, d, scale)
        return ops.xattn_scores(q, k, B=B, Nq=N, L=L, heads=heads, d=d, scale=scale)

    @staticmethod
    def backward(ctx, dscore):
        q, k = ctx.saved_tensors
        B, N, L, heads, d, scale = ctx.cfg
        dscore = dscore.to(torch.float32).contiguous()
        kw = dict(B=B, Nq=N, L=L, heads=heads, d=d)
        dq = ops.xattn_rowmix(dscore, k, scale, **kw) if ctx.needs_input_grad[0] else None
        dk = ops.xattn_colmix(dscore, q, scale, **kw) if ctx.needs_input_grad[1] else None
        return dq, dk, None, None, None, None, None, None


class _SoftmaxPVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, v, B, N, L, heads, d):
        prob, o = ops.xattn_softmax_pv(score.contiguous(), v, B=B, Nq=N, L=L, heads=heads, d=d)
        ctx.save_for_backward(prob, v)
        ctx.cfg = (B, N, L, heads, d)
        return prob, o

    @staticmethod
    def backward(ctx, dprob, do):
        prob, v = ctx.saved_tensors
        B, N, L, heads, d = ctx.cfg
        kw = dict(B=B, Nq=N, L=L, heads=heads, d=d)
        if do is None:
            do = torch.zeros((B * N, heads * d), dtype=F16, device=prob.device)
        do = do.to(F16).contiguous()
        dscore = ops.xattn_softmax_pv_bwd(prob, v, do, dprob, **kw) if ctx.needs_input_grad[0] else None
        dv = ops.xattn_colmix(prob, do, 1.0, **kw) if ctx.needs_input_grad[1] else None
        return dscore, dv, None, None, None, None, None


# ----------------------------------------------------------------------------- manual nodes over existing forward / backward walks
class _TrunkFn(torch.autograd.Function):
    """Everything below the captured tail: outputs (h, skip_0 .. skip_{n_tail-1}) in the order the tail pops them."""

    @staticmethod
    def forward(ctx, unet, x, emb, context, img_mask, n_tail, res_gradscale):
        xh = to_nhwc_f16(x.detach(), ops.round_up(unet.in_channels, 8))
        h, skips, saved = unet.hip_train_trunk(xh, emb, context.detach().to(F16).contiguous(), img_mask, n_tail)
        ctx.unet, ctx.saved, ctx.n_tail, ctx.res_gradscale = unet, saved, n_tail, res_gradscale
        ctx.need_dx, ctx.x_dtype, ctx.c_dtype = x.requires_grad, x.dtype, context.dtype
        return (h,) + tuple(skips)

    @staticmethod
    def backward(ctx, dh, *dskips):
        unet = ctx.unet
        zeros = lambda g, like: torch.zeros_like(like) if g is None else g
        # one power-of-two rescale for the whole walk, as _UNetFunction does (long fp16 chain)
        gs = [g for g in (dh,) + dskips if g is not None]
        amax = torch.stack([g.detach().abs().amax().float() for g in gs]).amax().clamp_min(1e-30)
        scale = torch.exp2(torch.floor(torch.log2(256.0 / amax))).clamp(max=2.0 ** 24)
        sc = lambda g: None if g is None else (g.float() * scale).to(F16).contiguous()
        dx, dctx = unet.hip_bwd_trunk(ctx.saved, sc(dh), [sc(g) for g in dskips], need_dx=ctx.need_dx, res_gradscale=ctx.res_gradscale)
        ctx.saved = None
        gx = (from_nhwc_f16(dx, torch.float32, unet.in_channels) / scale).to(ctx.x_dtype) if dx is not None else None
        return None, gx, None, (dctx.float() / scale).to(ctx.c_dtype), None, None, None


class _ResBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, block, h, skip, emb, lora, *lora_params):
        from .openaimodel import SkipCat
        block._lora = lora
        try:
            out, saved = block.hip_train(SkipCat((h, skip)), emb)
        finally:
            block._lora = None
        ctx.block, ctx.saved, ctx.lora = block, saved, lora
        return out

    @staticmethod
    def backward(ctx, dy):
        dh, dskip = ctx.block.hip_bwd(ctx.saved, dy.contiguous())
        grads = []
        if ctx.lora:
            g = ctx.saved[5][4]
            for key in sorted(ctx.lora):
                for pname in ("lora_A", "lora_B", "lora_magnitude_vector"):
                    grads.append(g[key][pname].to(getattr(ctx.lora[key], pname).dtype))
        ctx.saved = None
        return (None, dh, dskip, None, None) + tuple(grads)


def resblock_lora_params(lora):
    return [getattr(lora[key], pname) for key in sorted(lora or {}) for pname in ("lora_A", "lora_B", "lora_magnitude_vector")]


class _STPreFn(torch.autograd.Function):
    """SpatialTransformer up to the cross-attention's query input: GroupNorm(1e-6), proj_in, x1 = attn1(LN1(y)) + y, LN2(x1)."""

    @staticmethod
    def forward(ctx, st, x, keybias):
        B, H, W, Cn = x.shape
        N = H * W
        blk = st.transformer_blocks[0]
        y, gst = st.norm.hip_train(x)
        y = st.proj_in.hip(y).reshape(B * N, -1)
        x1, s1 = blk.attn1.hip_train(blk.norm1.hip(y), B, N, None, keybias, residual=y)
        qin = blk.norm2.hip(x1)
        ctx.st, ctx.saved = st, (x, gst, y, s1, x1)
        return x1, qin

    @staticmethod
    def backward(ctx, dx1, dqin):
        st = ctx.st
        blk = st.transformer_blocks[0]
        x, gst, y, s1, x1 = ctx.saved
        B, H, W, Cn = x.shape
        if dqin is None:
            d1 = dx1.contiguous()
        else:
            d1 = blk.norm2.hip_bwd(x1, dqin.contiguous(), add=None if dx1 is None else dx1.contiguous())
        dn1, _ = blk.attn1.hip_bwd(s1, d1, None)
        dy = blk.norm1.hip_bwd(y, dn1, add=d1)
        dn = ops.gemm(dy, st.proj_in.packed_bwd()).reshape(B, H, W, Cn)
        ctx.saved = None
        return None, st.norm.hip_bwd(x, gst, dn, silu=False), None


class _STPostFn(torch.autograd.Function):
    """x3 = FF(LN3(x2)) + x2 ; out = proj_out(x3) + x_in."""

    @staticmethod
    def forward(ctx, st, x2, x_in):
        B, H, W, Cn = x_in.shape
        blk = st.transformer_blocks[0]
        x3, hp = blk.ff.hip_train(blk.norm3.hip(x2), residual=x2)
        out = ops.gemm(x3, st.proj_out.packed(), residual=x_in.reshape(B * H * W, Cn))
        ctx.st, ctx.saved, ctx.shape = st, (x2, hp), (B, H, W, Cn)
        return out.reshape(B, H, W, Cn)

    @staticmethod
    def backward(ctx, dout):
        st = ctx.st
        blk = st.transformer_blocks[0]
        x2, hp = ctx.saved
        B, H, W, Cn = ctx.shape
        dout = dout.contiguous()
        d = ops.gemm(dout.reshape(B * H * W, Cn), st.proj_out.packed_bwd())
        dx2 = blk.norm3.hip_bwd(x2, blk.ff.hip_bwd(hp, d), add=d)
        ctx.saved = None
        return None, dx2, dout


class _HeadFn(torch.autograd.Function):
    """GroupNorm + SiLU + the final 3x3 convolution (openaimodel.py:686-690)."""

    @staticmethod
    def forward(ctx, unet, h):
        g, st = unet.out[0].hip_train(h, silu=True)
        ctx.unet, ctx.saved = unet, (h, st)
        return unet.out[2].hip(g)

    @staticmethod
    def backward(ctx, deps):
        unet = ctx.unet
        h, st = ctx.saved
        pad = ops.round_up(unet.out_channels, 8)
        d = deps.contiguous()
        if d.shape[-1] != pad:
            d = F.pad(d, (0, pad - d.shape[-1]))
        return None, unet.out[0].hip_bwd(h, st, unet.out[2].hip_dgrad(d), silu=True)


class _NHWCToNCHW(torch.autograd.Function):
    """fp16 [B,H,W,C] -> NCHW in `dtype` keeping the first `channels` (the eps output and the outfeat captures)."""

    @staticmethod
    def forward(ctx, y, dtype, channels):
        ctx.cfg = (y.shape, y.dtype)
        return from_nhwc_f16(y, dtype, channels).contiguous()

    @staticmethod
    def backward(ctx, g):
        shape, dt = ctx.cfg
        out = torch.zeros(shape, dtype=dt, device=g.device)
        out[..., :g.shape[1]] = g.permute(0, 2, 3, 1).to(dt)
        return out, None, None


class _FrozenLinearFn(torch.autograd.Function):
    """y = x W^T + b of a FROZEN layer (the U-Net's base weights, ddpm.py:4131-4132): gradient to the input only."""

    @staticmethod
    def forward(ctx, x, mod):
        ctx.mod = mod
        return mod.hip(x)

    @staticmethod
    def backward(ctx, dy):
        return ctx.mod.hip_dgrad(dy.to(F16).contiguous()), None


# ----------------------------------------------------------------------------- the pass
def _linear(mod, x, lora=None, generator=None):
    """frozen Linear with gradient to its input; `lora`: a DoRA linear adapter (modules/dora.py) trained through this call."""
    if lora is not None:
        from ..dora import dora_linear
        return dora_linear(mod, lora, x, generator=generator)
    return _FrozenLinearFn.apply(x, mod)


def captured_cross_attention(attn2, qin, context2d, B, N, flags, loras=None):
    """The explicit attention of one captured layer.  qin [B*N, C] fp16 (= LN2(x1)), context2d [B*L, Cc] fp16 -> (attention output
    before the residual [B*N, C] fp16, captures dict).  ``flags``: normalize_cross_attn, mix_attn_mats_in_batch, subj_indices,
    cross_attn_scale_factor, q_lora_updates_query; ``loras``: {'q' | 'k' | 'v' | 'out': adapter} when the attention LoRAs are on."""
    loras = loras or {}
    C, heads, d = attn2.inner_dim, attn2.heads, attn2.dim_head
    L = context2d.shape[0] // B
    scale = attn2.scale
    q = _linear(attn2.to_q, qin)
    q2 = q
    if "q" in loras:                               # the q LoRA feeds query2 only, unless q_lora_updates_query (:239-249)
        q2 = _linear(attn2.to_q, qin, loras["q"])
        if flags.get("q_lora_updates_query", False):
            q = q2
    k = _linear(attn2.to_k, context2d, loras.get("k"))
    v = _linear(attn2.to_v, context2d, loras.get("v"))
    score = _ScoresFn.apply(q, k, B, N, L, heads, d, scale)
    score = rewrite_scores(score, flags.get("cross_attn_scale_factor"), flags.get("subj_indices"), flags.get("normalize_cross_attn", False),
                           flags.get("mix_attn_mats_in_batch", False))
    prob, o = _SoftmaxPVFn.apply(score, v, B, N, L, heads, d)
    out = _linear(attn2.to_out[0], o, loras.get("out"))
    rs = math.sqrt(scale)
    chan_first = lambda t, n: t.reshape(B, n, C).permute(0, 2, 1).float() * rs          # 'b h n d -> b (h d) n' * sqrt(scale)  (:347-354)
    caps = {"q": chan_first(q, N), "q2": chan_first(q2, N), "k": chan_first(k, L), "v": chan_first(v, L), "attn": prob, "attnscore": score,
            "attn_out": out.reshape(B, N, -1).permute(0, 2, 1).float()}
    return out, caps


def cached_trunk(extra_info, B):
    """(h, skips) of this call's rows from the step's shared trunk (LatentDiffusion.guided_denoise puts ``_trunk_cache`` = {row key: (h, skips)}
    into extra_info, each call names its rows in ``_trunk_rows``), or None when there is none / a row is missing / the batch does not match."""
    if not extra_info:
        return None
    cache, rows = extra_info.get("_trunk_cache"), extra_info.get("_trunk_rows")
    if cache is None or rows is None or len(rows) != B or any(r not in cache for r in rows):
        return None
    if len(rows) == 1:
        return cache[rows[0]]
    hs = [cache[r] for r in rows]
    return torch.cat([c[0] for c in hs], dim=0), [torch.cat([c[1][i] for c in hs], dim=0) for i in range(len(hs[0][1]))]


def unet_forward_captured(unet, x, timesteps, context, extra_info, n_tail=3):
    """eps [B, out_channels, H, W] in x.dtype; fills extra_info['ca_layers_activations'] = {key: {layer index: tensor}}."""
    from .openaimodel import lora_param_order  # noqa: F401  (same adapter ordering as the Stage-1 node)
    ei = extra_info
    img_mask = ei.get("img_mask")
    gs = float(ei.get("res_hidden_states_gradscale", 1) or 1)
    ffn_lora = ei.get("_ffn_lora_adapters") or {}
    attn_loras = ei.get("_attn_lora_adapters") or {}
    n_out = len(unet.output_blocks)
    first_layer = len(unet.input_blocks) + 1 + n_out - n_tail          # layer index of the first tail block (22 for SD-1.5)
    emb = unet._embed(timesteps)
    if not INFER_TRUNK or (torch.is_grad_enabled() and (x.requires_grad or context.requires_grad)):
        outs = _TrunkFn.apply(unet, x, emb, context, img_mask, n_tail, gs)
        h, skips = outs[0], outs[1:]
    else:
        # nothing below the tail needs a gradient (the no-grad instances of a compositional step, the class-prompt pass of a recon
        # step): the inference walk instead of the activation-saving one
        shared = cached_trunk(ei, x.shape[0]) if img_mask is None and n_tail == 3 else None
        if shared is not None:
            h, skips = shared                     # this step's shared gradient-free trunk already holds these rows
        else:
            with torch.no_grad():
                h, skips = unet.hip_trunk(to_nhwc_f16(x.detach(), ops.round_up(unet.in_channels, 8)), emb, context.detach().to(F16).contiguous(), img_mask, n_tail)
    B = x.shape[0]
    ctx2d = context.to(F16).reshape(B * context.shape[1], context.shape[2])
    acts = {k: {} for k in CAPTURE_KEYS}
    for ti in range(n_tail):
        bi = n_out - n_tail + ti
        block = unet.output_blocks[bi]
        res, st = block[0], block[1]
        skip = skips[ti]
        if gs != 1.0:
            skip = ScaleGrad.apply(skip, gs)
        lora = ffn_lora.get(bi)
        h = _ResBlockFn.apply(res, h, skip, emb, lora, *resblock_lora_params(lora))
        Bh, H, W, Cn = h.shape
        N = H * W
        kb = None
        if img_mask is not None:
            kb = ops.make_keybias(F.interpolate(img_mask.float(), size=(H, W), mode="nearest").reshape(B, N), N)
        x1, qin = _STPreFn.apply(st, h, kb)
        attn2 = st.transformer_blocks[0].attn2
        factors = ei.get("_cross_attn_scale_factors")            # the wrapper's learnable factors (init 0.8, :168); plain 0.8 without a wrapper
        factor = factors[ti] if factors is not None else torch.tensor(0.8, device=x.device)
        flags = dict(normalize_cross_attn=ei.get("normalize_cross_attn", False), mix_attn_mats_in_batch=ei.get("mix_attn_mats_in_batch", False),
                     subj_indices=ei.get("subj_indices"), cross_attn_scale_factor=factor,
                     q_lora_updates_query=ei.get("q_lora_updates_query", False))
        ao, caps = captured_cross_attention(attn2, qin, ctx2d, B, N, flags, attn_loras.get(bi))
        x2 = ao + x1
        h = _STPostFn.apply(st, x2, h)
        caps["outfeat"] = _NHWCToNCHW.apply(h, torch.float32, Cn)
        for k, v in caps.items():
            acts[k][first_layer + ti] = v
    eps = _HeadFn.apply(unet, h)
    ei["ca_layers_activations"] = acts
    return _NHWCToNCHW.apply(eps, x.dtype, unet.out_channels)
