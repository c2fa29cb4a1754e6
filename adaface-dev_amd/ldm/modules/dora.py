"""Trainable DoRA adapters on convolutions of the U-Net: the reference attaches peft DoRA ``Conv2d``
adapters (rank 192, ``lora_alpha`` 16, dropout 0.1, three adapter names) to ``up_blocks.3.resnets.{1,2}.{conv1,conv2,conv_shortcut}``
(``adaface/diffusers_attn_lora_capture.py:541-591``) and ALWAYS enables the ``unet_distill`` one in Stage-1 distillation
(``ddpm.py:3130-3134``).

peft (third party, absent here, unpinned in the reference) computes, with ``xd = dropout(x)``, ``scaling = alpha / r`` and
``s_c = m_c / ||W + scaling * B A||_c`` (norm detached):

    y = base(x) + (s - 1) * conv(xd, W) + s * scaling * B(A(xd))                       (DoraConv2dLayer.forward)

This module runs that forward and its backward on the gfx950 kernels: the three convolutions and their input gradients are ``af_gemm``
launches (A: 3x3 or 1x1 Cin -> r; B: 1x1 r -> Cout), the combination is ``af_dora_combine``, dropout is ``af_mul_f16`` with a
pre-drawn mask, the weight gradients are GEMMs over pixels: dB = d(lb)^T . t,  dA = d(t)^T . im2col(xd)  (``af_im2col3x3`` +
``autograd_ops.wgrad``), dm = colsum(dy * (c2 + scaling * lb)) / norm.  Parameters are named like peft's so reference checkpoints map
one to one (``adaface/lora.py::extract_adapter``)."""
import torch
import torch.nn as nn

from ... import ops
from ...autograd_ops import low_rank_product, wgrad
from ...ops import F16


def _cached_scales(ad, base, compute):
    """The DoRA scale vectors of adapter ``ad`` on layer ``base`` depend on (W, A, B, m) only, yet every pass of a step asked for them again: a weight
    cast, a rank-192 product, a row norm and five element-wise launches per adapter and pass (tools/torch_op_census.py: ~900 launches per Stage-2
    micro-batch).  Computed once per parameter state; the buffers are REFRESHED IN PLACE when a parameter changes (as the weight packs are), so a
    captured hipGraph that read them keeps reading the right addresses -- ``_packs(refresh_scales=True)`` (called before every replay) refreshes
    them too.  The three vectors are the rows of ONE [3, Cout] buffer.  That buffer belongs to the cache: it is rewritten whenever a parameter
    (or the base weight -- e.g. while inference adapters are merged into it, adaface/lora.py) changes, so a forward that needs the values in its
    backward keeps a SNAPSHOT (``snapshot_scales``: one copy launch), never the buffer itself."""
    key = tuple(ops.param_key(t) for t in (base.weight, ad.lora_A, ad.lora_B, ad.lora_magnitude_vector))
    if getattr(ad, "_scale_key", None) != key:
        new = torch.stack(compute())
        old = getattr(ad, "_sc", None)
        if old is not None and old.shape == new.shape and old.device == new.device:
            old.copy_(new)
        else:
            ad._sc = new
        ad._scale_key = key
        object.__setattr__(ad, "_scale_base", base)       # (a plain attribute: nn.Module.__setattr__ would register the base layer as a sub-module)
    return ad._sc[0], ad._sc[1], ad._sc[2]


def snapshot_scales(ad, base):
    """(u, v, norm) of ``ad.scales(base)`` as rows of a private copy: what a forward hands to its backward.  The cache buffer may be refreshed in
    place between the two (a gradient-free pass with merged adapters in between changes ``base.weight``'s version twice)."""
    ad.scales(base)
    sc = ad._sc.clone()
    return sc[0], sc[1], sc[2]


class DoRAConvAdapter(nn.Module):
    def __init__(self, conv, rank=192, lora_alpha=16, lora_dropout=0.1, generator=None):
        super().__init__()
        from ...adaface.lora import init_dora_adapter
        a, b, m = init_dora_adapter(conv.weight, rank, generator)
        self.lora_A = nn.Parameter(a)                        # [r, Cin, k, k]
        self.lora_B = nn.Parameter(b)                        # [Cout, r, 1, 1]   (zero: the adapter starts as the identity)
        self.lora_magnitude_vector = nn.Parameter(m)         # [Cout]            (||W|| per output channel)
        self.rank, self.scaling, self.p = rank, lora_alpha / rank, lora_dropout
        self.k = conv.kernel_size[0]

    # ---- packs (rebuilt when the parameters change: once per optimizer step)
    def _packs(self, refresh_scales=False):
        """refresh_scales: also bring the cached scale vectors up to date (the graph-replay path, where no forward code runs that would ask for
        them).  The backward must NOT do that: between a forward and its backward the base weight may be in its merged state."""
        key = (ops.param_key(self.lora_A), ops.param_key(self.lora_B))
        if getattr(self, "_pack_key", None) != key:
            dev = self.lora_A.device
            A, Bm = self.lora_A.detach(), self.lora_B.detach().flatten(1)
            if self.k == 3:
                pa = ops.pack_conv3x3(A, None, dev)
                pa_t = ops.pack_conv3x3(A.flip(2, 3).permute(1, 0, 2, 3).contiguous(), None, dev, ops.round_up(self.rank, 8))
            else:
                pa = ops.pack_matrix(A.flatten(1), None, dev)
                pa_t = ops.pack_matrix(A.flatten(1).t().contiguous(), None, dev)
            new = (pa, pa_t, ops.pack_matrix(Bm, None, dev), ops.pack_matrix(Bm.t().contiguous(), None, dev))
            old = getattr(self, "_pk", None)
            if old is not None and all(o.wt.shape == n.wt.shape and o.wt.device == n.wt.device and not o.aliased for o, n in zip(old, new)):
                for o, n in zip(old, new):          # same buffers, new values: captured hipGraphs keep reading these addresses
                    o.wt.copy_(n.wt)
            else:
                self._pk = new
            self._pack_key = key
        if refresh_scales and getattr(self, "_scale_base", None) is not None:
            self.scales(self._scale_base)                  # (a graph replay follows: the scale buffers must be current too)
        return self._pk

    def scales(self, conv):
        """(u = s - 1, v = s * scaling, norm) fp32 [Cout]; s = m / ||W + scaling * B A|| with the norm detached (peft).  A function of the
        parameters alone: kept until one of them changes (_cached_scales)."""
        return _cached_scales(self, conv, lambda: self._scales(conv))

    def _scales(self, conv):
        w = conv.weight.detach().float()
        delta = low_rank_product(self.lora_B.detach().flatten(1), self.lora_A.detach().flatten(1)).reshape(w.shape)
        norm = (w + self.scaling * delta).flatten(1).norm(dim=1)
        s = self.lora_magnitude_vector.detach().float() / norm
        return (s - 1).contiguous(), (s * self.scaling).contiguous(), norm

    def draw_mask(self, shape, device, generator=None):
        """Dropout mask with values 0 or 1 / (1 - p) (None in eval mode / p = 0)."""
        if not self.training or self.p == 0:
            return None
        keep = torch.rand(shape, device=device, generator=generator) >= self.p
        return (keep.to(torch.float32) / (1 - self.p)).to(F16)


def _conv_nobias(conv, x):
    pw = conv.packed()
    pw0 = ops.PackedWeight(pw.wt, None, pw.N, pw.K, pw.kpad, pw.taps, pw.cin)
    if conv.kernel_size == (3, 3):
        return ops.conv3x3(x, pw0, stride=conv.stride[0])
    B, H, W, c = x.shape
    return ops.gemm(x.reshape(B * H * W, c), pw0).reshape(B, H, W, -1)


def dora_conv_fwd(conv, ad, x, x2=None, rowbias=None, residual=None, mask=None):
    """y = base(x) + (s - 1) * conv(xd, W) + s * scaling * B(A(xd)).  x (+ x2 concatenated along channels) NHWC fp16;
    rowbias / residual are fused into the base convolution exactly as without adapters.  Returns (y, saved)."""
    xin = x if x2 is None else torch.cat([x, x2], dim=-1)
    y0 = conv.hip(x, x2=x2, rowbias=rowbias, residual=residual)
    xd = xin if mask is None else ops.mul(xin, mask)
    pa, _, pb, _ = ad._packs()
    c2 = _conv_nobias(conv, xd)
    Bn, H, W, _ = xd.shape
    if ad.k == 3:
        t = ops.conv3x3(xd, pa, stride=conv.stride[0])
    else:
        t = ops.gemm(xd.reshape(Bn * H * W, -1), pa).reshape(Bn, H, W, -1)
    M = t.shape[0] * t.shape[1] * t.shape[2]
    lb = ops.gemm(t.reshape(M, ad.rank), pb).reshape(c2.shape)
    u, v, norm = snapshot_scales(ad, conv)              # a private copy: `saved` outlives in-place refreshes of the cache
    y = ops.dora_combine(y0, c2, lb, u, v)
    return y, (xd, c2, lb, t, mask, u, v, norm)


def dora_conv_bwd(conv, ad, saved, dy):
    """-> (dx [B,H,W,Cin] w.r.t. the (concatenated) input, {lora_A, lora_B, lora_magnitude_vector: fp32 gradients})."""
    xd, c2, lb, t, mask, u, v, norm = saved
    Bn, Ho, Wo, cout = dy.shape
    M = Bn * Ho * Wo
    zeros = torch.zeros_like(u)
    _, pa_t, _, pb_t = ad._packs()
    dy2 = dy.reshape(M, cout)
    dc2 = ops.affine_prelu(dy, u, zeros)
    dlb = ops.affine_prelu(dy, v, zeros).reshape(M, cout)
    dt = ops.gemm(dlb, pb_t)                                               # [M, r]
    if ad.k == 3:
        dxa = ops.conv3x3(dt.reshape(Bn, Ho, Wo, ad.rank), pa_t)
    else:
        dxa = ops.gemm(dt, pa_t).reshape(Bn, Ho, Wo, -1)
    dxd = ops.add(conv.hip_dgrad(dc2), dxa)
    if mask is not None:
        dxd = ops.mul(dxd, mask)
    dx = ops.add(conv.hip_dgrad(dy), dxd)
    # ---- parameter gradients
    t2 = t.reshape(M, ad.rank)
    dB = wgrad(dlb, t2).float().reshape(ad.lora_B.shape)
    if ad.k == 3:
        cin = xd.shape[-1]
        dA = wgrad(dt, ops.im2col3x3(xd, conv.stride[0])).float().reshape(ad.rank, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
    else:
        dA = wgrad(dt, xd.reshape(M, -1)).float().reshape(ad.lora_A.shape)
    dm = (ops.colsum(dy2, c2.reshape(M, cout)) + ad.scaling * ops.colsum(dy2, lb.reshape(M, cout))) / norm
    return dx, {"lora_A": dA, "lora_B": dB, "lora_magnitude_vector": dm}


class UNetLoRA(nn.Module):
    """The FFN DoRA adapters of one U-Net: for every adapter name, one DoRAConvAdapter on conv1 / conv2 / conv_shortcut of the last
    two output blocks' ResBlocks (diffusers ``up_blocks.3.resnets.{1,2}``; reference ``set_up_ffn_loras``,
    diffusers_attn_lora_capture.py:541-591: three adapters ``recon_loss`` / ``unet_distill`` / ``comp_distill``)."""

    CONVS = (("conv1", lambda rb: rb.in_layers[2]), ("conv2", lambda rb: rb.out_layers[3]), ("conv_shortcut", lambda rb: rb.skip_connection))

    def __init__(self, unet, adapter_names=("recon_loss", "unet_distill", "comp_distill"), rank=192, lora_alpha=16, lora_dropout=0.1):
        super().__init__()
        n_out = len(unet.output_blocks)
        self.blocks = (n_out - 2, n_out - 1)
        self.adapters = nn.ModuleDict()
        for name in adapter_names:
            d = nn.ModuleDict()
            for ri, bi in enumerate(self.blocks, start=1):
                rb = unet.output_blocks[bi][0]
                for key, get in self.CONVS:
                    conv = get(rb)
                    if isinstance(conv, nn.Conv2d):
                        d[f"up_blocks_3_resnets_{ri}_{key}"] = DoRAConvAdapter(conv, rank, lora_alpha, lora_dropout)
            self.adapters[name] = d

    def active(self, adapter_name):
        """{output-block index: {conv key: adapter}} of one adapter name (what _UNetFunction consumes)."""
        out = {}
        for k, ad in self.adapters[adapter_name].items():
            ri, key = int(k.split("_")[4]), k.split("_", 5)[5]
            out.setdefault(self.blocks[ri - 1], {})[key] = ad
        return out

    def peft_state_dict(self, adapter_name=None):
        """peft / diffusers naming: ``up_blocks.3.resnets.1.conv1.lora_A.<adapter>.weight`` ..."""
        sd = {}
        for name, d in self.adapters.items():
            if adapter_name is not None and name != adapter_name:
                continue
            for k, ad in d.items():
                parts = k.split("_", 5)
                target = f"up_blocks.3.resnets.{parts[4]}.{parts[5]}"
                sd[f"{target}.lora_A.{name}.weight"] = ad.lora_A.detach()
                sd[f"{target}.lora_B.{name}.weight"] = ad.lora_B.detach()
                sd[f"{target}.lora_magnitude_vector.{name}.weight"] = ad.lora_magnitude_vector.detach()
        return sd

    @torch.no_grad()
    def load_peft_state_dict(self, sd):
        for name, d in self.adapters.items():
            for k, ad in d.items():
                parts = k.split("_", 5)
                target = f"up_blocks.3.resnets.{parts[4]}.{parts[5]}"
                for pname in ("lora_A", "lora_B", "lora_magnitude_vector"):
                    key = f"{target}.{pname}.{name}.weight"
                    if key not in sd and pname == "lora_magnitude_vector":
                        key = f"{target}.{pname}.{name}"
                    if key in sd:
                        getattr(ad, pname).copy_(sd[key].reshape(getattr(ad, pname).shape))


# ----------------------------------------------------------------------------- attention LoRAs (Linear layers)
class DoRALinearAdapter(nn.Module):
    """peft DoRA adapter of one ``nn.Linear`` (the to_q / to_k / to_v / to_out.0 projections of the captured cross-attention layers;
    reference ``AttnProcessor_LoRA_Capture.__init__`` diffusers_attn_lora_capture.py:171-181: rank 192, alpha rank // 8, dropout 0.1).
    Parameters are stored like peft's Linear adapter: lora_A [r, Cin], lora_B [Cout, r], lora_magnitude_vector [Cout]."""

    def __init__(self, linear, rank=192, lora_alpha=24, lora_dropout=0.1, generator=None):
        super().__init__()
        from ...adaface.lora import init_dora_adapter
        a, b, m = init_dora_adapter(linear.weight, rank, generator)
        self.lora_A, self.lora_B, self.lora_magnitude_vector = nn.Parameter(a), nn.Parameter(b), nn.Parameter(m)
        self.rank, self.scaling, self.p = rank, lora_alpha / rank, lora_dropout

    def _packs(self, refresh_scales=False):
        key = (ops.param_key(self.lora_A), ops.param_key(self.lora_B))
        if getattr(self, "_pack_key", None) != key:
            dev = self.lora_A.device
            A, Bm = self.lora_A.detach(), self.lora_B.detach()
            self._pk = (ops.pack_matrix(A, None, dev), ops.pack_matrix(A.t().contiguous(), None, dev), ops.pack_matrix(Bm, None, dev),
                        ops.pack_matrix(Bm.t().contiguous(), None, dev))
            self._pack_key = key
        if refresh_scales and getattr(self, "_scale_base", None) is not None:
            self.scales(self._scale_base)
        return self._pk

    def scales(self, linear):
        return _cached_scales(self, linear, lambda: self._scales(linear))

    def _scales(self, linear):
        w = linear.weight.detach().float()
        norm = (w + self.scaling * low_rank_product(self.lora_B.detach(), self.lora_A.detach())).norm(dim=1)
        s = self.lora_magnitude_vector.detach().float() / norm
        return (s - 1).contiguous(), (s * self.scaling).contiguous(), norm

    def draw_mask(self, shape, device, generator=None):
        if not self.training or self.p == 0:
            return None
        keep = torch.rand(shape, device=device, generator=generator) >= self.p
        return (keep.to(torch.float32) / (1 - self.p)).to(F16)


class _DoRALinearFn(torch.autograd.Function):
    """y = base(x) + (s - 1) * (xd W^T) + s * scaling * (xd A^T) B^T with xd = dropout(x) -- peft's DoraLinearLayer in training form,
    the 1x1 case of dora_conv_fwd / dora_conv_bwd: three af_gemm launches + af_dora_combine forward, the transposed GEMMs, weight
    gradients as GEMMs over tokens and column sums backward.  Gradients: x, lora_A, lora_B, lora_magnitude_vector."""

    @staticmethod
    def forward(ctx, x, lora_A, lora_B, lora_m, linear, ad, mask):
        y0 = linear.hip(x)
        xd = x if mask is None else ops.mul(x, mask)
        pa, _, pb, _ = ad._packs()
        pw = linear.packed()
        c2 = ops.gemm(xd, ops.PackedWeight(pw.wt, None, pw.N, pw.K, pw.kpad, pw.taps, pw.cin))
        t = ops.gemm(xd, pa)
        lb = ops.gemm(t, pb)
        u, v, norm = snapshot_scales(ad, linear)
        M, cout = c2.shape
        y = ops.dora_combine(y0.reshape(1, 1, M, cout), c2.reshape(1, 1, M, cout), lb.reshape(1, 1, M, cout), u, v).reshape(M, cout)
        ctx.linear, ctx.ad, ctx.saved = linear, ad, (xd, c2, lb, t, mask, u, v, norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        linear, ad = ctx.linear, ctx.ad
        xd, c2, lb, t, mask, u, v, norm = ctx.saved
        dy = dy.to(F16).contiguous()
        M, cout = dy.shape
        zeros = torch.zeros_like(u)
        _, pa_t, _, pb_t = ad._packs()
        dy4 = dy.reshape(1, 1, M, cout)
        dc2 = ops.affine_prelu(dy4, u, zeros).reshape(M, cout)
        dlb = ops.affine_prelu(dy4, v, zeros).reshape(M, cout)
        dt = ops.gemm(dlb, pb_t)
        dxd = ops.add(linear.hip_dgrad(dc2), ops.gemm(dt, pa_t))
        if mask is not None:
            dxd = ops.mul(dxd, mask)
        dx = ops.add(linear.hip_dgrad(dy), dxd)
        dB = wgrad(dlb, t).float().reshape(ad.lora_B.shape)
        dA = wgrad(dt, xd).float().reshape(ad.lora_A.shape)
        dm = (ops.colsum(dy, c2) + ad.scaling * ops.colsum(dy, lb)) / norm
        ctx.saved = None
        return dx, dA.to(ad.lora_A.dtype), dB.to(ad.lora_B.dtype), dm.to(ad.lora_magnitude_vector.dtype), None, None, None


def dora_linear(linear, ad, x, generator=None):
    """x [M, Cin] fp16 -> [M, Cout] through ``linear`` with the trainable DoRA adapter ``ad`` (dropout mask drawn here)."""
    return _DoRALinearFn.apply(x, ad.lora_A, ad.lora_B, ad.lora_magnitude_vector, linear, ad, ad.draw_mask(x.shape, x.device, generator))


class UNetAttnLoRA(nn.Module):
    """The attention DoRA adapters of one U-Net: q / k / v / out of the cross-attention of the last three output blocks (diffusers
    ``up_blocks.3.attentions.{0,1,2}``; reference ``set_up_attn_processors`` diffusers_attn_lora_capture.py:451-537), keys as in the
    reference's ``unet_lora_modules``: ``up_blocks_3_attentions_<i>_transformer_blocks_0_attn2_processor_to_<q|k|v|out>_lora``."""

    LAYERS = (("q", lambda a: a.to_q), ("k", lambda a: a.to_k), ("v", lambda a: a.to_v), ("out", lambda a: a.to_out[0]))

    def __init__(self, unet, layer_names=("q", "k", "v", "out"), rank=192, lora_scale_down=8, lora_dropout=0.1):
        super().__init__()
        n_out = len(unet.output_blocks)
        self.blocks = (n_out - 3, n_out - 2, n_out - 1)
        self.adapters = nn.ModuleDict()
        for i, bi in enumerate(self.blocks):
            attn2 = unet.output_blocks[bi][1].transformer_blocks[0].attn2
            for name, get in self.LAYERS:
                if name in layer_names:
                    self.adapters[f"up_blocks_3_attentions_{i}_transformer_blocks_0_attn2_processor_to_{name}_lora"] = DoRALinearAdapter(
                        get(attn2), rank, rank // lora_scale_down, lora_dropout)

    def active(self):
        """{output-block index: {'q' | 'k' | 'v' | 'out': adapter}} (what the capture graph consumes)."""
        out = {}
        for k, ad in self.adapters.items():
            i, name = int(k.split("_")[4]), k.split("_to_")[1].split("_")[0]
            out.setdefault(self.blocks[i], {})[name] = ad
        return out

    def peft_state_dict(self, adapter_name="default"):
        sd = {}
        for k, ad in self.adapters.items():
            i, name = int(k.split("_")[4]), k.split("_to_")[1].split("_")[0]
            target = f"up_blocks.3.attentions.{i}.transformer_blocks.0.attn2." + ("to_out.0" if name == "out" else f"to_{name}")
            sd[f"{target}.lora_A.{adapter_name}.weight"] = ad.lora_A.detach()
            sd[f"{target}.lora_B.{adapter_name}.weight"] = ad.lora_B.detach()
            sd[f"{target}.lora_magnitude_vector.{adapter_name}.weight"] = ad.lora_magnitude_vector.detach()
        return sd
