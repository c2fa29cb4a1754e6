"""Host-side mirror of the reference's ``ldm/modules/diffusionmodules/openaimodel.py``:
``UNetModel`` / ``ResBlock`` / ``TimestepEmbedSequential`` / ``Upsample`` / ``Downsample``
with the reference constructor arguments, state-dict keys and ``forward`` contracts
(openaimodel.py:73-161, 164-276, 414-952), executed by hand-written gfx950 kernels.

Execution plan of one epsilon-prediction (what differs from the reference's op-by-op torch):

* NCHW fp32 latents are converted once to NHWC fp16 (4 -> 8 channels zero-padded for 16-byte
  loads) and back once at the end; everything in between is NHWC fp16 in HBM;
* ResBlock = GroupNorm+SiLU kernel -> conv3x3 (bias + time-embedding add in the epilogue) ->
  GroupNorm+SiLU -> conv3x3 (bias + skip add in the epilogue); the 1x1 skip conv, the nearest
  upsample (openaimodel.py:117) and the ``torch.cat([h, hs.pop()], 1)`` of the decoder
  (openaimodel.py:918) are fused into the operand loaders of the kernels that consume them;
* the 22 per-ResBlock ``emb_layers`` Linear(SiLU(emb)) (openaimodel.py:219-225,265) are one
  GEMM per forward (weights concatenated at pack time), SiLU fused into the epilogue of
  ``time_embed.2``.
"""
import os
from abc import abstractmethod

import torch
import torch.nn as nn

from .... import ops
from ....ops import AF_ACT_SILU, F16
from ..attention import SpatialTransformer
from .util import (_PackCache, checkpoint, conv_nd, from_nhwc_f16, linear, normalization, timestep_embedding,
                   to_nhwc_f16, zero_module)


class TimestepBlock(nn.Module):
    @abstractmethod
    def forward(self, x, emb):
        """Apply the module to `x` given `emb` timestep embeddings."""


# AF_FUSE_SKIP=0: the ResBlock's 1x1 skip_connection as its own launch (A/B runs); default: K-concatenated into the second convolution
FUSE_SKIP = os.environ.get("AF_FUSE_SKIP", "1") != "0"


class SkipCat(tuple):
    """(h, skip): the decoder's channel concat, kept as two tensors so consumers read both
    sources directly instead of materialising torch.cat (openaimodel.py:918)."""


class EmbPack:
    """Time embedding handed to ResBlocks inside UNetModel: SiLU(emb) plus, when available, the
    batched output of every ResBlock's emb_layers projection."""

    def __init__(self, silu_emb, all_out=None):
        self.silu_emb = silu_emb  # [B, emb_ch] fp16
        self.all_out = all_out    # [B, sum(Cout)] fp16 or None


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def hip(self, x, emb, context=None, mask=None):
        n = len(self)
        for i, layer in enumerate(self):
            if isinstance(layer, ResBlock):
                # a ResBlock followed by a SpatialTransformer: the transformer's first launch is its GroupNorm on the block's output, so a split-K
                # second convolution may leave its reduce pass to it (ops.PendingReduce)
                x = layer.hip(x, emb, defer_out=i + 1 < n and isinstance(self[i + 1], SpatialTransformer))
            elif isinstance(layer, TimestepBlock):
                x = layer.hip(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer.hip(x, context, mask)
            else:
                x = layer.hip(x)
        return x

    def hip_train(self, x, emb, context=None, mask=None):
        saved = []
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x, s = layer.hip_train(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x, s = layer.hip_train(x, context, mask)
            else:
                x, s = layer.hip_train(x)
            saved.append(s)
        return x, saved

    def hip_bwd(self, saved, dy, dctx=None):
        """Reverse walk; returns (dx or (dx_h, dx_skip) for a SkipCat input, dctx)."""
        for layer, s in zip(reversed(list(self)), reversed(saved)):
            if isinstance(layer, SpatialTransformer):
                dy, dctx = layer.hip_bwd(s, dy, dctx)
            else:
                dy = layer.hip_bwd(s, dy)
        return dy, dctx

    def forward(self, x, emb, context=None, mask=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context, mask=mask)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    """Nearest x2 + optional 3x3 conv; the upsample is folded into the conv's input gather."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if not use_conv:
            raise NotImplementedError("Upsample(use_conv=False) is not on the SD-1.5 path")
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def hip(self, x):
        assert x.shape[-1] == self.channels
        return self.conv.hip(x, upsample=True)

    def hip_train(self, x):
        return self.hip(x), None

    def hip_bwd(self, saved, dy):
        return self.conv.hip_dgrad(dy, upsampled=True)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return from_nhwc_f16(self.hip(to_nhwc_f16(x)), x.dtype)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if not use_conv:
            raise NotImplementedError("Downsample(use_conv=False) is not on the SD-1.5 path")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def hip(self, x):
        assert x.shape[-1] == self.channels
        return self.op.hip(x)

    def hip_train(self, x):
        return self.hip(x), (x.shape[1], x.shape[2])

    def hip_bwd(self, saved, dy):
        return self.op.hip_dgrad(dy, in_hw=saved)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return from_nhwc_f16(self.hip(to_nhwc_f16(x)), x.dtype)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down:
            raise NotImplementedError("scale-shift norm / resblock_updown are not on the SD-1.5 path")
        if dropout != 0:
            raise NotImplementedError("dropout > 0 is not supported (SD-1.5 uses 0)")
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.updown = False
        self.in_layers = nn.Sequential(
            normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1)
        )
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)),
        )
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)
        self._emb_slice = None  # (offset, width) into EmbPack.all_out, assigned by UNetModel

    def hip(self, x, emb, defer_out=False):
        """x: [B,H,W,C] fp16 or SkipCat((h, skip)); emb: EmbPack or fp16 [B, emb_ch] (raw emb).  defer_out: the caller's next launch is a
        GroupNorm on the returned tensor (TimestepEmbedSequential.hip: a SpatialTransformer follows)."""
        x1, x2 = (x[0], x[1]) if isinstance(x, SkipCat) else (x, None)
        if isinstance(emb, EmbPack) and emb.all_out is not None and self._emb_slice is not None:
            off, width = self._emb_slice
            e = emb.all_out[:, off:off + width]          # [B, Cout] view, row stride = sum(Cout)
        else:
            se = emb.silu_emb if isinstance(emb, EmbPack) else ops.silu(emb.to(F16).contiguous())
            e = self.emb_layers[1].hip(se)
        h = self.in_layers[0].hip(x1, silu=True, x2=x2)
        # both convolutions feed a GroupNorm(32) (this block's second norm; the next block's first norm, a transformer's norm or the output
        # norm): they leave the partial statistics of what they store (Conv2d.hip gn_groups)
        # ... and when the launch is split-K, the GroupNorm right behind it absorbs the reduce pass (defer_gn: ops.PendingReduce)
        h = self.in_layers[2].hip(h, rowbias=e, gn_groups=32, defer_gn=True)
        h = self.out_layers[0].hip(h, silu=True)
        if isinstance(self.skip_connection, nn.Identity):
            skip = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
        elif self._skip_fusable(x1, x2):
            return ops.conv3x3(h, self._packed_conv2_skip(), skip=(x1, x2), gn_cpg=self.out_channels // 32, defer_gn=defer_out)
        else:
            skip = self.skip_connection.hip(x1, x2=x2)
        return self.out_layers[3].hip(h, residual=skip, gn_groups=32, defer_gn=defer_out)

    # ---- out_layers convolution + channel-changing 1x1 skip_connection as ONE K-concatenated implicit GEMM (ops.pack_conv3x3_skip): the block's
    # `skip_connection(x) + h` (openaimodel.py:256-276) costs no launch, no [B,H,W,Cout] round trip and no residual read of its own
    def _skip_fusable(self, x1, x2) -> bool:
        sc = self.skip_connection
        return (FUSE_SKIP and isinstance(sc, nn.Conv2d) and sc.kernel_size == (1, 1) and self.out_channels % 64 == 0 and x1.shape[-1] % 64 == 0
                and (x2 is None or x2.shape[-1] % 64 == 0))

    def _packed_conv2_skip(self) -> ops.PackedWeight:
        c2, sc = self.out_layers[3], self.skip_connection
        if not hasattr(self, "_cache_c2s"):
            self._cache_c2s = _PackCache()
        return self._cache_c2s.get((c2.weight, c2.bias, sc.weight, sc.bias),
                                   lambda: ops.pack_conv3x3_skip(c2.weight, c2.bias, sc.weight, sc.bias, c2.weight.device))

    def _emb_out(self, emb):
        if isinstance(emb, EmbPack) and emb.all_out is not None and self._emb_slice is not None:
            off, width = self._emb_slice
            return emb.all_out[:, off:off + width]
        se = emb.silu_emb if isinstance(emb, EmbPack) else ops.silu(emb.to(F16).contiguous())
        return self.emb_layers[1].hip(se)

    _lora = None      # {"conv1" | "conv2" | "conv_shortcut": DoRAConvAdapter}: set by _UNetFunction around hip_train (modules/dora.py)

    def hip_train(self, x, emb):
        """hip() keeping what the activation-gradient backward needs: the block input(s), the conv1
        output and the two GroupNorm statistics (GroupNorm outputs are only needed for weight gradients,
        which do not exist here: base weights are frozen, ddpm.py:4131-4132).  With trainable DoRA adapters attached
        (`_lora`) the three convolutions run through modules/dora.py, which keeps its own operands."""
        from ..dora import dora_conv_fwd
        lora = self._lora or {}
        x1, x2 = (x[0], x[1]) if isinstance(x, SkipCat) else (x, None)
        a, st1 = self.in_layers[0].hip_train(x1, silu=True, x2=x2)
        s1 = s2 = ssc = None
        if "conv1" in lora:
            ad = lora["conv1"]
            h1, s1 = dora_conv_fwd(self.in_layers[2], ad, a, rowbias=self._emb_out(emb), mask=ad.draw_mask(a.shape, a.device))
        else:
            h1 = self.in_layers[2].hip(a, rowbias=self._emb_out(emb), gn_groups=32, defer_gn=True)    # the GroupNorm below is its next launch
        b, st2 = self.out_layers[0].hip_train(h1, silu=True)
        fuse = "conv2" not in lora and "conv_shortcut" not in lora and not isinstance(self.skip_connection, nn.Identity) and self._skip_fusable(x1, x2)
        if fuse:
            skip = None                               # the shortcut runs inside the second convolution (see hip())
        elif isinstance(self.skip_connection, nn.Identity):
            skip = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
        elif "conv_shortcut" in lora:
            ad = lora["conv_shortcut"]
            cin = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
            skip, ssc = dora_conv_fwd(self.skip_connection, ad, x1, x2=x2, mask=ad.draw_mask(x1.shape[:-1] + (cin,), x1.device))
        else:
            skip = self.skip_connection.hip(x1, x2=x2)
        if fuse:
            out = ops.conv3x3(b, self._packed_conv2_skip(), skip=(x1, x2), gn_cpg=self.out_channels // 32)
        elif "conv2" in lora:
            ad = lora["conv2"]
            out, s2 = dora_conv_fwd(self.out_layers[3], ad, b, residual=skip, mask=ad.draw_mask(b.shape, b.device))
        else:
            out = self.out_layers[3].hip(b, residual=skip, gn_groups=32)
        return out, (x1, x2, st1, h1, st2, (lora, s1, s2, ssc, {}) if lora else None)

    def hip_bwd(self, saved, dy):
        """Reverse of openaimodel.py:256-276 (no gradient flows into the time embedding: it depends on t only).  Adapter
        parameter gradients are left in the dict at the end of `saved`."""
        from ..dora import dora_conv_bwd
        x1, x2, st1, h1, st2, ls = saved
        lora, s1, s2, ssc, grads = ls if ls is not None else ({}, None, None, None, None)
        if s2 is not None:
            db, grads["conv2"] = dora_conv_bwd(self.out_layers[3], lora["conv2"], s2, dy)
        else:
            db = self.out_layers[3].hip_dgrad(dy)
        dh1 = self.out_layers[0].hip_bwd(h1, st2, db, silu=True)
        if s1 is not None:
            da, grads["conv1"] = dora_conv_bwd(self.in_layers[2], lora["conv1"], s1, dh1)
        else:
            da = self.in_layers[2].hip_dgrad(dh1)
        if isinstance(self.skip_connection, nn.Identity):
            dskip = dy
        elif ssc is not None:
            dskip, grads["conv_shortcut"] = dora_conv_bwd(self.skip_connection, lora["conv_shortcut"], ssc, dy)
        else:
            dskip = self.skip_connection.hip_dgrad(dy)
        return self.in_layers[0].hip_bwd(x1, st1, da, silu=True, x2=x2, add=dskip)

    def forward(self, x, emb):
        return checkpoint(self._forward, (x, emb), self.parameters(), self.use_checkpoint)

    def _forward(self, x, emb):
        return from_nhwc_f16(self.hip(to_nhwc_f16(x), emb), x.dtype)


class UNetModel(nn.Module):
    """The SD-1.5 U-Net with the reference's constructor signature (openaimodel.py:444-469).
    Only the options SD-1.5 uses are implemented; the others raise at construction."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("only the use_spatial_transformer=True / context_dim path (SD-1.x) is implemented")
        if num_classes is not None or n_embed is not None or resblock_updown or use_scale_shift_norm or dims != 2:
            raise NotImplementedError("class-conditional / codebook / resblock_updown / scale-shift variants are off the hot path")
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)[0]
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1 and num_head_channels == -1:
            raise ValueError("Either num_heads or num_head_channels has to be set")

        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = False
        self.debug_attn = False
        self.backup_vars = {"save_cross_attn_vars": False}

        def heads_of(ch, nh):
            if num_head_channels == -1:
                return nh, ch // nh
            return ch // num_head_channels, num_head_channels

        def transformer(ch, nh):
            n, d = heads_of(ch, nh)
            return SpatialTransformer(ch, n, d, depth=transformer_depth, context_dim=context_dim)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), nn.SiLU(), linear(time_embed_dim, time_embed_dim))

        conv_in = conv_nd(dims, in_channels, model_channels, 3, padding=1)
        conv_in.cin_pad = ops.round_up(in_channels, 8)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_in)])
        input_block_chans = [model_channels]
        ch = model_channels
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(transformer(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint),
            transformer(ch, num_heads),
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint),
        )

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, time_embed_dim, dropout, out_channels=model_channels * mult, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(transformer(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = nn.Sequential(
            normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1))
        )

        # batched emb_layers projection: slices assigned in module order
        self._resblocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        off = 0
        for rb in self._resblocks:
            rb._emb_slice = (off, rb.out_channels)
            off += ops.round_up(rb.out_channels, 8)
        self._emb_total = off
        self._emb_cache = _PackCache()

    # ------------------------------------------------------------------ reference flag plumbing
    ALL_CA_LAYER_INDICES = [1, 2, 4, 5, 7, 8, 12, 16, 17, 18, 19, 20, 21, 22, 23, 24]  # openaimodel.py:721

    def _layer_blocks(self):
        return list(self.input_blocks) + [self.middle_block] + list(self.output_blocks)

    def set_cross_attn_flags(self, ca_flag_dict=None, ca_layer_indices=None, trans_flag_dict=None, trans_layer_indices=None):
        """Set attributes on attn2 (ca_flag_dict) / the transformer block (trans_flag_dict) of the
        selected cross-attention layers; returns the previous values (openaimodel.py:716-817).
        The ':layerwise' / ':layerwise-dict' value forms of the reference are accepted."""
        if ca_flag_dict is None and trans_flag_dict is None:
            return None, None
        l2ca = {li: i for i, li in enumerate(self.ALL_CA_LAYER_INDICES)}
        blocks = self._layer_blocks()

        def apply(flag_dict, indices, target):
            if flag_dict is None or len(indices) == 0:
                return None
            old = {}
            for k, v in flag_dict.items():
                old[k] = self.backup_vars.get(k)
                self.backup_vars[k] = v
                key, lw_arr, lw_dict = k, False, False
                if key.endswith(":layerwise"):
                    key, lw_arr = key[: -len(":layerwise")], v is not None
                if key.endswith(":layerwise-dict"):
                    key, lw_dict = key[: -len(":layerwise-dict")], v is not None
                for li in indices:
                    if li >= len(blocks) or li not in l2ca:
                        continue
                    blk = blocks[li][1].transformer_blocks[0]
                    if lw_arr:
                        v2 = v[l2ca[li]]
                    elif lw_dict:
                        v2 = v.get(li, None) if hasattr(v, "get") else v
                    else:
                        v2 = v
                    (blk.attn2 if target == "ca" else blk).__dict__[key] = v2
            return old

        ca_idx = self.ALL_CA_LAYER_INDICES if ca_layer_indices is None else ca_layer_indices
        tr_idx = self.ALL_CA_LAYER_INDICES if trans_layer_indices is None else trans_layer_indices
        return apply(ca_flag_dict, ca_idx, "ca"), apply(trans_flag_dict, tr_idx, "trans")

    # ------------------------------------------------------------------ packing
    def _packed_emb_all(self):
        params = []
        for rb in self._resblocks:
            params += [rb.emb_layers[1].weight, rb.emb_layers[1].bias]

        def build():
            dev = params[0].device
            K = self._resblocks[0].emb_channels
            w = torch.zeros((self._emb_total, K), dtype=torch.float32, device=dev)
            b = torch.zeros((self._emb_total,), dtype=torch.float32, device=dev)
            for rb in self._resblocks:
                off, width = rb._emb_slice
                w[off:off + width] = rb.emb_layers[1].weight.detach().float()
                b[off:off + width] = rb.emb_layers[1].bias.detach().float()
            return ops.pack_matrix(w, b, dev)

        return self._emb_cache.get(params, build)

    def prepare(self):
        """Pack every weight for the current device now (otherwise done lazily on first use)."""
        for m in self.modules():
            if hasattr(m, "packed"):
                m.packed()
            if hasattr(m, "_packed_qkv") and m.to_k.in_features == m.to_q.in_features:
                m._packed_qkv()
            if hasattr(m, "_packed_kv") and m.to_k.in_features != m.to_q.in_features:
                m._packed_kv()
        self._packed_emb_all()
        return self

    # ------------------------------------------------------------------ forward
    def hip(self, x_nhwc, timesteps, context, img_mask=None, capture_layers=()):
        """x_nhwc [B,H,W,8] fp16 (4 latent channels + zero pad), timesteps int64 [B],
        context [B,L,ctx] fp16 -> (eps [B,H,W,out_channels] fp16, captured activations)."""
        emb = self._embed(timesteps)   # SiLU(emb) (its only consumers are the emb_layers) + all 22 projections
        kv_layers = self._project_context_all(context)
        try:
            return self._hip_blocks(x_nhwc, emb, context, img_mask, capture_layers)
        finally:
            for m in kv_layers:
                m._kv_pre = None

    def _cross_attn_layers(self):
        out = []
        for blocks in (self.input_blocks, [self.middle_block], self.output_blocks):
            for module in blocks:
                for layer in module:
                    if isinstance(layer, SpatialTransformer):
                        out += [tb.attn2 for tb in layer.transformer_blocks]
        return out

    def _project_context_all(self, context):
        """The K and V^T projections of the context for ALL cross-attention layers in ONE GEMM (they depend on the context only):
        weights stacked [all to_k | all to_v] -> k_all [B*L, sum_C] and, through the transposing epilogue, vt_all [B, sum_C, ld];
        every layer then reads its column / row slice in place (row stride sum_C, vt batch stride sum_C * ld).  16 launches of
        ~10 us with M = 616 rows become one."""
        layers = self._cross_attn_layers()
        if not layers or context is None:
            return []
        ws = [m.to_k.weight for m in layers] + [m.to_v.weight for m in layers]
        if not hasattr(self, "_kv_all_cache"):
            self._kv_all_cache = _PackCache()
        pack = self._kv_all_cache.get(ws, lambda: ops.pack_matrix(torch.cat([w.detach() for w in ws], 0), None, ws[0].device))
        B, L, cc = context.shape
        sc = pack.N // 2
        k_all, vt_all = ops.gemm(context.reshape(B * L, cc), pack, rows_per_batch=L, split_col=sc)
        off = 0
        for m in layers:
            c = m.inner_dim
            m._kv_pre = (k_all[:, off:off + c], vt_all[:, off:off + c, :], sc)
            off += c
        return layers

    def _hip_blocks(self, x_nhwc, emb, context, img_mask, capture_layers):
        acts = {}
        hs = []
        h = x_nhwc
        layer_idx = 0

        def collect(module, h):
            attn2 = module[1].transformer_blocks[0].attn2
            a = attn2.cached_activations
            a["outfeat"] = from_nhwc_f16(h, torch.float32)
            acts[layer_idx] = a
            attn2.cached_activations = None

        for module in self.input_blocks:
            h = module.hip(h, emb, context, img_mask)
            hs.append(h)
            if layer_idx in capture_layers:
                collect(module, h)
            layer_idx += 1
        h = self.middle_block.hip(h, emb, context, img_mask)
        if layer_idx in capture_layers:
            collect(self.middle_block, h)
        layer_idx += 1
        for module in self.output_blocks:
            h = module.hip(SkipCat((h, hs.pop())), emb, context, img_mask)
            if layer_idx in capture_layers:
                collect(module, h)
            layer_idx += 1
        h = self.out[0].hip(h, silu=True)
        return self.out[2].hip(h), acts

    def hip_tail(self, trunk, timesteps, context, capture_layers, n_tail=3):
        """The inference walk of the last ``n_tail`` decoder blocks + the output head on trunk = (h, skips) as ``hip_trunk`` returns them
        (a step's shared gradient-free trunk, LatentDiffusion.guided_denoise): -> (eps NHWC, captured activations) exactly as ``_hip_blocks``
        leaves them for those layers."""
        h, skips = trunk
        emb = self._embed(timesteps)
        kv_layers = self._project_context_all(context)
        acts = {}
        try:
            n_out = len(self.output_blocks)
            layer_idx = len(self.input_blocks) + 1 + n_out - n_tail
            for k, module in enumerate(list(self.output_blocks)[n_out - n_tail:]):
                h = module.hip(SkipCat((h, skips[k])), emb, context, None)
                if layer_idx in capture_layers:
                    attn2 = module[1].transformer_blocks[0].attn2
                    a = attn2.cached_activations
                    a["outfeat"] = from_nhwc_f16(h, torch.float32)
                    acts[layer_idx] = a
                    attn2.cached_activations = None
                layer_idx += 1
            h = self.out[0].hip(h, silu=True)
            return self.out[2].hip(h), acts
        finally:
            for m in kv_layers:
                m._kv_pre = None

    def _embed(self, timesteps):
        t_emb = timestep_embedding(timesteps, self.model_channels)
        e0 = ops.gemm(t_emb, self.time_embed[0].packed(), act=AF_ACT_SILU)
        semb = ops.gemm(e0, self.time_embed[2].packed(), act=AF_ACT_SILU)
        return EmbPack(semb, ops.gemm(semb, self._packed_emb_all()))

    def hip_train(self, x_nhwc, timesteps, context, img_mask=None):
        """Forward that keeps the activations the backward needs.  -> (eps [B,H,W,out_channels], saved)."""
        emb = self._embed(timesteps)
        saved_in, saved_out, hs = [], [], []
        h = x_nhwc
        for module in self.input_blocks:
            h, s = module.hip_train(h, emb, context, img_mask)
            saved_in.append(s)
            hs.append(h)
        h, saved_mid = self.middle_block.hip_train(h, emb, context, img_mask)
        for module in self.output_blocks:
            h, s = module.hip_train(SkipCat((h, hs.pop())), emb, context, img_mask)
            saved_out.append(s)
        g, st = self.out[0].hip_train(h, silu=True)
        return self.out[2].hip(g), (saved_in, saved_mid, saved_out, h, st, context.shape)

    def hip_train_trunk(self, x_nhwc, emb, context, img_mask, n_tail):
        """hip_train without the last ``n_tail`` decoder blocks and the output head (modules/diffusionmodules/capture_graph.py):
        -> (h, [the skips those blocks will pop, in pop order], saved)."""
        saved_in, saved_out, hs = [], [], []
        h = x_nhwc
        for module in self.input_blocks:
            h, s = module.hip_train(h, emb, context, img_mask)
            saved_in.append(s)
            hs.append(h)
        h, saved_mid = self.middle_block.hip_train(h, emb, context, img_mask)
        for module in list(self.output_blocks)[:len(self.output_blocks) - n_tail]:
            h, s = module.hip_train(SkipCat((h, hs.pop())), emb, context, img_mask)
            saved_out.append(s)
        skips = [hs.pop() for _ in range(n_tail)]
        assert not hs
        return h, skips, (saved_in, saved_mid, saved_out, context.shape)

    def hip_trunk(self, x_nhwc, emb, context, img_mask, n_tail):
        """``hip_train_trunk`` for a pass that needs NO gradient below the tail (a no-grad instance of a compositional step, or one whose
        input and context carry none): the inference walk -- every cross-attention layer's k / v from one GEMM, the fused C = 320 blocks,
        nothing saved -- up to the last ``n_tail`` decoder blocks.  -> (h, [the skips those blocks pop, in pop order])."""
        kv_layers = self._project_context_all(context)
        try:
            hs = []
            h = x_nhwc
            for module in self.input_blocks:
                h = module.hip(h, emb, context, img_mask)
                hs.append(h)
            h = self.middle_block.hip(h, emb, context, img_mask)
            for module in list(self.output_blocks)[:len(self.output_blocks) - n_tail]:
                h = module.hip(SkipCat((h, hs.pop())), emb, context, img_mask)
            skips = [hs.pop() for _ in range(n_tail)]
            assert not hs
            return h, skips
        finally:
            for m in kv_layers:
                m._kv_pre = None

    def hip_bwd_trunk(self, saved, dh, dskips_tail, need_dx=True, res_gradscale=1.0):
        """Backward of hip_train_trunk: dh = gradient of h, dskips_tail[i] = gradient of the i-th popped skip (already scaled by the
        tail where the live path scales it) or None.  Same walk as hip_bwd."""
        saved_in, saved_mid, saved_out, ctx_shape = saved
        n_in, n_out, n_tail = len(self.input_blocks), len(self.output_blocks), len(dskips_tail)
        d, dctx = dh, None
        dskips = {}
        for k, g in enumerate(dskips_tail):                      # the k-th tail block popped hs[n_tail - 1 - k]
            dskips[n_tail - 1 - k] = (g, False)
        blocks = list(self.output_blocks)[:n_out - n_tail]
        for oi in range(len(blocks) - 1, -1, -1):
            (d, dskip), dctx = blocks[oi].hip_bwd(saved_out[oi], d, dctx)
            dskips[n_in - 1 - oi] = (dskip, res_gradscale != 1.0 and oi >= self.num_res_blocks + 1)
        d, dctx = self.middle_block.hip_bwd(saved_mid, d, dctx)
        for i in range(n_in - 1, -1, -1):
            g, scaled = dskips[i]
            if g is not None:
                d = ops.axpy(d, g, res_gradscale) if scaled else ops.add(d, g)
            if i == 0 and not need_dx:
                return None, dctx.reshape(ctx_shape)
            d, dctx = self.input_blocks[i].hip_bwd(saved_in[i], d, dctx)
        return d, dctx.reshape(ctx_shape)

    def hip_bwd(self, saved, deps_nhwc, need_dx=True, res_gradscale=1.0):
        """Activation-gradient backward: deps [B,H,W,roundup(out_channels,8)] fp16 (zero padded) ->
        (dx [B,H,W,in_channels] or None, dcontext [B,L,ctx] fp16).  Skip-connection gradients from the
        decoder are added to the encoder's output gradients as the walk reaches them; `res_gradscale` multiplies the
        gradient of the skips consumed by output blocks >= num_res_blocks + 1 (the live path's
        res_hidden_states_gradscale on diffusers up_blocks[1:], diffusers_attn_lora_capture.py:382-396, 606-609)."""
        saved_in, saved_mid, saved_out, h_last, st, ctx_shape = saved
        d = self.out[0].hip_bwd(h_last, st, self.out[2].hip_dgrad(deps_nhwc), silu=True)
        dctx = None
        dskips = []
        for module, s in zip(reversed(self.output_blocks), reversed(saved_out)):
            (d, dskip), dctx = module.hip_bwd(s, d, dctx)
            dskips.append(dskip)   # the LAST decoder block consumed hs[0], so this appends d(hs[0]), d(hs[1]), ...
        d, dctx = self.middle_block.hip_bwd(saved_mid, d, dctx)
        n_in = len(self.input_blocks)
        n_out = len(self.output_blocks)
        for i in range(n_in - 1, -1, -1):
            scaled = res_gradscale != 1.0 and (n_out - 1 - i) >= self.num_res_blocks + 1     # hs[i] fed output block n_out-1-i
            d = ops.axpy(d, dskips[i], res_gradscale) if scaled else ops.add(d, dskips[i])
            if i == 0 and not need_dx:
                return None, dctx.reshape(ctx_shape)
            d, dctx = self.input_blocks[i].hip_bwd(saved_in[i], d, dctx)
        return d, dctx.reshape(ctx_shape)

    def forward(self, x, timesteps=None, context=None, y=None, context_in=None, extra_info=None, **kwargs):
        """Reference contract (openaimodel.py:820-952): x [N,4,H,W], timesteps [N], context
        [N,L,ctx]; ``extra_info`` carries ``img_mask`` / ``capture_ca_activations`` in and
        receives ``ca_layers_activations`` out.  Returns eps [N,out_channels,H,W] in x.dtype."""
        assert y is None, "must specify y if and only if the model is class-conditional"
        capture = extra_info.get("capture_ca_activations", False) if extra_info is not None else False
        img_mask = extra_info.get("img_mask", None) if extra_info is not None else None
        captured = [22, 23, 24] if capture else []
        old_flags = None
        if capture:
            old_flags, _ = self.set_cross_attn_flags(ca_flag_dict={"save_cross_attn_vars": True}, ca_layer_indices=captured)
        ei = extra_info or {}
        rewrites = bool(ei.get("normalize_cross_attn", False) or ei.get("mix_attn_mats_in_batch", False))
        trainable = torch.is_grad_enabled() and (x.requires_grad or context.requires_grad or ei.get("_ffn_lora_adapters")
                                                 or ei.get("_attn_lora_adapters"))
        if rewrites or ei.get("_attn_lora_adapters") or (capture and trainable):
            # Stage-2 pass: explicit attention in the last three cross-attention layers (score rewrites, captures with gradients,
            # attention LoRAs) as a chain of autograd nodes -- modules/diffusionmodules/capture_graph.py
            from .capture_graph import unet_forward_captured
            if extra_info is None:
                raise ValueError("extra_info must be a dict on the capture / score-rewrite path")
            if capture:
                self.set_cross_attn_flags(ca_flag_dict=old_flags, ca_layer_indices=captured)     # this path does its own capturing
            eps = unet_forward_captured(self, x, timesteps, context, extra_info)
            if not capture:
                extra_info.pop("ca_layers_activations", None)
            return eps
        if trainable:
            gs = float((extra_info or {}).get("res_hidden_states_gradscale", 1) or 1)
            lora = (extra_info or {}).get("_ffn_lora_adapters")           # set by UNetWrapper when use_ffn_lora is on
            return _UNetFunction.apply(self, x, timesteps, context, img_mask, gs, lora, *[t[3] for t in lora_param_order(lora)])
        try:
            ctx = context.to(F16).contiguous()
            shared = None
            if img_mask is None and ei.get("_trunk_cache") is not None:
                from .capture_graph import cached_trunk
                shared = cached_trunk(ei, x.shape[0])
            if shared is not None:
                eps, acts = self.hip_tail(shared, timesteps, ctx, captured)      # the step's shared gradient-free trunk holds these rows
            else:
                xh = to_nhwc_f16(x, ops.round_up(self.in_channels, 8))
                eps, acts = self.hip(xh, timesteps, ctx, img_mask, captured)
        finally:
            if capture:
                self.set_cross_attn_flags(ca_flag_dict=old_flags, ca_layer_indices=captured)
        if extra_info is not None:
            extra_info["ca_layers_activations"] = {
                key: {li: acts[li][key] for li in acts} for key in ("outfeat", "attn", "attnscore", "q", "q2", "k", "v", "attn_out")
            }
        return from_nhwc_f16(eps, x.dtype, self.out_channels)


class _UNetFunction(torch.autograd.Function):
    """One autograd node for the whole U-Net: forward = UNetModel.hip_train, backward = the manual
    activation-gradient walk UNetModel.hip_bwd (gradients w.r.t. x and context; the base weights are
    frozen as in the reference, ddpm.py:4131-4132, so no weight gradients exist on this path)."""

    @staticmethod
    def forward(ctx, unet, x, timesteps, context, img_mask, res_gradscale=1.0, lora=None, *lora_params):
        """lora: {output-block index: {"conv1" | "conv2" | "conv_shortcut": DoRAConvAdapter}} or None; lora_params: the adapters'
        parameters in `lora_param_order(lora)` order (passed so that autograd routes their gradients).
        With ``unet.train_graphs`` (graphs.GraphedSegment pair, set by the trainer) the forward and the backward walk are each replayed
        as ONE hipGraph per input signature instead of ~450 / ~700 launches from Python."""
        ctx.res_gradscale = res_gradscale
        ctx.lora = lora
        ctx.need_dx, ctx.x_dtype, ctx.c_dtype = x.requires_grad, x.dtype, context.dtype

        def body(xs, ts, cs, ms):
            xh = to_nhwc_f16(xs, ops.round_up(unet.in_channels, 8))
            try:
                for bi, ads in (lora or {}).items():
                    unet.output_blocks[bi][0]._lora = ads
                eps, saved = unet.hip_train(xh, ts, cs.to(F16).contiguous(), ms)
            finally:
                for bi in (lora or {}):
                    unet.output_blocks[bi][0]._lora = None
            return from_nhwc_f16(eps, xs.dtype, unet.out_channels), saved

        graphs = getattr(unet, "train_graphs", None)
        if graphs is not None and x.is_cuda:
            ctx.key = (tuple(x.shape), tuple(context.shape), str(x.dtype), str(context.dtype), img_mask is not None, float(res_gradscale),
                       tuple(id(a) for _, _, _, a in lora_param_order(lora)), ctx.need_dx)
            # a second forward of the same signature before this node's backward (batch_student_steps=False: one student call per
            # denoising step) must not replay into the buffers this node still needs: GraphedSegment.busy -> that call runs eagerly
            from_graph = not graphs[0].busy(ctx.key) and graphs[0].entries.get(ctx.key, {}).get("state") in ("warm", "graph")
            out, ctx.saved = graphs[0].run(ctx.key, body, [x.detach(), timesteps, context.detach(), img_mask],
                                           refresh=lambda: _refresh_adapter_packs(lora))
            if not from_graph:
                ctx.key = None                # eager forward (first call of a signature, or the graph's buffers were held): its saved
                                              # activations are ordinary tensors, so its backward must not be captured / replayed either
            elif any(ctx.needs_input_grad):
                graphs[0].claim(ctx.key, ctx)
        else:
            ctx.key = None
            out, ctx.saved = body(x.detach(), timesteps, context.detach(), img_mask)
        ctx.unet = unet
        return out

    @staticmethod
    def backward(ctx, deps):
        unet = ctx.unet
        lora = ctx.lora
        saved = ctx.saved

        def body(de):
            # The backward is linear in d(eps), and loss gradients are tiny (MSE over ~16k pixels): scale d(eps) by a
            # power of two so that its largest entry is ~256 before the fp16 cast, unscale the results in fp32.  The
            # scale is computed and applied on the device (no host sync); a power of two makes it exact.
            amax = de.abs().amax().float().clamp_min(1e-30)
            scale = torch.exp2(torch.floor(torch.log2(256.0 / amax)))
            dh = to_nhwc_f16((de * scale).contiguous(), ops.round_up(unet.out_channels, 8))
            dx, dctx = unet.hip_bwd(saved, dh, need_dx=ctx.need_dx, res_gradscale=ctx.res_gradscale)
            lora_grads = []
            for bi, key, pname, p in lora_param_order(lora):
                g = saved[2][bi][0][5][4][key][pname]             # ResBlock saved -> (lora, s1, s2, ssc, grads)
                lora_grads.append((g / scale).to(p.dtype))
            gx = (from_nhwc_f16(dx, torch.float32, unet.in_channels) / scale).to(ctx.x_dtype) if dx is not None else None
            return (gx, (dctx.float() / scale).to(ctx.c_dtype)) + tuple(lora_grads)

        graphs = getattr(unet, "train_graphs", None)
        if graphs is not None and ctx.key is not None and graphs[0].entries.get(ctx.key, {}).get("state") == "graph":
            # only once the forward of this signature is itself a graph: its saved activations then sit at fixed addresses
            res = graphs[1].run(ctx.key, body, [deps.detach()])
        else:
            res = body(deps.detach())
        ctx.saved = None
        ctx.af_holds_replay = False                               # the replayed forward's buffers are free again
        return (None, res[0], None, res[1], None, None, None) + tuple(res[2:])


def _refresh_adapter_packs(lora):
    """Before a graph replay: re-derive the adapters' fp16 weight packs IN PLACE if the optimizer moved their parameters (the captured
    kernels read the packs at fixed addresses; modules/dora.py::DoRAConvAdapter._packs copies into the existing buffers)."""
    for bi in (lora or {}):
        for ad in lora[bi].values():
            ad._packs(refresh_scales=True)


def lora_param_order(lora):
    """Deterministic flattening of {block: {conv key: adapter}} -> [(block, key, parameter name, parameter)]."""
    out = []
    for bi in sorted(lora or {}):
        for key in sorted(lora[bi]):
            for pname in ("lora_A", "lora_B", "lora_magnitude_vector"):
                out.append((bi, key, pname, getattr(lora[bi][key], pname)))
    return out


def unet_param_shapes(cfg):
    """(name, shape) of every parameter of UNetModel(**cfg), without allocating them."""
    with torch.device("meta"):
        m = UNetModel(**cfg)
    return [(n, tuple(p.shape)) for n, p in m.named_parameters()]
