"""Host-side mirror of the decoder half of the reference's ``ldm/modules/diffusionmodules/model.py`` -- the SD-1.5 KL-f8 VAE decoder
that turns the denoised latent into the image (BASELINE configs[1] ends with it): ``Normalize`` (:39-40),
``Upsample`` (:43-58), ``ResnetBlock`` (:83-142), ``AttnBlock`` (:151-243, the mask-free branch), ``Decoder`` (:502-608), plus the
``post_quant_conv`` + decoder wrapper of ``ldm/models/autoencoder.py:29,56-59``.  Same module tree and parameter names
(``decoder.conv_in``, ``decoder.mid.block_1.norm1``, ``decoder.mid.attn_1.q``, ``decoder.up.3.block.0.conv1``,
``decoder.up.1.upsample.conv``, ``decoder.norm_out``, ``decoder.conv_out``, ``post_quant_conv``) so ``first_stage_model.*`` checkpoint
keys load with ``load_state_dict``.

Execution is NHWC fp16 through the C ABI, with the kernels the U-Net already uses: GroupNorm(32, eps 1e-6)+swish fused, 3x3 convs as
implicit GEMM (the nearest x2 upsample folded into the following conv's gather, the residual / 1x1 ``nin_shortcut`` in the epilogue), and
the one attention layer (single head, 512 channels, 4096 tokens at 512x512) as  q.k^T GEMM -> ``af_softmax_rows`` -> P.v GEMM per
image: its head dim is beyond the flash kernel's register budget and the layer is ~2 % of the decoder.

The weights are frozen, but the ArcFace alignment terms differentiate through the decoder into the latent
(``decode_first_stage_with_grad``, ddpm.py:899-908 -> ``calc_arcface_align_loss``, ddpm.py:2511-2535), so ``decode`` of a latent that
requires grad is a ``torch.autograd`` node (``VAEDecodeFn``) with the INPUT gradient only: every layer has ``hip_train`` (forward that
keeps the GroupNorm inputs + statistics) and ``hip_bwd`` (``af_groupnorm_bwd``, the flipped-weight dgrad convolutions -- a folded nearest
x2 upsample becomes ``af_sumpool2x2`` of the dgrad --, and for the attention layer the explicit P recomputed, ``af_softmax_rows_bwd`` and
four GEMMs per image).  No parameter gradient is formed."""
import torch
import torch.nn as nn

from .... import ops
from ....ops import F16
from .util import Conv2d, GroupNorm32, _require_cuda, from_nhwc_f16, to_nhwc_f16


def Normalize(in_channels, num_groups=32):
    return GroupNorm32(num_groups, in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Upsample without conv is not used by the SD VAE")
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def hip(self, x):
        return self.conv.hip(x, upsample=True)

    def hip_train(self, x):
        return self.hip(x), None

    def hip_bwd(self, saved, dy):
        return self.conv.hip_dgrad(dy, upsampled=True)


class Downsample(nn.Module):
    """3x3 stride-2 conv on the input padded (0, 1, 0, 1) (model.py:61-80): the asymmetric padding is a +1 shift of the taps in the
    implicit-GEMM loader (``tap_shift``), not a padded copy."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Downsample without conv (avg_pool) is not used by the SD VAE")
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)
        self._key, self._pack = None, None

    def hip(self, x):
        c = self.conv
        key = (c.weight._version, c.bias._version, c.weight.data_ptr())
        if key != self._key:
            self._pack, self._key = ops.pack_conv3x3(c.weight, c.bias, c.weight.device), key
        return ops.conv3x3(x, self._pack, stride=2, tap_shift=1)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        if temb_channels > 0 or dropout != 0:
            raise NotImplementedError("the VAE's ResnetBlocks have no time embedding and no dropout")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def hip(self, x):
        h = self.conv1.hip(self.norm1.hip(x, silu=True))
        h = self.norm2.hip(h, silu=True)
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut.hip(x) if self.use_conv_shortcut else self.nin_shortcut.hip(x)
        return self.conv2.hip(h, residual=x)

    def hip_train(self, x):
        n1, st1 = self.norm1.hip_train(x, silu=True)
        h = self.conv1.hip(n1)
        n2, st2 = self.norm2.hip_train(h, silu=True)
        xs = x
        if self.in_channels != self.out_channels:
            xs = self.conv_shortcut.hip(x) if self.use_conv_shortcut else self.nin_shortcut.hip(x)
        return self.conv2.hip(n2, residual=xs), (x, st1, h, st2)

    def hip_bwd(self, saved, dy):
        x, st1, h, st2 = saved
        dh = self.norm2.hip_bwd(h, st2, self.conv2.hip_dgrad(dy), silu=True)
        dxs = dy
        if self.in_channels != self.out_channels:
            dxs = (self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut).hip_dgrad(dy)
        return self.norm1.hip_bwd(x, st1, self.conv1.hip_dgrad(dh), silu=True, add=dxs)


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self._qs_key, self._qs = None, None

    def _q_scaled_pack(self):
        """q projection with the c^-0.5 score scale folded in (model.py:187)."""
        key = (self.q.weight._version, self.q.bias._version, self.q.weight.data_ptr())
        if key != self._qs_key:
            sc = float(self.in_channels) ** -0.5
            self._qs = ops.pack_matrix(self.q.weight.detach().float().reshape(self.in_channels, -1) * sc, self.q.bias.detach().float() * sc,
                                       self.q.weight.device)
            self._qs_key = key
        return self._qs

    @staticmethod
    def pair_classes(mask, H, W):
        """mask dict {'aug_mask', 'fg_mask'} ([B,1,h,w] or None) -> uint8 [B, H*W]: bit 0 = fg*aug != 0, bit 1 = (1-fg)*aug != 0,
        nearest-resized to the feature map (model.py:192-201); None when fg_mask is absent (the reference then ignores masks)."""
        if mask is None or mask.get("fg_mask") is None:
            return None
        import torch.nn.functional as F
        fg = F.interpolate(mask["fg_mask"].float(), size=(H, W), mode="nearest")
        aug = mask.get("aug_mask")
        aug = torch.ones_like(fg) if aug is None else F.interpolate(aug.float(), size=(H, W), mode="nearest")
        # the reference tests fg_i*fg_j != 0 or bg_i*bg_j != 0 per pair: (cls_i & cls_j) != 0 with these two bits
        cls = (fg * aug != 0).to(torch.uint8) + 2 * ((1 - fg) * aug != 0).to(torch.uint8)
        return cls.reshape(cls.shape[0], H * W).to(torch.uint8).contiguous()

    def hip(self, x, mask=None):
        B, H, W, C = x.shape
        N = H * W
        cls = self.pair_classes(mask, H, W)
        if N % 8 != 0 or N > 4096:
            raise NotImplementedError(f"VAE attention over {N} tokens (af_softmax_rows holds rows up to 4096)")
        hn = self.norm.hip(x).reshape(B * N, C)
        q = ops.gemm(hn, self._q_scaled_pack())
        k = ops.gemm(hn, self.k.packed())
        v = ops.gemm(hn, self.v.packed())
        vt = ops.transpose_tokens(v, B, N, C, C)                                  # [B, C, N]: the P.v GEMM's K-contiguous operand
        kp, cp = ops.round_up(C, 64), ops.round_up(C, 128)
        if kp != C or N % 128 != 0 or vt.shape[2] % 64 != 0 or cp != C:
            raise NotImplementedError("VAE attention operands must already be tile aligned (C % 128 == 0, tokens % 128 == 0)")
        out = torch.empty((B * N, C), dtype=F16, device=x.device)
        for b in range(B):                                                        # one image at a time: [N, N] scores = 32 MB at 64x64
            kb = ops.PackedWeight(k[b * N:(b + 1) * N], None, N, C, C, 1, C)      # keys as the "weight" [N, C], K-contiguous
            p = ops.softmax_rows(ops.gemm(q[b * N:(b + 1) * N], kb))
            if cls is not None:
                ops.mask_pairs_(p, cls[b].to(x.device))
            vb = ops.PackedWeight(vt[b], None, C, N, N, 1, N)                     # V^T [C, N]
            out[b * N:(b + 1) * N] = ops.gemm(p, vb)
        y = ops.gemm(out, self.proj_out.packed(), residual=x.reshape(B * N, C))
        return y.reshape(B, H, W, C)

    def _q_scaled_pack_bwd(self):
        key = (self.q.weight._version, self.q.weight.data_ptr())
        if key != getattr(self, "_qsb_key", None):
            sc = float(self.in_channels) ** -0.5
            self._qsb = ops.pack_matrix(self.q.weight.detach().float().reshape(self.in_channels, -1).t() * sc, None, self.q.weight.device)
            self._qsb_key = key
        return self._qsb

    def hip_train(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("the masked AttnBlock belongs to the encoder, which is never differentiated")
        B, H, W, C = x.shape
        hn, st = self.norm.hip_train(x)
        return self.hip(x), (x, st, hn.reshape(B * H * W, C))

    def hip_bwd(self, saved, dy):
        """Per image: P recomputed (q.k^T GEMM + row softmax), dV = P^T dO, dP = dO V^T, dS = softmax_bwd(P, dP), dQ = dS K, dK = dS^T Q;
        then the three projections' dgrad summed into d(hn) and the GroupNorm backward with the residual gradient added."""
        from ....autograd_ops import wgrad
        x, st, hn = saved
        B, H, W, C = x.shape
        N = H * W
        dy2 = dy.reshape(B * N, C)
        q = ops.gemm(hn, self._q_scaled_pack())
        k = ops.gemm(hn, self.k.packed())
        v = ops.gemm(hn, self.v.packed())
        kt = ops.transpose_tokens(k, B, N, C, C)                                  # [B, C, N]
        do = ops.gemm(dy2, self.proj_out.packed_bwd())
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            p = ops.softmax_rows(ops.gemm(q[sl], ops.PackedWeight(k[sl], None, N, C, C, 1, C)))
            dv[sl] = wgrad(p, do[sl]).to(F16)
            dp = ops.gemm(do[sl], ops.PackedWeight(v[sl], None, N, C, C, 1, C))
            ds = ops.softmax_rows_bwd(p, dp)
            dq[sl] = ops.gemm(ds, ops.PackedWeight(kt[b], None, C, N, N, 1, N))
            dk[sl] = wgrad(ds, q[sl]).to(F16)
        dhn = ops.gemm(dq, self._q_scaled_pack_bwd())
        dhn = ops.gemm(dk, self.k.packed_bwd(), residual=dhn)
        dhn = ops.gemm(dv, self.v.packed_bwd(), residual=dhn)
        return self.norm.hip_bwd(x, st, dhn.reshape(B, H, W, C), add=dy)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False, attn_type="vanilla",
                 **ignorekwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla" or give_pre_end or tanh_out:
            raise NotImplementedError("only the vanilla-attention SD VAE decoder is built")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.out_ch = resolution, in_channels, out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.conv_in.cin_pad = ops.round_up(z_channels, 8)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)
        self._out_key, self._out_pack = None, None

    def _conv_out_pack(self):
        """conv_out with its 3 output channels zero padded to 8 (the GEMM epilogue stores 8-byte groups)."""
        c = self.conv_out
        key = (c.weight._version, c.bias._version, c.weight.data_ptr())
        if key != self._out_key:
            n8 = ops.round_up(self.out_ch, 8)
            w = torch.zeros((n8,) + tuple(c.weight.shape[1:]), device=c.weight.device)
            b = torch.zeros((n8,), device=c.weight.device)
            w[:self.out_ch], b[:self.out_ch] = c.weight.detach().float(), c.bias.detach().float()
            self._out_pack = ops.pack_conv3x3(w, b, c.weight.device)
            self._out_key = key
        return self._out_pack

    def hip(self, z):
        """z [B, h, w, roundup(z_channels, 8)] fp16 -> [B, 8h, 8w, roundup(out_ch, 8)] fp16."""
        h = self.conv_in.hip(z)
        h = self.mid.block_1.hip(h)
        h = self.mid.attn_1.hip(h)
        h = self.mid.block_2.hip(h)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block].hip(h)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block].hip(h)
            if i_level != 0:
                h = self.up[i_level].upsample.hip(h)
        h = self.norm_out.hip(h, silu=True)
        return ops.conv3x3(h, self._conv_out_pack())

    def _layers(self):
        """The decoder as the flat layer sequence ``hip`` runs between conv_in and norm_out."""
        seq = [self.mid.block_1, self.mid.attn_1, self.mid.block_2]
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                seq.append(self.up[i_level].block[i_block])
                if len(self.up[i_level].attn) > 0:
                    seq.append(self.up[i_level].attn[i_block])
            if i_level != 0:
                seq.append(self.up[i_level].upsample)
        return seq

    def _bwd_packs(self):
        """dgrad packs of the two end convolutions, whose narrow sides (z_channels / out_ch) are zero padded to 8."""
        ci, co = self.conv_in, self.conv_out
        key = (ci.weight._version, co.weight._version, ci.weight.data_ptr(), co.weight.data_ptr())
        if key != getattr(self, "_bwd_key", None):
            dev = ci.weight.device
            wi = ci.weight.detach().float().flip(2, 3).permute(1, 0, 2, 3)           # [z_channels, block_in, 3, 3]
            z8 = ops.round_up(wi.shape[0], 8)
            wi = torch.cat([wi, torch.zeros((z8 - wi.shape[0],) + tuple(wi.shape[1:]), device=dev)]) if z8 != wi.shape[0] else wi
            wo = co.weight.detach().float().flip(2, 3).permute(1, 0, 2, 3)           # [block_out, out_ch, 3, 3]
            self._bwd = (ops.pack_conv3x3(wi.contiguous(), None, dev), ops.pack_conv3x3(wo.contiguous(), None, dev, ops.round_up(self.out_ch, 8)))
            self._bwd_key = key
        return self._bwd

    def hip_train(self, z):
        """``hip`` -> (image, what ``hip_bwd`` reads)."""
        h = self.conv_in.hip(z)
        saved = []
        for layer in self._layers():
            h, sv = layer.hip_train(h)
            saved.append(sv)
        n, st = self.norm_out.hip_train(h, silu=True)
        return ops.conv3x3(n, self._conv_out_pack()), (saved, h, st)

    def hip_bwd(self, saved, dy):
        """dy [B, 8h, 8w, roundup(out_ch, 8)] fp16 -> [B, h, w, roundup(z_channels, 8)] fp16."""
        per_layer, h, st = saved
        p_in, p_out = self._bwd_packs()
        d = self.norm_out.hip_bwd(h, st, ops.conv3x3(dy, p_out), silu=True)
        for layer, sv in reversed(list(zip(self._layers(), per_layer))):
            d = layer.hip_bwd(sv, d)
        return ops.conv3x3(d, p_in)

    def forward(self, z):
        _require_cuda(self.conv_in.weight, "VAE Decoder")
        y = self.hip(to_nhwc_f16(z, ops.round_up(z.shape[1], 8)))
        return from_nhwc_f16(y, z.dtype, self.out_ch)


class Encoder(nn.Module):
    """Reference ``Encoder`` (model.py:408-500) incl. the fg / aug attention masks of the mid block: image [B,3,H,W] -> moments [B, 2*z_channels, H/8, W/8]."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, double_z=True, use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla":
            raise NotImplementedError("only the vanilla-attention SD VAE encoder is built")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        self.conv_in.cin_pad = ops.round_up(in_channels, 8)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.out_channels = 2 * z_channels if double_z else z_channels
        self.conv_out = Conv2d(block_in, self.out_channels, kernel_size=3, stride=1, padding=1)

    def hip(self, x, mask=None):
        """x [B,H,W,roundup(in_channels,8)] fp16 -> [B,H/8,W/8,out_channels] fp16 (out_channels % 8 == 0 for double_z z=4)."""
        h = self.conv_in.hip(x)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block].hip(h)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block].hip(h, mask)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample.hip(h)
        h = self.mid.block_1.hip(h)
        h = self.mid.attn_1.hip(h, mask)
        h = self.mid.block_2.hip(h)
        return self.conv_out.hip(self.norm_out.hip(h, silu=True))

    def forward(self, x, mask=None):
        _require_cuda(self.conv_in.weight, "VAE Encoder")
        if self.out_channels % 8 != 0:
            raise NotImplementedError("encoder output channels must be a multiple of 8 (double_z with z_channels = 4)")
        y = self.hip(to_nhwc_f16(x, ops.round_up(x.shape[1], 8)), mask)
        return from_nhwc_f16(y, x.dtype, self.out_channels)


class VAEDecodeFn(torch.autograd.Function):
    """image = decoder(post_quant_conv(z)) with the gradient w.r.t. z (frozen weights).  The incoming image gradient is normalised by a
    power of two to a largest entry ~ 1 before its fp16 cast -- the backward amplifies towards the latent (three 2x2 sum-pools, GroupNorm
    rstd), so the headroom is spent upwards -- and the result is unscaled in fp32."""

    @staticmethod
    def forward(ctx, z, model):
        zh = to_nhwc_f16(z.detach(), ops.round_up(z.shape[1], 8))
        B, H, W, c8 = zh.shape
        zq = ops.gemm(zh.reshape(B * H * W, c8), model._pq_pack()).reshape(B, H, W, -1)
        y, saved = model.decoder.hip_train(zq)
        ctx.model, ctx.saved, ctx.zshape, ctx.dtype = model, saved, tuple(z.shape), z.dtype
        return from_nhwc_f16(y, z.dtype, model.decoder.out_ch)

    @staticmethod
    def backward(ctx, dy):
        model = ctx.model
        if ctx.saved is None:
            raise RuntimeError("VAEDecodeFn: the saved activations were released by the first backward (they are ~0.5 GB per 512x512 image); "
                               "a second backward through the same decode is not supported")
        amax = dy.detach().abs().amax().float().clamp_min(1e-30)
        scale = torch.exp2(torch.floor(torch.log2(1.0 / amax)))
        d = to_nhwc_f16((dy.float() * scale).contiguous(), ops.round_up(dy.shape[1], 8))
        dzq = model.decoder.hip_bwd(ctx.saved, d)
        ctx.saved = None
        B, H, W, c8 = dzq.shape
        dz = ops.gemm(dzq.reshape(B * H * W, c8), model._pq_pack_bwd()).reshape(B, H, W, -1)
        return (from_nhwc_f16(dz, torch.float32, ctx.zshape[1]) / scale).to(ctx.dtype), None


class AutoencoderKLDecoder(nn.Module):
    """``first_stage_model`` restricted to what inference needs: ``decode(z) = decoder(post_quant_conv(z))``
    (ldm/models/autoencoder.py:29, 56-59).  SD-1.5: embed_dim 4, ddconfig below."""

    SD15_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                         attn_resolutions=[], dropout=0.0)

    def __init__(self, ddconfig=None, embed_dim=4):
        super().__init__()
        ddconfig = dict(ddconfig or self.SD15_DDCONFIG)
        ddconfig.pop("double_z", None)
        self.decoder = Decoder(**ddconfig)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self._pq_key, self._pq = None, None

    def _pq_pack(self):
        """post_quant_conv on the channel-padded latent: [4 -> 4] embedded in an [8 -> 8] matrix."""
        c = self.post_quant_conv
        key = (c.weight._version, c.bias._version, c.weight.data_ptr())
        if key != self._pq_key:
            n8, k8 = ops.round_up(c.out_channels, 8), ops.round_up(c.in_channels, 8)
            w = torch.zeros((n8, k8), device=c.weight.device)
            b = torch.zeros((n8,), device=c.weight.device)
            w[:c.out_channels, :c.in_channels] = c.weight.detach().float().reshape(c.out_channels, c.in_channels)
            b[:c.out_channels] = c.bias.detach().float()
            self._pq = ops.pack_matrix(w, b, c.weight.device)
            self._pq_key = key
        return self._pq

    def _pq_pack_bwd(self):
        c = self.post_quant_conv
        key = (c.weight._version, c.weight.data_ptr())
        if key != getattr(self, "_pqb_key", None):
            n8, k8 = ops.round_up(c.out_channels, 8), ops.round_up(c.in_channels, 8)
            w = torch.zeros((k8, n8), device=c.weight.device)
            w[:c.in_channels, :c.out_channels] = c.weight.detach().float().reshape(c.out_channels, c.in_channels).t()
            self._pqb, self._pqb_key = ops.pack_matrix(w, None, c.weight.device), key
        return self._pqb

    def decode(self, z):
        """z [B, 4, h, w] (already divided by the 0.18215 scale factor) -> image [B, 3, 8h, 8w] in z.dtype, roughly [-1, 1].  A latent that
        requires grad (``decode_first_stage_with_grad``, ddpm.py:899-908) gets the autograd node with the input gradient."""
        _require_cuda(self.post_quant_conv.weight, "AutoencoderKLDecoder")
        if z.requires_grad and torch.is_grad_enabled():
            return VAEDecodeFn.apply(z, self)
        with torch.no_grad():
            zh = to_nhwc_f16(z, ops.round_up(z.shape[1], 8))
            B, H, W, c8 = zh.shape
            zq = ops.gemm(zh.reshape(B * H * W, c8), self._pq_pack()).reshape(B, H, W, -1)
            return from_nhwc_f16(self.decoder.hip(zq), z.dtype, self.decoder.out_ch)


class AutoencoderKL(AutoencoderKLDecoder):
    """``first_stage_model`` with both halves: ``encode(x)`` -> DiagonalGaussian moments (mean, logvar) through ``quant_conv``
    (ldm/models/autoencoder.py:9, 30-34) and ``decode(z)``.  ``encode`` returns (mean, logvar) in fp32; sampling
    ``mean + exp(0.5 logvar) * eps`` and the 0.18215 scale are the caller's (``LatentDiffusion.get_first_stage_encoding``)."""

    def __init__(self, ddconfig=None, embed_dim=4):
        super().__init__(ddconfig, embed_dim)
        ddconfig = dict(ddconfig or self.SD15_DDCONFIG)
        self.encoder = Encoder(**ddconfig)
        self.quant_conv = Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)

    @torch.no_grad()
    def encode(self, x, mask=None):
        moments = self.quant_conv(self.encoder(x, mask).float())          # quant_conv: 8 -> 8 channels, 1x1 (plain af_gemm)
        mean, logvar = torch.chunk(moments.float(), 2, dim=1)
        return mean, torch.clamp(logvar, -30.0, 20.0)
