"""Host-side mirror of the reference's ``evaluation/arcface_resnet.py`` face-recognition trunk that feeds the ID-embedding
path: ``SEBlock`` (:139-154), ``IRBlock`` (:62-97), ``ResNetFace`` (:157-217) and the ``resnet_face18`` factory (:337-339).

Same module tree and parameter / buffer names (``conv1.weight``, ``bn1.running_mean``, ``layer2.0.downsample.0.weight``,
``layer1.0.se.fc.0.weight``, ``fc5.weight`` ...), so ``arcface-resnet18_110.pth`` loads with ``load_state_dict``.  The modules
only hold parameters; execution is NHWC fp16 through the C ABI:

* every convolution is the implicit-GEMM MFMA kernel (``af_gemm``), with the eval-mode BatchNorm that FOLLOWS it folded into
  weights and bias (bn1 / bn2 / downsample BN), the 1x1 stride-2 shortcut embedded in the centre tap of a 3x3 stride-2 filter;
* ``bn0`` precedes a zero-padded conv, so it cannot be folded (the shift would leak into the border): ``af_affine_prelu``;
* SE squeeze ``af_global_avgpool`` -> two tiny GEMMs (hidden width zero-padded to 8) -> excite, shortcut add and the block's
  final PReLU in ONE pass (``af_se_residual_prelu``);
* ``bn4`` -> flatten (NCHW order) -> ``fc5`` -> ``bn5`` collapses into one GEMM on NHWC-ordered, pre-scaled weights.

Frozen and in eval mode, like the reference uses it (``arcface_wrapper.py:65-76``, fp16): training-mode BatchNorm statistics / Dropout
are not part of the hot path and raise.  The ArcFace alignment loss differentiates THROUGH the frozen encoder into the decoded image
(``arcface_wrapper.py:89-166``; ``ddpm.py:2511-2535``), so ``forward`` is a ``torch.autograd`` node with an INPUT gradient
(``FaceEncodeFn``): transposed / flipped packs of the same folded weights for the convolutions' dgrad (stride 2 as the zero-inserted
implicit GEMM the U-Net's Downsample uses), ``af_affine_prelu_bwd`` / ``af_maxpool2x2_bwd`` / ``af_se_gate_grad`` /
``af_se_residual_prelu_bwd`` for the rest.  Parameter gradients are never formed."""
import torch
import torch.nn as nn

from .. import ops
from ..ops import F16


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _bn_affine(bn):
    """eval-mode BatchNorm as y = x * s + t."""
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
    return s, bn.bias.detach().float() - bn.running_mean.float() * s


def _fold_conv(conv, bn, dev, cin_pad=0):
    w = conv.weight.detach().float()
    if bn is not None:
        s, t = _bn_affine(bn)
        w = w * s[:, None, None, None]
    else:
        t = None
    if w.shape[-1] == 1:                                   # 1x1 (stride-2) shortcut -> centre tap of a 3x3 filter
        w3 = torch.zeros((w.shape[0], w.shape[1], 3, 3), dtype=w.dtype, device=w.device)
        w3[:, :, 1, 1] = w[:, :, 0, 0]
        w = w3
    return ops.pack_conv3x3(w, t, dev, cin_pad=cin_pad)


def _fold_conv_bwd(conv, bn, dev, n_pad=0):
    """Pack of the input-gradient convolution of ``_fold_conv(conv, bn)``: spatially flipped, in/out channels swapped, no bias;
    n_pad zero-pads the gradient's channel count (the grey input's 1 -> 8)."""
    w = conv.weight.detach().float()
    if bn is not None:
        w = w * _bn_affine(bn)[0][:, None, None, None]
    if w.shape[-1] == 1:
        w3 = torch.zeros((w.shape[0], w.shape[1], 3, 3), dtype=w.dtype, device=w.device)
        w3[:, :, 1, 1] = w[:, :, 0, 0]
        w = w3
    wd = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()                  # [Cin, Cout, 3, 3]
    if n_pad > wd.shape[0]:
        wd = torch.cat([wd, torch.zeros((n_pad - wd.shape[0],) + tuple(wd.shape[1:]), dtype=wd.dtype, device=wd.device)])
    return ops.pack_conv3x3(wd, None, dev)


class SEBlock(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction), nn.PReLU(), nn.Linear(channel // reduction, channel),
                                nn.Sigmoid())

    def pack(self, dev):
        f0, f2 = self.fc[0], self.fc[2]
        hid = ops.round_up(f0.out_features, 8)             # zero-padded hidden units: prelu(0) = 0, zero columns in fc2
        w0 = torch.zeros((hid, f0.in_features), device=f0.weight.device)
        b0 = torch.zeros((hid,), device=f0.weight.device)
        w0[:f0.out_features], b0[:f0.out_features] = f0.weight.detach().float(), f0.bias.detach().float()
        w2 = torch.zeros((f2.out_features, hid), device=f2.weight.device)
        w2[:, :f2.in_features] = f2.weight.detach().float()
        return ops.pack_matrix(w0, b0, dev), ops.pack_matrix(w2, f2.bias, dev)

    def pack_bwd(self, dev):
        """Transposes of ``pack``'s two matrices (same zero padding), for the squeeze branch's input gradient."""
        f0, f2 = self.fc[0], self.fc[2]
        hid = ops.round_up(f0.out_features, 8)
        w0t = torch.zeros((f0.in_features, hid), device=f0.weight.device)
        w0t[:, :f0.out_features] = f0.weight.detach().float().t()
        w2t = torch.zeros((hid, f2.out_features), device=f2.weight.device)
        w2t[:f2.in_features] = f2.weight.detach().float().t()
        return ops.pack_matrix(w0t, None, dev), ops.pack_matrix(w2t, None, dev)

    def logits(self, x, packs, keep=None):
        """x NHWC fp16 -> pre-sigmoid channel gates [B, C]; keep (a list) receives the hidden pre-activation for ``logits_bwd``."""
        s = ops.gemm(ops.global_avgpool(x), packs[0])
        if keep is not None:
            keep.append(s)
        return ops.gemm(ops.affine_prelu(s, slope=self.fc[1].weight), packs[1])

    def logits_bwd(self, dgl, s, packs_bwd):
        """dgl [B, C] (gate-logit gradient / HW) -> gradient of the pooled input [B, C] (/ HW: what every pixel of x receives)."""
        ds = ops.affine_prelu_bwd(ops.gemm(dgl, packs_bwd[1]), s, slope=self.fc[1].weight)
        return ops.gemm(ds, packs_bwd[0])


class IRBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_se=True):
        super().__init__()
        self.bn0 = nn.BatchNorm2d(inplanes)
        self.conv1 = conv3x3(inplanes, inplanes)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.prelu = nn.PReLU()
        self.conv2 = conv3x3(inplanes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.use_se = use_se
        if self.use_se:
            self.se = SEBlock(planes)

    def pack(self, dev):
        s0, t0 = _bn_affine(self.bn0)
        return dict(bn0=(s0.to(dev).contiguous(), t0.to(dev).contiguous()), conv1=_fold_conv(self.conv1, self.bn1, dev),
                    conv2=_fold_conv(self.conv2, self.bn2, dev), se=self.se.pack(dev) if self.use_se else None,
                    down=None if self.downsample is None else _fold_conv(self.downsample[0], self.downsample[1], dev))

    def pack_bwd(self, dev):
        return dict(conv1=_fold_conv_bwd(self.conv1, self.bn1, dev), conv2=_fold_conv_bwd(self.conv2, self.bn2, dev),
                    se=self.se.pack_bwd(dev) if self.use_se else None,
                    down=None if self.downsample is None else _fold_conv_bwd(self.downsample[0], self.downsample[1], dev))

    def hip_train(self, x, P):
        """``hip`` that also returns what ``hip_bwd`` reads."""
        c1 = ops.conv3x3(ops.affine_prelu(x, P["bn0"][0], P["bn0"][1]), P["conv1"])
        c2 = ops.conv3x3(ops.affine_prelu(c1, slope=self.prelu.weight), P["conv2"], stride=self.stride)
        keep = []
        gates = self.se.logits(c2, P["se"], keep) if self.use_se else None
        res = x if P["down"] is None else ops.conv3x3(x, P["down"], stride=self.stride)
        y = ops.se_residual_prelu(c2, gates, res, self.prelu.weight)
        return y, (c1, c2, gates, res, keep[0] if keep else None, (x.shape[1], x.shape[2]))

    def hip_bwd(self, saved, dy, P, Pb):
        """dy [B, Ho, Wo, planes] -> gradient of the block input."""
        c1, c2, gates, res, s, in_hw = saved
        sl = self.prelu.weight
        dpool = None
        if self.use_se:
            dpool = self.se.logits_bwd(ops.se_gate_grad(c2, gates, res, sl, dy), s, Pb["se"])
        dc2, dres = ops.se_residual_prelu_bwd(c2, gates, res, sl, dy, dpool)
        if self.stride == 2:
            da = ops.conv3x3(dc2, Pb["conv2"], upsample=2, out_hw=in_hw)
        else:
            da = ops.conv3x3(dc2, Pb["conv2"])
        dc1 = ops.affine_prelu_bwd(da, c1, slope=sl)
        dx = ops.affine_prelu_bwd(ops.conv3x3(dc1, Pb["conv1"]), None, P["bn0"][0], P["bn0"][1])
        if Pb["down"] is not None:
            dres = ops.conv3x3(dres, Pb["down"], upsample=2, out_hw=in_hw) if self.stride == 2 else ops.conv3x3(dres, Pb["down"])
        return ops.add(dx, dres)

    def hip(self, x, P):
        out = ops.affine_prelu(x, P["bn0"][0], P["bn0"][1])
        out = ops.affine_prelu(ops.conv3x3(out, P["conv1"]), slope=self.prelu.weight)
        out = ops.conv3x3(out, P["conv2"], stride=self.stride)
        gates = self.se.logits(out, P["se"]) if self.use_se else None
        res = x if P["down"] is None else ops.conv3x3(x, P["down"], stride=self.stride)
        return ops.se_residual_prelu(out, gates, res, self.prelu.weight)


class FaceEncodeFn(torch.autograd.Function):
    """grey crops [B, 1, 128, 128] -> embeddings [B, 512] with the input gradient (see the module docstring).  The incoming gradient is
    normalised by a power of two before its fp16 cast (largest entry ~ 256, like the U-Net's backward node) and the result unscaled."""

    @staticmethod
    def forward(ctx, x, net):
        y, saved = net.hip_train(x.detach())
        ctx.net, ctx.saved, ctx.dtype = net, saved, x.dtype
        return y if x.dtype == F16 else y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        if ctx.saved is None:
            raise RuntimeError("FaceEncodeFn: the saved activations were released by the first backward; a second backward through the same "
                               "embedding call is not supported")
        amax = dy.detach().abs().amax().float().clamp_min(1e-30)
        scale = torch.exp2(torch.floor(torch.log2(256.0 / amax)))
        dx = ctx.net.hip_bwd(ctx.saved, (dy.float() * scale).to(F16).contiguous())
        ctx.saved = None
        return (dx / scale).to(ctx.dtype), None


class ResNetFace(nn.Module):
    inference_only = False       # forward() carries an input gradient (FaceEncodeFn); parameters stay frozen

    def __init__(self, block, layers, use_se=True):
        self.inplanes = 64
        self.use_se = use_se
        super().__init__()
        self.conv1 = nn.Conv2d(1, 64, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.prelu = nn.PReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.bn4 = nn.BatchNorm2d(512)
        self.dropout = nn.Dropout()
        self.fc5 = nn.Linear(512 * 8 * 8, 512)
        self.bn5 = nn.BatchNorm1d(512)
        for m in self.modules():                                              # reference initialisation (:174-182)
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)
        self._packs, self._packs_key = None, None
        self._packs_bwd, self._packs_bwd_key = None, None

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, use_se=self.use_se)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, use_se=self.use_se))
        return nn.Sequential(*layers)

    def blocks(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            yield from layer

    def _prepared(self):
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if key != self._packs_key:
            dev = self.conv1.weight.device
            if not self.conv1.weight.is_cuda:
                raise RuntimeError("ResNetFace: parameters are on the CPU; this model only runs on an MI355X (HIP extension, no CPU "
                                   "fallback). Move it with .cuda() first.")
            s4, t4 = _bn_affine(self.bn4)
            s5, t5 = _bn_affine(self.bn5)
            w5 = self.fc5.weight.detach().float().reshape(512, 512, 8, 8)          # columns in NCHW flatten order (:212)
            b5 = self.fc5.bias.detach().float() + (w5 * t4[None, :, None, None]).sum(dim=(1, 2, 3))
            w5 = (w5 * s4[None, :, None, None]).permute(0, 2, 3, 1).reshape(512, 8 * 8 * 512)       # -> NHWC flatten order
            self._packs = dict(conv1=_fold_conv(self.conv1, self.bn1, dev, cin_pad=8), blocks=[b.pack(dev) for b in self.blocks()],
                               fc5=ops.pack_matrix(w5 * s5[:, None], b5 * s5 + t5, dev))
            self._packs_key = key
        return self._packs

    def _prepared_bwd(self):
        P = self._prepared()
        if self._packs_bwd_key != self._packs_key:
            dev = self.conv1.weight.device
            self._packs_bwd = dict(conv1=_fold_conv_bwd(self.conv1, self.bn1, dev, n_pad=8), blocks=[b.pack_bwd(dev) for b in self.blocks()],
                                   fc5=ops.pack_matrix(P["fc5"].wt[:512, :8 * 8 * 512].t(), None, dev))
            self._packs_bwd_key = self._packs_key
        return self._packs_bwd

    def _check(self, x):
        if self.training:
            raise NotImplementedError("ResNetFace runs frozen in eval mode on the AdaFace path (arcface_wrapper.py:65-76); "
                                      "training-mode BatchNorm / Dropout are not implemented")
        assert x.shape[1:] == (1, 128, 128), "ResNetFace-18 takes [B, 1, 128, 128] grey crops (fc5 is 512*8*8 wide)"

    def hip_train(self, x):
        """forward -> (embeddings fp16 [B, 512], activations kept for ``hip_bwd``)."""
        self._check(x)
        P = self._prepared()
        B = x.shape[0]
        c1 = ops.conv3x3(ops.nchw_f32_to_nhwc_f16(x.float().contiguous(), cpad=8), P["conv1"])
        a1 = ops.affine_prelu(c1, slope=self.prelu.weight)
        h = ops.maxpool2x2(a1)
        saved = []
        for blk, bp in zip(self.blocks(), P["blocks"]):
            h, sv = blk.hip_train(h, bp)
            saved.append(sv)
        return ops.gemm(h.reshape(B, 8 * 8 * 512), P["fc5"]), (c1, a1, saved)

    def hip_bwd(self, saved, dy):
        """dy fp16 [B, 512] -> fp32 [B, 1, 128, 128]."""
        P, Pb = self._prepared(), self._prepared_bwd()
        c1, a1, per_block = saved
        B = dy.shape[0]
        dh = ops.gemm(dy, Pb["fc5"]).reshape(B, 8, 8, 512)
        for blk, bp, bpb, sv in reversed(list(zip(self.blocks(), P["blocks"], Pb["blocks"], per_block))):
            dh = blk.hip_bwd(sv, dh, bp, bpb)
        dc1 = ops.affine_prelu_bwd(ops.maxpool2x2_bwd(a1, dh), c1, slope=self.prelu.weight)
        return ops.nhwc_f16_to_nchw_f32(ops.conv3x3(dc1, Pb["conv1"]), 1)

    def forward(self, x):
        if x.requires_grad and torch.is_grad_enabled():
            self._check(x)
            return FaceEncodeFn.apply(x, self)
        self._check(x)
        P = self._prepared()
        B = x.shape[0]
        h = ops.nchw_f32_to_nhwc_f16(x.float().contiguous(), cpad=8)
        h = ops.affine_prelu(ops.conv3x3(h, P["conv1"]), slope=self.prelu.weight)
        h = ops.maxpool2x2(h)
        for blk, bp in zip(self.blocks(), P["blocks"]):
            h = blk.hip(h, bp)
        y = ops.gemm(h.reshape(B, 8 * 8 * 512), P["fc5"])                          # bn4 . flatten . fc5 . bn5
        return y if x.dtype == F16 else y.to(x.dtype)


def resnet_face18(use_se=True, **kwargs):
    return ResNetFace(IRBlock, [2, 2, 2, 2], use_se=use_se, **kwargs)
