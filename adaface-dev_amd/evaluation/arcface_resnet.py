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

Inference only, like the reference uses it (``arcface_wrapper.py:65-76`` runs it frozen in eval mode, fp16): training-mode
BatchNorm statistics / Dropout are not part of the hot path and raise."""
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

    def logits(self, x, packs):
        """x NHWC fp16 -> pre-sigmoid channel gates [B, C]."""
        s = ops.gemm(ops.global_avgpool(x), packs[0])
        return ops.gemm(ops.affine_prelu(s, slope=self.fc[1].weight), packs[1])


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

    def hip(self, x, P):
        out = ops.affine_prelu(x, P["bn0"][0], P["bn0"][1])
        out = ops.affine_prelu(ops.conv3x3(out, P["conv1"]), slope=self.prelu.weight)
        out = ops.conv3x3(out, P["conv2"], stride=self.stride)
        gates = self.se.logits(out, P["se"]) if self.use_se else None
        res = x if P["down"] is None else ops.conv3x3(x, P["down"], stride=self.stride)
        return ops.se_residual_prelu(out, gates, res, self.prelu.weight)


class ResNetFace(nn.Module):
    inference_only = True        # forward kernels only (ldm/modules/arcface_wrapper.py refuses a call that would need its backward)

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

    def forward(self, x):
        if self.training:
            raise NotImplementedError("ResNetFace runs frozen in eval mode on the AdaFace path (arcface_wrapper.py:65-76); "
                                      "training-mode BatchNorm / Dropout are not implemented")
        P = self._prepared()
        B = x.shape[0]
        assert x.shape[1:] == (1, 128, 128), "ResNetFace-18 takes [B, 1, 128, 128] grey crops (fc5 is 512*8*8 wide)"
        h = ops.nchw_f32_to_nhwc_f16(x.float().contiguous(), cpad=8)
        h = ops.affine_prelu(ops.conv3x3(h, P["conv1"]), slope=self.prelu.weight)
        h = ops.maxpool2x2(h)
        for blk, bp in zip(self.blocks(), P["blocks"]):
            h = blk.hip(h, bp)
        y = ops.gemm(h.reshape(B, 8 * 8 * 512), P["fc5"])                          # bn4 . flatten . fc5 . bn5
        return y if x.dtype == F16 else y.to(x.dtype)


def resnet_face18(use_se=True, **kwargs):
    return ResNetFace(IRBlock, [2, 2, 2, 2], use_se=use_se, **kwargs)
