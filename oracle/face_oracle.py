"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product path) for the ArcFace ResNetFace-18 IR-SE trunk:
a functional fp32 restatement of reference evaluation/arcface_resnet.py (IRBlock.forward :76-97, SEBlock.forward :149-153,
ResNetFace.forward :199-216) over a plain state dict, eval-mode BatchNorm.

Pinned: tests/test_oracle_vs_golden.py checks it against outputs of the REFERENCE module itself on seeded weights
(tests/golden/arcface.npz, written by tests/golden/gen_golden.py which imports the reference in the build container)."""
import torch
import torch.nn.functional as F


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, eps)


def se_block(sd, p, x):
    b, c = x.shape[:2]
    y = x.mean(dim=(2, 3))
    y = F.linear(y, sd[p + "fc.0.weight"], sd[p + "fc.0.bias"])
    y = F.prelu(y, sd[p + "fc.1.weight"])
    y = torch.sigmoid(F.linear(y, sd[p + "fc.2.weight"], sd[p + "fc.2.bias"]))
    return x * y.view(b, c, 1, 1)


def ir_block(sd, p, x, stride, use_se=True):
    out = _bn(sd, p + "bn0.", x)
    out = F.conv2d(out, sd[p + "conv1.weight"], None, 1, 1)
    out = F.prelu(_bn(sd, p + "bn1.", out), sd[p + "prelu.weight"])
    out = _bn(sd, p + "bn2.", F.conv2d(out, sd[p + "conv2.weight"], None, stride, 1))
    if use_se:
        out = se_block(sd, p + "se.", out)
    res = x
    if p + "downsample.0.weight" in sd:
        res = _bn(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], None, stride, 0))
    return F.prelu(out + res, sd[p + "prelu.weight"])


def resnet_face18(sd, x, use_se=True, return_features=False):
    """x [B,1,128,128] -> [B,512]."""
    h = F.prelu(_bn(sd, "bn1.", F.conv2d(x, sd["conv1.weight"], None, 1, 1)), sd["prelu.weight"])
    h = F.max_pool2d(h, 2, 2)
    feats = []
    for li in range(1, 5):
        for bi in range(2):
            h = ir_block(sd, f"layer{li}.{bi}.", h, 2 if (li > 1 and bi == 0) else 1, use_se)
        feats.append(h)
    h = _bn(sd, "bn4.", h)
    h = F.linear(h.reshape(h.shape[0], -1), sd["fc5.weight"], sd["fc5.bias"])
    h = _bn(sd, "bn5.", h)
    return (h, feats) if return_features else h
