"""ArcFace ResNetFace-18 IR-SE on a real MI355X (`pytest -m gpu`): the face-encoder kernels one by one against torch fp32,
then the whole trunk against the REFERENCE module's outputs (tests/golden/arcface.npz) and the CPU oracle.
fp16 storage, fp32 accumulation: 2e-3 per op, 1e-2 through the 17-conv trunk."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def test_face_elementwise_kernels_vs_torch(dev):
    from adaface_dev_amd import ops, rng
    B, H, W, C = 3, 12, 10, 72
    x = rng.synth_input("fk.x", (B, H, W, C), seed=51)
    r = rng.synth_input("fk.r", (B, H, W, C), seed=51)
    s = 1.0 + 0.2 * rng.synth_input("fk.s", (C,), seed=51)
    t = 0.3 * rng.synth_input("fk.t", (C,), seed=51)
    se = rng.synth_input("fk.se", (B, C), seed=51)
    slope = torch.tensor([0.2])
    xd, rd = x.to(dev).half(), r.to(dev).half()
    xh, rh = xd.float().cpu(), rd.float().cpu()
    y = ops.affine_prelu(xd, s.to(dev), t.to(dev), slope.to(dev))
    assert rel_l2(y.float().cpu().numpy(), F.prelu(xh * s + t, slope).numpy()) < 1e-3
    y = ops.affine_prelu(xd, None, None, slope.to(dev))
    assert rel_l2(y.float().cpu().numpy(), F.prelu(xh, slope).numpy()) < 1e-3
    y = ops.affine_prelu(xd, s.to(dev), t.to(dev), None)
    assert rel_l2(y.float().cpu().numpy(), (xh * s + t).numpy()) < 1e-3
    y = ops.maxpool2x2(xd)
    assert torch.equal(y.float().cpu(), F.max_pool2d(xh.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1))       # exact
    y = ops.global_avgpool(xd)
    assert rel_l2(y.float().cpu().numpy(), xh.mean(dim=(1, 2)).numpy()) < 1e-3
    sed = se.to(dev).half()
    y = ops.se_residual_prelu(xd, sed, rd, slope.to(dev))
    ref = F.prelu(xh * torch.sigmoid(sed.float().cpu())[:, None, None, :] + rh, slope)
    assert rel_l2(y.float().cpu().numpy(), ref.numpy()) < 1e-3
    y = ops.se_residual_prelu(xd, None, rd, slope.to(dev))
    assert rel_l2(y.float().cpu().numpy(), F.prelu(xh + rh, slope).numpy()) < 1e-3


def test_resnet_face18_vs_reference_and_oracle(dev):
    from adaface_dev_amd import rng
    from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
    from oracle import face_oracle as FO
    g = np.load(os.path.join(GOLDEN, "arcface.npz"))
    m = resnet_face18(use_se=True).eval()
    sd = rng.synth_face_state_dict(m.state_dict(), seed=50)
    m.load_state_dict(sd)
    m = m.to(dev)
    x = rng.synth_input("face.x", (3, 1, 128, 128), seed=50)
    with torch.no_grad():
        y = m(x.to(dev))
    assert y.shape == (3, 512) and y.dtype == torch.float32
    e_ref = rel_l2(y.cpu().numpy(), g["emb"])
    print(f"ResNetFace-18 embedding rel-L2 vs reference module: {e_ref:.3e}")
    assert e_ref < 1e-2
    # fp16 in / fp16 out, as arcface_wrapper.py:65,76 calls it
    with torch.no_grad():
        y16 = m(x.to(dev).half())
    assert y16.dtype == torch.float16 and rel_l2(y16.float().cpu().numpy(), g["emb"]) < 1e-2
    # a different batch size / input against the oracle; cosine similarity of the embeddings (what the ID loss consumes)
    x2 = rng.synth_input("face.x2", (5, 1, 128, 128), seed=52)
    with torch.no_grad():
        y2 = m(x2.to(dev)).cpu()
        r2 = FO.resnet_face18(sd, x2)
    assert rel_l2(y2.numpy(), r2.numpy()) < 1e-2
    assert float(F.cosine_similarity(y2, r2, dim=-1).min()) > 0.9999
    # use_se=False variant (reference ctor flag)
    m0 = resnet_face18(use_se=False).eval()
    sd0 = rng.synth_face_state_dict(m0.state_dict(), seed=53)
    m0.load_state_dict(sd0)
    with torch.no_grad():
        y0 = m0.to(dev)(x.to(dev)).cpu()
        r0 = FO.resnet_face18(sd0, x, use_se=False)
    assert rel_l2(y0.numpy(), r0.numpy()) < 1e-2
    # weight update invalidates the folded packs
    with torch.no_grad():
        m.bn5.bias.add_(1.0)
        y3 = m(x.to(dev)).cpu()
    assert rel_l2((y3 - 1.0).numpy(), g["emb"]) < 1e-2
