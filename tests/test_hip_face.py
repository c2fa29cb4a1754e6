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


def test_face_backward_kernels_vs_torch_autograd(dev):
    """The input-gradient kernels of the encoder's element-wise layers against torch autograd on the same fp16-rounded inputs."""
    from adaface_dev_amd import ops, rng
    B, H, W, C = 3, 12, 10, 72
    mk = lambda name, shape: rng.synth_input(name, shape, seed=57)
    x, r, dy = mk("fb.x", (B, H, W, C)), mk("fb.r", (B, H, W, C)), mk("fb.dy", (B, H, W, C))
    s, t = 1.0 + 0.2 * mk("fb.s", (C,)), 0.3 * mk("fb.t", (C,))
    se, slope = mk("fb.se", (B, C)), torch.tensor([0.2])
    h = lambda a: a.to(dev).half()
    back = lambda a: a.float().cpu()
    xd, rd, dyd, sed = h(x), h(r), h(dy), h(se)
    xh, rh, dyh, seh = back(xd), back(rd), back(dyd), back(sed)
    # affine + prelu, prelu only, affine only
    for sc, sh, sl in ((s, t, slope), (None, None, slope), (s, t, None)):
        xr = xh.clone().requires_grad_(True)
        f = xr if sc is None else xr * sc + sh
        (F.prelu(f, sl) if sl is not None else f).backward(dyh)
        got = ops.affine_prelu_bwd(dyd, xd if sl is not None else None, None if sc is None else sc.to(dev), None if sc is None else sh.to(dev),
                                   None if sl is None else sl.to(dev))
        assert rel_l2(back(got).numpy(), xr.grad.numpy()) < 1e-3
    # max pool: gradient to the first maximum (ties made on purpose)
    xt = xh.clone()
    xt[:, ::2, ::2] = xt[:, 1::2, 1::2]
    xr = xt.clone().requires_grad_(True)
    dyp = dyh[:, :H // 2, :W // 2].contiguous()
    F.max_pool2d(xr.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).backward(dyp)
    got = ops.maxpool2x2_bwd(h(xt), h(dyp))
    assert torch.equal(back(got), xr.grad)
    # SE excite + shortcut + PReLU: d(logits) / HW, dx (with a squeeze-branch gradient added), dresidual
    xr, rr, ser = xh.clone().requires_grad_(True), rh.clone().requires_grad_(True), seh.clone().requires_grad_(True)
    F.prelu(xr * torch.sigmoid(ser)[:, None, None, :] + rr, slope).backward(dyh)
    dgl = ops.se_gate_grad(xd, sed, rd, slope.to(dev), dyd)
    assert rel_l2(back(dgl).numpy() * (H * W), ser.grad.numpy()) < 2e-3
    dpool = h(0.1 * mk("fb.dpool", (B, C)))
    dx, dres = ops.se_residual_prelu_bwd(xd, sed, rd, slope.to(dev), dyd, dpool)
    assert rel_l2(back(dx).numpy(), (xr.grad + back(dpool)[:, None, None, :]).numpy()) < 1e-3
    assert rel_l2(back(dres).numpy(), rr.grad.numpy()) < 1e-3
    xr, rr = xh.clone().requires_grad_(True), rh.clone().requires_grad_(True)
    F.prelu(xr + rr, slope).backward(dyh)
    dx, dres = ops.se_residual_prelu_bwd(xd, None, rd, slope.to(dev), dyd)
    assert rel_l2(back(dx).numpy(), xr.grad.numpy()) < 1e-3 and rel_l2(back(dres).numpy(), rr.grad.numpy()) < 1e-3


def _face_net(dev, use_se, linear_prelu=False):
    from adaface_dev_amd import rng
    from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
    m = resnet_face18(use_se=use_se).eval()
    sd = rng.synth_face_state_dict(m.state_dict(), seed=50)
    if linear_prelu:                          # slope 1: no sign decision left in the blocks (the SE sigmoid stays)
        sd = {k: (torch.ones_like(v) if k.endswith("prelu.weight") else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    return m.to(dev), sd


@pytest.mark.parametrize("use_se", [True, False])
def test_ir_block_input_gradients_vs_oracle_autograd(dev, use_se):
    """Every IRBlock's hip_train / hip_bwd against torch autograd through the oracle's block on the same (fp16-rounded) block input.
    With PReLU slope 1 the block has no sign decision, so the chain of dgrad convolutions / folded BatchNorms / SE gradient must
    agree to fp16 rounding (1.5e-3); with the real slopes the few activations whose sign differs between fp16 and fp32 arithmetic each
    contribute a full (1 - slope) * dy, measured ~1e-2 per block (tools/probes/r03t_face_grad.py): 3e-2."""
    from oracle import face_oracle as FO
    x = __import__("adaface_dev_amd").rng.synth_input("face.gx", (2, 1, 128, 128), seed=58)
    for linear, tol in ((True, 1.5e-3), (False, 3e-2)):
        m, sd = _face_net(dev, use_se, linear)
        P, Pb = m._prepared(), m._prepared_bwd()
        with torch.no_grad():
            h = F.max_pool2d(F.prelu(FO._bn(sd, "bn1.", F.conv2d(x, sd["conv1.weight"], None, 1, 1)), sd["prelu.weight"]), 2, 2)
        names = [(f"layer{li}.{bi}.", 2 if (li > 1 and bi == 0) else 1) for li in range(1, 5) for bi in range(2)]
        for (p, stride), blk, bp, bpb in zip(names, m.blocks(), P["blocks"], Pb["blocks"]):
            hin = h.half().float()
            hr = hin.clone().requires_grad_(True)
            out = FO.ir_block(sd, p, hr, stride, use_se)
            dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).half().float()
            out.backward(dy)
            y, sv = blk.hip_train(hin.permute(0, 2, 3, 1).contiguous().to(dev).half(), bp)
            dx = blk.hip_bwd(sv, dy.permute(0, 2, 3, 1).contiguous().to(dev).half(), bp, bpb)
            assert rel_l2(y.float().cpu().permute(0, 3, 1, 2).numpy(), out.detach().numpy()) < 1e-3, p
            e = rel_l2(dx.float().cpu().permute(0, 3, 1, 2).numpy(), hr.grad.numpy())
            assert e < tol, (p, linear, e)
            h = out.detach()


@pytest.mark.parametrize("use_se", [True, False])
def test_resnet_face18_input_gradient_vs_oracle_autograd(dev, use_se):
    """d(loss)/d(grey crops) through the whole frozen encoder (what the ArcFace alignment loss back-propagates,
    arcface_wrapper.py:89-166) against torch autograd through the CPU oracle on the same weights, fp32 and fp16 callers.  The
    kernels are exact to rounding (block test above); what accumulates over 8 blocks + the stem's max-pool is the sign / argmax
    decisions that differ between fp16 and fp32 arithmetic (~1e-2 per block, 2e-2 in the stem): 7e-2 rel-L2, cosine > 0.998.
    No parameter receives a gradient."""
    from adaface_dev_amd import rng
    from oracle import face_oracle as FO
    m, sd = _face_net(dev, use_se)
    x = rng.synth_input("face.gx", (3, 1, 128, 128), seed=58)
    target = F.normalize(rng.synth_input("face.gt", (3, 512), seed=58), dim=-1)
    loss_of = lambda emb, tg: (1 - F.cosine_similarity(emb.float(), tg, dim=-1)).sum() + 1e-3 * (emb.float() ** 2).mean()
    xr = x.clone().requires_grad_(True)
    loss_ref = loss_of(FO.resnet_face18(sd, xr, use_se=use_se), target)
    loss_ref.backward()
    for dt in (torch.float32, torch.float16):
        xd = x.to(dev).to(dt).requires_grad_(True)
        loss = loss_of(m(xd), target.to(dev))
        loss.backward()
        assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-2 * abs(float(loss_ref.detach()))
        got = xd.grad.float().cpu()
        e = rel_l2(got.numpy(), xr.grad.numpy())
        cos = float(F.cosine_similarity(got.flatten(), xr.grad.flatten(), dim=0))
        print(f"ResNetFace-18 (use_se={use_se}, {dt}) input-gradient rel-L2 vs oracle autograd: {e:.3e}  cosine {cos:.5f}")
        assert xd.grad.dtype == dt and e < 7e-2 and cos > 0.998
    assert all(p.grad is None for p in m.parameters())
    # the same call under no_grad / on a tensor without grad stays the plain forward
    with torch.no_grad():
        assert not m(x.to(dev)).requires_grad


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
