"""Backward kernels on a real MI355X against torch autograd (fp32, CPU) of the same op on the same
fp16-rounded inputs.  `pytest -m gpu`.  Gradients are rounded to fp16 once on output: bound 3e-3 rel-L2
(attention backward: 5e-3 -- P and dz are rounded to fp16 before the second MFMA product)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 3e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.float16)


@pytest.mark.parametrize("B,HW,c1,c2,silu,with_add", [(2, 64, 320, 0, True, True), (2, 100, 640, 320, True, False),
                                                        (1, 64, 2560, 0, True, True), (2, 256, 64, 0, False, True),
                                                        (2, 49, 64, 64, True, True)])
def test_groupnorm_bwd(dev, B, HW, c1, c2, silu, with_add):
    from adaface_dev_amd import ops
    C = c1 + c2
    x1 = (rnd((B, HW, c1), 1, 2.0).float() + 0.5).half()
    x2 = rnd((B, HW, c2), 2) if c2 else None
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.2 + 1
    b = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2
    dy, add = rnd((B, HW, C), 5), (rnd((B, HW, C), 6) if with_add else None)
    y, stats = ops.groupnorm_train(x1.to(dev), g.to(dev), b.to(dev), 1e-5, silu, x2=None if x2 is None else x2.to(dev))
    res = ops.groupnorm_bwd(x1.to(dev), g.to(dev), b.to(dev), stats, dy.to(dev), silu, x2=None if x2 is None else x2.to(dev),
                            add=None if add is None else add.to(dev))
    xc = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().requires_grad_(True)
    ref = F.group_norm(xc.permute(0, 2, 1), 32, g, b, 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1)
    assert rel_l2(y.float().cpu().numpy(), ref.detach().numpy()) < TOL
    ref.backward(dy.float())
    gx = xc.grad + (add.float() if add is not None else 0)
    got = res if x2 is None else torch.cat([res[0], res[1]], -1)
    assert rel_l2(got.float().cpu().numpy(), gx.numpy()) < TOL


@pytest.mark.parametrize("rows,C", [(77, 320), (300, 640), (130, 1280), (64, 64)])
def test_layernorm_bwd(dev, rows, C):
    from adaface_dev_amd import ops
    x = (rnd((rows, C), 1, 2.0).float() - 0.3).half()
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.2 + 1
    b = torch.zeros(C)
    dy, add = rnd((rows, C), 5), rnd((rows, C), 6)
    dx = ops.layernorm_bwd(x.to(dev), g.to(dev), dy.to(dev), 1e-5, add=add.to(dev))
    xr = x.float().requires_grad_(True)
    F.layer_norm(xr, (C,), g, b, 1e-5).backward(dy.float())
    assert rel_l2(dx.float().cpu().numpy(), (xr.grad + add.float()).numpy()) < TOL


@pytest.mark.parametrize("rows,C", [(77, 320), (300, 640), (388, 768), (2048, 1280), (5, 8), (1, 64)])
def test_layernorm_param_grads(dev, rows, C):
    """af_layernorm_param_grads: dgamma / dbeta of a LayerNorm from (x, dy) against torch autograd in fp32."""
    from adaface_dev_amd import ops
    x, dy = rnd((rows, C), 1, 2.0) + 0.5, rnd((rows, C), 2)
    dg, db = ops.layernorm_param_grads(x.to(dev), dy.to(dev), 1e-5)
    g = torch.ones(C, requires_grad=True)
    b = torch.zeros(C, requires_grad=True)
    F.layer_norm(x.float(), (C,), g, b, 1e-5).backward(dy.float())
    assert rel_l2(dg.cpu().numpy(), g.grad.numpy()) < 1e-4 and rel_l2(db.cpu().numpy(), b.grad.numpy()) < 1e-5


def test_geglu_fwd_bwd(dev):
    from adaface_dev_amd import ops
    M, I = 200, 128
    h = rnd((M, 2 * I), 1)           # natural layout [value | gate]
    dout = rnd((M, I), 2)
    hv, hg = h[:, :I].reshape(M, I // 16, 16), h[:, I:].reshape(M, I // 16, 16)
    hp = torch.stack([hv, hg], dim=2).reshape(M, 2 * I).contiguous()      # interleaved 16-col groups
    out = ops.geglu_fwd(hp.to(dev))
    dhp = ops.geglu_bwd(hp.to(dev), dout.to(dev))
    hr = h.float().requires_grad_(True)
    xv, gv = hr.chunk(2, dim=-1)
    ref = xv * F.gelu(gv)
    ref.backward(dout.float())
    assert rel_l2(out.float().cpu().numpy(), ref.detach().numpy()) < TOL
    dh = dhp.float().cpu().reshape(M, I // 16, 2, 16)
    got = torch.cat([dh[:, :, 0].reshape(M, I), dh[:, :, 1].reshape(M, I)], -1)
    assert rel_l2(got.numpy(), hr.grad.numpy()) < TOL


def test_sumpool_add_transpose(dev):
    from adaface_dev_amd import ops
    x = rnd((2, 8, 6, 16), 1)
    y = ops.sumpool2x2(x.to(dev))
    ref = F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2) * 4
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    a, b = rnd((3, 40), 2), rnd((3, 40), 3)
    assert torch.equal(ops.add(a.to(dev), b.to(dev)).cpu(), (a.float() + b.float()).half())
    t = rnd((2 * 77, 96), 4)
    tt = ops.transpose_tokens(t.to(dev)[:, 32:], 2, 77, 64, 96)
    assert tt.shape == (2, 64, 80)
    assert torch.equal(tt[:, :, :77].cpu(), t[:, 32:].reshape(2, 77, 64).permute(0, 2, 1))
    assert float(tt[:, :, 77:].abs().max()) == 0
    # the 16-byte form (aligned views) and the scalar fallback (a view that starts 4 channels in; 12 channels) give the same bits
    for (Bn, N, C, ld, off) in [(2, 77, 64, 96, 32), (3, 4096, 320, 960, 320), (1, 130, 40, 40, 0), (2, 200, 96, 100, 4), (2, 64, 12, 16, 0), (4, 1, 8, 8, 0)]:
        src = rnd((Bn * N, ld), 5)
        out = ops.transpose_tokens(src.to(dev)[:, off:], Bn, N, C, ld)
        assert out.shape == (Bn, C, (N + 7) // 8 * 8)
        assert torch.equal(out[:, :, :N].cpu(), src[:, off:off + C].reshape(Bn, N, C).permute(0, 2, 1)), (Bn, N, C, ld, off)
        assert float(out[:, :, N:].abs().max()) == 0 if out.shape[2] > N else True
    # two operands of a weight gradient in one launch (af_transpose_tokens_pair), incl. a pair that must fall back to the scalar form
    from adaface_dev_amd import _lib
    for (M, n1, n2) in [(300, 16, 320), (4096, 192, 2880), (77, 12, 64)]:
        a, b = rnd((M, n1), 6).to(dev), rnd((M, n2), 7).to(dev)
        m64 = (M + 63) // 64 * 64
        at, bt = torch.empty((n1, m64), dtype=torch.float16, device=dev), torch.empty((n2, m64), dtype=torch.float16, device=dev)
        _lib.check(_lib.lib().af_transpose_tokens_pair(ops._p(a), ops._p(at), n1, n1, ops._p(b), ops._p(bt), n2, n2, 1, M, m64, ops._stream()), "pair")
        assert torch.equal(at[:, :M], a.t()) and torch.equal(bt[:, :M], b.t())
        assert m64 == M or (float(at[:, M:].abs().max()) == 0 and float(bt[:, M:].abs().max()) == 0)


def test_conv_dgrad_via_transposed_weights(dev):
    """conv3x3 input gradient = conv3x3 with spatially flipped, in/out-transposed weights (stride 1);
    stride-2 forward: the same over the zero-inserted output gradient (upsample mode 2)."""
    from adaface_dev_amd import ops
    B, H, W, cin, cout = 2, 8, 8, 64, 128
    x = rnd((B, H, W, cin), 1)
    w = rnd((cout, cin, 3, 3), 2, (9 * cin) ** -0.5)
    wd = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()                       # [cin, cout, 3, 3]
    for stride in (1, 2):
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = F.conv2d(xr, w.float(), None, stride=stride, padding=1)
        dy = rnd(tuple(yr.permute(0, 2, 3, 1).shape), 3)
        yr.backward(dy.float().permute(0, 3, 1, 2))
        pw = ops.pack_conv3x3(wd, None, dev)
        if stride == 1:
            dx = ops.conv3x3(dy.to(dev), pw)
        else:
            dx = ops.conv3x3(dy.to(dev), pw, upsample=2, out_hw=(H, W))
        assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2).numpy(), xr.grad.numpy()) < TOL


@pytest.mark.parametrize("B,N,L,heads,d,masked", [(2, 64, 64, 8, 8, False), (1, 200, 77, 8, 40, False), (2, 160, 160, 4, 40, True),
                                                   (1, 256, 256, 2, 80, False), (1, 96, 96, 2, 160, False), (2, 70, 97, 4, 64, False),
                                                   (1, 1024, 1024, 2, 40, False)])
def test_attention_bwd(dev, B, N, L, heads, d, masked):
    from adaface_dev_amd import ops
    from oracle.unet_oracle import attention_core
    C = heads * d
    q, k, v, do = rnd((B, N, C), 1), rnd((B, L, C), 2), rnd((B, L, C), 3), rnd((B, N, C), 4)
    mask, kb = None, None
    if masked:
        mask = torch.rand(B, L, generator=torch.Generator().manual_seed(9)) > 0.3
        kb = ops.make_keybias(mask.to(dev), L)
    qd, kd, vd = q.reshape(B * N, C).to(dev), k.reshape(B * L, C).to(dev), v.reshape(B * L, C).to(dev)
    vt = ops.transpose_tokens(vd, B, L, C, C)
    o, lse = ops.attention(qd, kd, vt, B=B, Nq=N, L=L, heads=heads, d=d, ldq=C, ldk=C, keybias=kb, want_lse=True)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attention_bwd(qd, kd, vd, o, do.reshape(B * N, C).to(dev), lse, B=B, Nq=N, L=L, heads=heads, d=d, ldq=C, ldk=C, ldv=C,
                      dq=dq, dk=dk, dv=dv, lddq=C, lddk=C, lddv=C, keybias=kb)
    qr, kr, vr = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    ref = attention_core(qr, kr, vr, heads, mask)
    ref.backward(do.float())
    assert rel_l2(o.float().cpu().reshape(B, N, C).numpy(), ref.detach().numpy()) < TOL
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        err = rel_l2(got.float().cpu().reshape(want.shape).numpy(), want.numpy())
        assert err < 5e-3, (name, err)


def test_cadamw_step_matches_reference_trace(dev):
    """3 steps of the fused kernel == a plain restatement of c_adamw.py:65-123 (two tensors, wd on)."""
    from adaface_dev_amd import ops
    torch.manual_seed(0)
    shapes = [(37, 5), (130,)]
    ps = [torch.randn(s) for s in shapes]
    sizes = [p.numel() for p in ps]
    off = torch.tensor([0, sizes[0], sizes[0] + sizes[1]], dtype=torch.int64)
    flat = torch.cat([p.reshape(-1) for p in ps]).to(dev)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    rp = [p.clone() for p in ps]
    rm, rv = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.995, 1e-6, 0.02
    for step in range(1, 4):
        gs = [torch.randn(s) for s in shapes]
        gflat = torch.cat([g.reshape(-1) for g in gs]).to(dev)
        ops.cadamw_step(flat, gflat, m, v, off.to(dev), counts, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, step=step)
        for p, g, em, ev in zip(rp, gs, rm, rv):
            p.add_(p, alpha=-lr * wd)
            em.mul_(b1).add_(g, alpha=1 - b1)
            ev.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = ev.sqrt().add_(eps)
            ss = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
            mask = (em * g > 0).to(g.dtype)
            mask.div_(mask.mean().clamp_(min=1e-3))
            p.add_((em * mask) / denom, alpha=-ss)
    want = torch.cat([p.reshape(-1) for p in rp])
    assert rel_l2(flat.cpu().numpy(), want.numpy()) < 1e-5


@pytest.mark.parametrize("k,cin,cout,r", [(3, 96, 64, 32), (1, 96, 64, 32), (3, 64, 64, 192)])
def test_dora_conv_forward_backward_vs_oracle_autograd(dev, k, cin, cout, r):
    """Trainable DoRA adapter on a conv (3x3 and the 1x1 shortcut), with a dropout mask, fused rowbias + residual on the base
    conv and a two-source (concatenated) input: y, dx and the gradients of lora_A / lora_B / magnitude against autograd through
    the branch-form oracle (peft's DoraConv2dLayer restated; parity unpinned)."""
    from adaface_dev_amd import ops, rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import Conv2d
    from adaface_dev_amd.ldm.modules.dora import DoRAConvAdapter, dora_conv_bwd, dora_conv_fwd
    from oracle import lora_oracle as LO
    B, H, W = 2, 12, 10
    conv = Conv2d(cin, cout, k, padding=k // 2)
    with torch.no_grad():
        conv.weight.copy_(rng.synth_input(f"dc.w{k}", conv.weight.shape, seed=80, scale=(cin * k * k) ** -0.5))
        conv.bias.copy_(rng.synth_input(f"dc.b{k}", (cout,), seed=80, scale=0.1))
    ad = DoRAConvAdapter(conv, rank=r, lora_alpha=16, lora_dropout=0.1)
    with torch.no_grad():
        ad.lora_A.copy_(rng.synth_input(f"dc.A{k}", ad.lora_A.shape, seed=80, scale=0.05))
        ad.lora_B.copy_(rng.synth_input(f"dc.B{k}", ad.lora_B.shape, seed=80, scale=0.2))
        ad.lora_magnitude_vector.mul_(1.0 + 0.2 * rng.synth_input(f"dc.m{k}", (cout,), seed=80).abs())
    c1 = cin // 3 * 2 // 8 * 8
    x = rng.synth_input(f"dc.x{k}", (B, cin, H, W), seed=80)
    rowb = rng.synth_input(f"dc.rb{k}", (B, cout), seed=80, scale=0.3)
    res = rng.synth_input(f"dc.res{k}", (B, cout, H, W), seed=80)
    cot = rng.synth_input(f"dc.cot{k}", (B, cout, H, W), seed=80)
    keep = (torch.rand((B, cin, H, W), generator=torch.Generator().manual_seed(9)) >= 0.1).float() / 0.9
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().half().to(dev)
    conv, ad = conv.to(dev), ad.to(dev).train()
    xh = nhwc(x)
    y, saved = dora_conv_fwd(conv, ad, xh[..., :c1].contiguous(), x2=xh[..., c1:].contiguous(), rowbias=rowb.half().to(dev),
                             residual=nhwc(res), mask=nhwc(keep))
    dx, grads = dora_conv_bwd(conv, ad, saved, nhwc(cot))
    # oracle (fp32, on the fp16-rounded inputs)
    xr = xh.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    P = {n: p.detach().float().cpu().requires_grad_(True) for n, p in ad.named_parameters()}
    ref = LO.dora_conv2d_train(xr, conv.weight.detach().float().cpu(), conv.bias.detach().float().cpu(), P["lora_A"], P["lora_B"],
                               P["lora_magnitude_vector"], 16 / r, nhwc(keep).float().cpu().permute(0, 3, 1, 2), 1, k // 2)
    ref = ref + rowb.half().float()[:, :, None, None] + nhwc(res).float().cpu().permute(0, 3, 1, 2)
    (ref * nhwc(cot).float().cpu().permute(0, 3, 1, 2)).sum().backward()
    tol = 5e-3
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.detach().numpy()) < tol
    assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2).numpy(), xr.grad.numpy()) < tol
    for n in ("lora_A", "lora_B", "lora_magnitude_vector"):
        e = rel_l2(grads[n].cpu().numpy(), P[n].grad.numpy())
        assert e < tol, (n, e)
    # eval mode / no mask: equals the merged-weight convolution
    from adaface_dev_amd.adaface.lora import dora_merged_weight
    y_eval, _ = dora_conv_fwd(conv, ad.eval(), xh)
    wm = dora_merged_weight(conv.weight.detach().cpu(), ad.lora_A.detach().cpu(), ad.lora_B.detach().cpu(),
                            ad.lora_magnitude_vector.detach().cpu(), 16 / r)
    ref_eval = F.conv2d(xh.float().cpu().permute(0, 3, 1, 2), wm, conv.bias.detach().float().cpu(), 1, k // 2)
    assert rel_l2(y_eval.float().cpu().permute(0, 3, 1, 2).numpy(), ref_eval.numpy()) < tol


@pytest.mark.parametrize("k", [3, 1])
def test_dora_saved_scales_survive_a_merged_pass_between_forward_and_backward(dev, k):
    """The cached DoRA scale vectors are refreshed IN PLACE when the base weight changes.  A gradient-free pass with merged inference adapters
    (adaface/lora.py::merge_unet_loras writes W' into conv.weight, runs, restores W) sits between the main pass's forward and its backward in a
    recon step; the backward must still see the scales of ITS forward (round-5 advisor finding: with lora_B != 0 the merged weight has another
    norm, and the backward's _packs() refreshed u / v / norm from it).  Compared with the same forward / backward with nothing in between."""
    from adaface_dev_amd import ops, rng
    from adaface_dev_amd.adaface.lora import dora_merged_weight
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import Conv2d, Linear
    from adaface_dev_amd.ldm.modules.dora import DoRAConvAdapter, DoRALinearAdapter, dora_conv_bwd, dora_conv_fwd, dora_linear
    cin, cout, r, B, H, W = 64, 64, 32, 2, 8, 8
    conv = Conv2d(cin, cout, k, padding=k // 2)
    with torch.no_grad():
        conv.weight.copy_(rng.synth_input(f"ds.w{k}", conv.weight.shape, seed=81, scale=(cin * k * k) ** -0.5))
        conv.bias.copy_(rng.synth_input(f"ds.b{k}", (cout,), seed=81, scale=0.1))
    ad = DoRAConvAdapter(conv, rank=r, lora_alpha=16, lora_dropout=0.0)
    with torch.no_grad():
        ad.lora_A.copy_(rng.synth_input(f"ds.A{k}", ad.lora_A.shape, seed=81, scale=0.2))
        ad.lora_B.copy_(rng.synth_input(f"ds.B{k}", ad.lora_B.shape, seed=81, scale=0.5))            # B != 0: ||W + s B A|| != ||W||
        ad.lora_magnitude_vector.mul_(1.0 + 0.2 * rng.synth_input(f"ds.m{k}", (cout,), seed=81).abs())
    conv, ad = conv.to(dev), ad.to(dev).train()
    x = rng.synth_input(f"ds.x{k}", (B, H, W, cin), seed=81).half().to(dev)
    cot = rng.synth_input(f"ds.cot{k}", (B, H, W, cout), seed=81).half().to(dev)

    def merged_pass():
        w0 = conv.weight.detach().clone()
        wm = dora_merged_weight(w0.cpu(), ad.lora_A.detach().cpu(), ad.lora_B.detach().cpu(), ad.lora_magnitude_vector.detach().cpu(), 16 / r)
        with torch.no_grad():
            conv.weight.copy_(wm.to(dev))                       # version bump 1: merged
            conv.hip(x)
            ad.scales(conv)                                     # what a pass that consults the cache meanwhile would do
            conv.weight.copy_(w0)                               # version bump 2: restored

    y0, saved0 = dora_conv_fwd(conv, ad, x)
    dx0, g0 = dora_conv_bwd(conv, ad, saved0, cot)
    y1, saved1 = dora_conv_fwd(conv, ad, x)
    merged_pass()
    dx1, g1 = dora_conv_bwd(conv, ad, saved1, cot)
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    # and the cache itself is right again after the restore
    u, v, norm = ad.scales(conv)
    ref_norm = (conv.weight.detach().float() + (16 / r) * (ad.lora_B.detach().flatten(1) @ ad.lora_A.detach().flatten(1)).reshape(conv.weight.shape)) \
        .flatten(1).norm(dim=1)
    assert rel_l2(norm.cpu().numpy(), ref_norm.cpu().numpy()) < 1e-3

    if k == 1:                                                  # the Linear adapter (attention DoRA) through autograd
        lin = Linear(cin, cout)
        with torch.no_grad():
            lin.weight.copy_(rng.synth_input("ds.lw", lin.weight.shape, seed=82, scale=cin ** -0.5))
        la = DoRALinearAdapter(lin, rank=r, lora_alpha=4, lora_dropout=0.0)
        with torch.no_grad():
            la.lora_A.copy_(rng.synth_input("ds.lA", la.lora_A.shape, seed=82, scale=0.2))
            la.lora_B.copy_(rng.synth_input("ds.lB", la.lora_B.shape, seed=82, scale=0.5))
        lin, la = lin.to(dev), la.to(dev).train()
        x2 = rng.synth_input("ds.lx", (96, cin), seed=82).half().to(dev)
        c2 = rng.synth_input("ds.lc", (96, cout), seed=82).half().to(dev)

        def run(between):
            for p in la.parameters():
                p.grad = None
            xr = x2.clone().requires_grad_(True)
            y = dora_linear(lin, la, xr)
            if between:
                w0 = lin.weight.detach().clone()
                with torch.no_grad():
                    lin.weight.mul_(1.5)
                    la.scales(lin)
                    lin.weight.copy_(w0)
            (y.float() * c2.float()).sum().backward()
            return [xr.grad.clone()] + [p.grad.clone() for p in la.parameters()]

        for a, b in zip(run(False), run(True)):
            assert torch.equal(a, b)


@pytest.mark.parametrize("rows,C", [(300, 72), (5000, 72), (24576, 320), (70000, 8)])
def test_colsum_small_and_tall_vs_torch(dev, rows, C):
    """Column sums (bias / LayerNorm gamma / DoRA magnitude gradients): the single-workgroup-per-64-columns form and the tall
    two-pass form (rows >= 2048), with and without the element-wise second operand and accumulation."""
    from adaface_dev_amd import ops
    g = torch.Generator().manual_seed(rows)
    a = torch.randn(rows, C, generator=g).half()
    b = torch.randn(rows, C, generator=g).half()
    out = ops.colsum(a.to(dev))
    assert rel_l2(out.cpu().numpy(), a.float().sum(0).numpy()) < 1e-4 + 2e-3 * (rows > 20000)
    acc = torch.ones(C, device=dev)
    ops.colsum(a.to(dev), b.to(dev), out=acc, accumulate=True)
    assert rel_l2(acc.cpu().numpy(), (1.0 + (a.float() * b.float()).sum(0)).numpy()) < 1e-4
    assert torch.equal(ops.colsum(a.to(dev), b.to(dev)), ops.colsum(a.to(dev), b.to(dev)))       # deterministic
