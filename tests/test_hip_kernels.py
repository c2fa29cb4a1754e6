"""Per-kernel parity on a real MI355X: every C-ABI entry point against a plain fp32 CPU
computation of the same op on the same (fp16-rounded) inputs.  `pytest -m gpu`.

Tolerances (stated per test): the kernels read fp16, accumulate in fp32 and round the result
to fp16 once, so the expected relative L2 error against an fp32 reference on identical inputs
is ~3e-4 (one fp16 rounding, 2^-11 = 4.9e-4 max per element); 2e-3 is used as the bound.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 2e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.float16)


# ------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,tile", [
    (256, 128, 64, 1), (256, 128, 64, 2), (1000, 320, 320, 0), (8, 1280, 320, 0), (616, 640, 768, 2),
    (4096, 320, 1280, 1), (130, 4, 320, 2), (512, 1280, 2560, 0),
])
def test_gemm_bias_residual(dev, M, N, K, tile):
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    b, r = torch.randn(N, generator=torch.Generator().manual_seed(3)), rnd((M, N), 4)
    pw = ops.pack_matrix(w, b, dev)
    out = ops.gemm(a.to(dev), pw, residual=r.to(dev), tile=tile)
    ref = a.float() @ w.float().t() + b + r.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("tile,splits", [(1, 2), (2, 3), (1, 8)])
def test_gemm_and_conv_split_k(dev, tile, splits):
    """split-K (fp32 partials + reduce/epilogue pass) == the single-pass result."""
    from adaface_dev_amd import ops
    M, N, K = 512, 192, 1280
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    b, r = torch.randn(N, generator=torch.Generator().manual_seed(3)), rnd((M, N), 4)
    out = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), residual=r.to(dev), act=ops.AF_ACT_NONE, tile=tile, splits=splits)
    ref = a.float() @ w.float().t() + b + r.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    B, H, W, cin, cout = 2, 8, 8, 192, 128
    x, wc = rnd((B, H, W, cin), 5), rnd((cout, cin, 3, 3), 6, (9 * cin) ** -0.5)
    bias, rowb = torch.randn(cout, generator=torch.Generator().manual_seed(7)), rnd((B, cout), 8)
    oc = ops.conv3x3(x.to(dev), ops.pack_conv3x3(wc, bias, dev), rowbias=rowb.to(dev), tile=tile, splits=splits)
    refc = F.conv2d(x.float().permute(0, 3, 1, 2), wc.float(), bias, padding=1) + rowb.float()[:, :, None, None]
    assert rel_l2(oc.float().cpu().permute(0, 3, 1, 2).numpy(), refc.numpy()) < TOL


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 7, 8, 11, 12, 13])
def test_gemm_fp32_output_and_wgrad_past_the_fp16_range(dev, tile):
    """AF_OUT_F32: the epilogue (or the split-K reduce pass) stores the fp32 accumulator.  Entries far beyond 65504 (a weight gradient summed over
    thousands of tokens) come back finite and exact to fp32 summation error; autograd_ops.wgrad uses this mode."""
    from adaface_dev_amd import ops
    from adaface_dev_amd.autograd_ops import wgrad
    M, N, K = 48, 320, 4096
    a, w = rnd((M, K), 1, 30.0), rnd((N, K), 2, 20.0)
    a[:, :64] = a[:, :64].abs()
    w[:, :64] = w[:, :64].abs()                                       # a coherent block: sums of ~64 * 600 on top of the random walk
    b = torch.randn(N, generator=torch.Generator().manual_seed(3))
    ref = a.double() @ w.double().t() + b.double()
    assert ref.abs().max() > 70000
    for splits in (1, 3):                              # unsplit: tiles 1 / 2 store fp32 themselves (3 .. 10 fall back to 1); split: the reduce pass does
        out = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), tile=tile, splits=splits, out_f32=True)
        assert out.dtype == torch.float32 and torch.isfinite(out).all()
        assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-5
    if tile == 0:
        for (T, n, k) in [(4096, 16, 128), (77, 64, 96), (300, 8, 320)]:     # tokens, dy columns, x columns (ragged, < 128 tokens too)
            dy, x = rnd((T, n), 5, 8.0), rnd((T, k), 6, 8.0)
            dy[:, 0], x[:, 0] = 40.0, 45.0                                    # dW[0, 0] = T * 1800: 7.4e6 at T = 4096
            dw = wgrad(dy.to(dev), x.to(dev))
            refw = dy.double().t() @ x.double()
            assert dw.dtype == torch.float32 and tuple(dw.shape) == (n, k) and torch.isfinite(dw).all()
            assert rel_l2(dw.cpu().numpy(), refw.numpy()) < 1e-5


@pytest.mark.parametrize("M,N,K,splits", [(256, 128, 64, 1), (1000, 320, 320, 1), (616, 640, 768, 2), (4096, 320, 1280, 1), (130, 4, 320, 1),
                                            (512, 1280, 2560, 4), (77, 64, 32, 1)])
def test_gemm_tile3_pipelined(dev, M, N, K, splits):
    """LDS-DMA ring kernel (tile 3): same results as the reference matmul, incl. ragged M/N, K tails and split-K."""
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    b, r = torch.randn(N, generator=torch.Generator().manual_seed(3)), rnd((M, N), 4)
    out = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), residual=r.to(dev), tile=3, splits=splits)
    ref = a.float() @ w.float().t() + b + r.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride,splits", [(2, 16, 16, 64, 0, 64, 1, 1), (2, 16, 16, 64, 0, 128, 1, 2), (2, 15, 17, 32, 0, 64, 2, 1),
                                                            (2, 8, 8, 128, 64, 64, 1, 1), (1, 32, 32, 320, 0, 4, 1, 1), (2, 8, 8, 96, 32, 128, 1, 3),
                                                            (1, 64, 64, 320, 0, 320, 1, 1)])
def test_conv3x3_tile3_pipelined(dev, B, H, W, c1, c2, cout, stride, splits):
    from adaface_dev_amd import ops
    cin = c1 + c2
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias, rowb = torch.randn(cout, generator=torch.Generator().manual_seed(4)), rnd((B, cout), 5)
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1) + rowb.float()[:, :, None, None]
    res = rnd(tuple(ref.permute(0, 2, 3, 1).shape), 6)
    ref = ref + res.float().permute(0, 3, 1, 2)
    out = ops.conv3x3(x1.to(dev), ops.pack_conv3x3(w, bias, dev), x2=None if x2 is None else x2.to(dev), stride=stride,
                      rowbias=rowb.to(dev), residual=res.to(dev), tile=3, splits=splits)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,H,W,cin,cout,splits,extras", [
    (1, 64, 64, 320, 320, 1, True), (2, 64, 64, 64, 160, 1, False), (2, 32, 32, 640, 640, 2, True), (3, 16, 16, 1280, 320, 4, True),
    (1, 32, 32, 192, 160, 3, False), (2, 16, 16, 128, 160, 1, True), (1, 96, 64, 128, 320, 1, True), (2, 8, 32, 64, 160, 1, False),
    (4, 8, 8, 128, 160, 1, True), (8, 8, 8, 1280, 320, 4, True), (12, 8, 8, 64, 160, 1, False)])          # the 8 x 8 level: four whole images per tile
def test_conv3x3_tile14_halo_resident(dev, B, H, W, cin, cout, splits, extras):
    """Halo-resident 3x3 kernel (tile 14): whole image rows per workgroup, nine taps off one LDS halo per 64-channel chunk; every level's
    width (64 / 32 / 16), image borders (zero padding from the zero page), several images per launch, split-K over channel chunks (incl.
    an uneven 3-way split of 3 chunks), bias / per-image row bias / residual epilogue -- against torch conv2d in fp32."""
    from adaface_dev_amd import ops
    x = rnd((B, H, W, cin), 1)
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(4))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1)
    rowb = res = None
    if extras:
        rowb = rnd((B, cout), 5)
        res = rnd((B, H, W, cout), 6)
        ref = ref + rowb.float()[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    out = ops.conv3x3(x.to(dev), ops.pack_conv3x3(w, bias, dev), rowbias=None if rowb is None else rowb.to(dev),
                      residual=None if res is None else res.to(dev), tile=14, splits=splits)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    # same launch against the tap-by-tap kernel: identical products, different summation order
    out7 = ops.conv3x3(x.to(dev), ops.pack_conv3x3(w, bias, dev), rowbias=None if rowb is None else rowb.to(dev),
                       residual=None if res is None else res.to(dev), tile=8, splits=1)
    assert rel_l2(out.float().cpu().numpy(), out7.float().cpu().numpy()) < 2e-3
    # round 6: the loop with its address arithmetic hoisted (tile 14) against the round-5 loop (tile 19): same stages, same MFMA order -> same bits
    out19 = ops.conv3x3(x.to(dev), ops.pack_conv3x3(w, bias, dev), rowbias=None if rowb is None else rowb.to(dev),
                        residual=None if res is None else res.to(dev), tile=19, splits=splits)
    assert torch.equal(out, out19)


@pytest.mark.parametrize("B,H,W,cin,cs1,cs2,cout,tile,splits", [
    (1, 64, 64, 320, 640, 320, 320, 7, 1), (2, 32, 32, 640, 640, 640, 640, 7, 2), (2, 16, 16, 1280, 1280, 1280, 1280, 7, 4), (2, 32, 32, 640, 320, 0, 640, 8, 1),
    (2, 16, 16, 128, 64, 64, 160, 11, 1), (1, 12, 20, 64, 128, 0, 128, 8, 3), (2, 16, 16, 128, 64, 64, 160, 13, 2), (2, 8, 8, 192, 64, 0, 128, 12, 1),
    (3, 8, 8, 1280, 1280, 1280, 1280, 8, 4),
    # round 6: the halo-resident kernel (tile 14) takes the K tail too -- its 3x3 part on the ping-pong loop, the tail as lock-step stages behind it, in the
    # last K split; chunk boundaries balanced over the splits (incl. a split that holds the tail only)
    (1, 64, 64, 320, 640, 320, 320, 14, 1), (2, 32, 32, 640, 640, 640, 640, 14, 2), (2, 16, 16, 1280, 1280, 1280, 1280, 14, 4), (4, 8, 8, 1280, 1280, 1280, 1280, 14, 4),
    (2, 16, 16, 128, 64, 64, 160, 14, 1), (2, 32, 32, 640, 320, 0, 640, 14, 3), (1, 16, 16, 64, 640, 640, 160, 14, 3), (1, 64, 64, 64, 64, 0, 160, 14, 1)])
def test_conv3x3_with_k_concatenated_1x1_skip(dev, B, H, W, cin, cs1, cs2, cout, tile, splits):
    """out = conv3x3(h) + conv1x1(cat(x1, x2)) as ONE launch (af_gemm_desc.a3 / a4: the ResBlock's second convolution with its channel-changing
    skip_connection K-concatenated behind the nine tap blocks, openaimodel.py:256-276) against torch in fp32 and against the two-launch form;
    one / two skip sources, every whole-line tile form, split-K, ragged M (image border rows and a tile that overhangs M)."""
    from adaface_dev_amd import ops
    h = rnd((B, H, W, cin), 1)
    x1 = rnd((B, H, W, cs1), 2)
    x2 = rnd((B, H, W, cs2), 3) if cs2 else None
    w3 = rnd((cout, cin, 3, 3), 4, (9 * cin) ** -0.5)
    w1 = rnd((cout, cs1 + cs2, 1, 1), 5, (cs1 + cs2) ** -0.5)
    b3, b1 = torch.randn(cout, generator=torch.Generator().manual_seed(6)), torch.randn(cout, generator=torch.Generator().manual_seed(7))
    xs = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    ref = F.conv2d(h.float().permute(0, 3, 1, 2), w3.float(), b3, padding=1) + F.conv2d(xs, w1.float(), b1)
    pw = ops.pack_conv3x3_skip(w3, b3, w1, b1, dev)
    assert pw.K == 9 * cin + cs1 + cs2 and pw.k_tail == cs1 + cs2
    x2d = None if x2 is None else x2.to(dev)
    out = ops.conv3x3(h.to(dev), pw, skip=(x1.to(dev), x2d), tile=tile, splits=splits)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    # the table-driven choice (no explicit tile) lands on a whole-line tile too
    out_auto = ops.conv3x3(h.to(dev), pw, skip=(x1.to(dev), x2d))
    assert rel_l2(out_auto.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    # two launches: 1x1 GEMM, then the 3x3 with it as residual (one more fp16 rounding)
    sk = ops.gemm(x1.to(dev).reshape(B * H * W, cs1), ops.pack_matrix(w1.reshape(cout, cs1 + cs2), b1, dev),
                  a2=None if x2 is None else x2d.reshape(B * H * W, cs2))
    out2 = ops.conv3x3(h.to(dev), ops.pack_conv3x3(w3, b3, dev), residual=sk.reshape(B, H, W, cout), tile=8)
    assert rel_l2(out.float().cpu().numpy(), out2.float().cpu().numpy()) < 2e-3
    for _ in range(5):                                   # repeated launches: bit-stable (the hand-offs between the loops are ordered by barriers)
        assert torch.equal(ops.conv3x3(h.to(dev), pw, skip=(x1.to(dev), x2d), tile=tile, splits=splits), out)
    # outside the whole-line tiles the descriptor is refused, never silently mis-computed
    with pytest.raises(RuntimeError, match="K tail"):
        ops.conv3x3(h.to(dev), pw, skip=(x1.to(dev), x2d), tile=2)


@pytest.mark.parametrize("B,H,W,c1,c2,cout,splits", [(1, 64, 64, 320, 320, 320, 1), (2, 32, 32, 640, 320, 640, 2), (2, 16, 16, 64, 128, 160, 3), (1, 16, 16, 1280, 1280, 320, 4),
                                                     (3, 32, 32, 64, 64, 160, 1)])
def test_conv3x3_tile14_two_sources(dev, B, H, W, c1, c2, cout, splits):
    """The halo-resident kernel on a channel-concatenated input (the decoder's ResBlocks: h | skip): a 64-channel chunk of the halo comes from ONE
    of the two tensors (their own pixel strides), the weight columns stay in (tap, concatenated channel) order; split-K ranges that cut inside
    either source; repeated launches are bit-identical (the ping-pong loop's hand-offs are ordered by barriers, not by luck)."""
    from adaface_dev_amd import ops
    x1, x2 = rnd((B, H, W, c1), 1), rnd((B, H, W, c2), 2)
    w = rnd((cout, c1 + c2, 3, 3), 3, (9 * (c1 + c2)) ** -0.5)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(4))
    rowb, res = rnd((B, cout), 5), rnd((B, H, W, cout), 6)
    ref = F.conv2d(torch.cat([x1, x2], -1).float().permute(0, 3, 1, 2), w.float(), bias, padding=1) + rowb.float()[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    pw = ops.pack_conv3x3(w, bias, dev)
    run = lambda tile=14: ops.conv3x3(x1.to(dev), pw, x2=x2.to(dev), rowbias=rowb.to(dev), residual=res.to(dev), tile=tile, splits=splits)
    out = run()
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    for _ in range(10):
        assert torch.equal(run(), out)
    assert torch.equal(run(19), out)               # the round-5 loop (tile 19): bit-identical


@pytest.mark.parametrize("B,H,W,c1,c2,cout,splits", [(1, 32, 32, 640, 0, 640, 1), (2, 16, 16, 1280, 0, 320, 2), (2, 8, 8, 128, 64, 160, 3), (1, 8, 32, 64, 0, 160, 1)])
def test_conv3x3_tile14_nearest_x2_upsample(dev, B, H, W, c1, c2, cout, splits):
    """The halo-resident kernel on the nearest-x2 upsampled input (the U-Net's Upsample convolutions): halo pixel (y, x) of the 2H x 2W grid is source
    pixel (y >> 1, x >> 1), out-of-grid pixels are zero -- against torch interpolate + conv2d, and bit-stable over repeated launches."""
    from adaface_dev_amd import ops
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    w = rnd((cout, c1 + c2, 3, 3), 3, (9 * (c1 + c2)) ** -0.5)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(4))
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    ref = F.conv2d(F.interpolate(xin, scale_factor=2, mode="nearest"), w.float(), bias, padding=1)
    pw = ops.pack_conv3x3(w, bias, dev)
    run = lambda tile, sp: ops.conv3x3(x1.to(dev), pw, x2=None if x2 is None else x2.to(dev), upsample=True, tile=tile, splits=sp)
    out = run(14, splits)
    assert out.shape == (B, 2 * H, 2 * W, cout)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    assert rel_l2(out.float().cpu().numpy(), run(8, 1).float().cpu().numpy()) < 2e-3
    for _ in range(5):
        assert torch.equal(run(14, splits), out)
    assert torch.equal(run(19, splits), out)       # the round-5 loop (tile 19): bit-identical


@pytest.mark.parametrize("kind,B,H,W,cin,cout,tile", [("conv", 2, 16, 16, 128, 320, 7), ("conv", 1, 64, 64, 320, 320, 7), ("conv", 2, 32, 32, 64, 640, 7),
                                                       ("conv", 3, 16, 8, 64, 320, 11), ("gemm", 2, 16, 16, 320, 320, 7), ("gemm", 2, 32, 32, 128, 1280, 11),
                                                       ("skip", 2, 16, 16, 128, 320, 7), ("conv", 2, 16, 16, 64, 960, 7),
                                                       ("conv", 1, 64, 64, 320, 320, 14), ("conv", 2, 32, 32, 64, 640, 14), ("conv", 2, 16, 16, 128, 320, 14)])
def test_groupnorm_statistics_from_the_producing_gemm(dev, kind, B, H, W, cin, cout, tile):
    """A convolution / 1x1 GEMM / K-concatenated convolution whose output feeds a GroupNorm(32) leaves the partial (sum, sumsq) of the fp16 values
    it stores (af_gemm_desc.gn_partials); af_groupnorm_apply then normalises without a statistics pass.  Same result as the two-pass GroupNorm
    on the same tensor (identical statistics up to summation order) and as torch's group_norm in fp32 (util.py:195-212)."""
    from adaface_dev_amd import ops
    x = rnd((B, H, W, cin), 1)
    gam = torch.randn(cout, generator=torch.Generator().manual_seed(2)) * 0.3 + 1
    bet = torch.randn(cout, generator=torch.Generator().manual_seed(3)) * 0.3
    cpg = cout // 32
    if kind == "conv":
        pw = ops.pack_conv3x3(rnd((cout, cin, 3, 3), 4, (9 * cin) ** -0.5), torch.randn(cout, generator=torch.Generator().manual_seed(5)), dev)
        y = ops.conv3x3(x.to(dev), pw, rowbias=rnd((B, cout), 6).to(dev), tile=tile, gn_cpg=cpg)
    elif kind == "skip":
        pw = ops.pack_conv3x3_skip(rnd((cout, cin, 3, 3), 4, (9 * cin) ** -0.5), None, rnd((cout, 64, 1, 1), 7, 0.1), None, dev)
        y = ops.conv3x3(x.to(dev), pw, skip=(rnd((B, H, W, 64), 8).to(dev), None), tile=tile, gn_cpg=cpg)
    else:
        pw = ops.pack_matrix(rnd((cout, cin), 4, cin ** -0.5), None, dev)
        y = ops.gemm(x.to(dev).reshape(B * H * W, cin), pw, residual=rnd((B * H * W, cout), 9).to(dev), rows_per_batch=H * W, tile=tile, gn_cpg=cpg)
        y4 = y.reshape(B, H, W, cout)
        y4._gn_partials = y._gn_partials
        y = y4
    gn = getattr(y, "_gn_partials", None)
    if (320 if tile == 7 else 160) % cpg != 0:          # groups that straddle tiles (C = 960: 30 channels per group): no statistics, plain launch
        assert gn is None
        return
    assert gn is not None and gn.nblk == H * W // 128 and gn.cpg == cpg, "the launch was expected to leave its statistics"
    # the partials themselves: per (batch item, 128-row block, group) the sum of the stored fp16 values and M2 = the sum of squares about the
    # BLOCK'S OWN mean (round 5: merged pairwise by the consumer, no E[x^2] - mean^2 anywhere); the halo-resident kernel's 256-row tile leaves
    # its two 128-row blocks separately
    # The producers take the sums shifted by a pivot in packed fp16 (d = x - pivot rounded to 11 bits when x is not within a factor 2 of the
    # pivot, exact when it is): a block's mean is good to ~1e-4 of the block's spread, its variance to ~1e-3 relative -- checked in those units.
    yf = y.double().reshape(B, H * W // 128, 128, 32, cpg)
    n = 128 * cpg
    want_mean, want_var = yf.mean(dim=(2, 4)), yf.var(dim=(2, 4), unbiased=False)
    got = gn.ws[:, :gn.nblk].double().cpu()
    got_mean, got_var = got[..., 0] / n, got[..., 1] / n
    assert float(((got_mean - want_mean.cpu()).abs() / want_var.cpu().sqrt()).max()) < 5e-4
    assert float(((got_var - want_var.cpu()).abs() / want_var.cpu()).max()) < 2e-3
    for silu in (False, True):
        out = ops.groupnorm(y, gam.to(dev), bet.to(dev), 1e-5, silu)                       # picks the partials up
        y_plain = y.clone()                                                                  # no partials attached: statistics pass + normalise
        out2 = ops.groupnorm(y_plain, gam.to(dev), bet.to(dev), 1e-5, silu)
        ref = F.group_norm(y.float().cpu().permute(0, 3, 1, 2), 32, gam, bet, 1e-5)
        ref = F.silu(ref) if silu else ref
        assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
        assert rel_l2(out.float().cpu().numpy(), out2.float().cpu().numpy()) < 1e-3
    # outside the scope (a 128-wide tile, split-K) no statistics are left and the launch is unchanged
    if kind == "conv":
        y8 = ops.conv3x3(x.to(dev), pw, rowbias=rnd((B, cout), 6).to(dev), tile=8, gn_cpg=cpg)
        assert getattr(y8, "_gn_partials", None) is None and rel_l2(y8.float().cpu().numpy(), y.float().cpu().numpy()) < 2e-3


def test_groupnorm_drops_producer_statistics_after_an_in_place_write(dev):
    """The partial statistics ride on the tensor OBJECT (ops.GnPartials); they describe the bytes the producing launch stored.  An in-place write
    between the producer and the GroupNorm must not be normalised with stale statistics: ops.partials_of compares the tensor's version counter and
    storage address with the ones recorded at production and the consumer falls back to its own statistics pass."""
    from adaface_dev_amd import ops
    B, H, W, cin, cout = 2, 16, 16, 128, 320
    x = rnd((B, H, W, cin), 1)
    pw = ops.pack_conv3x3(rnd((cout, cin, 3, 3), 4, (9 * cin) ** -0.5), None, dev)
    gam, bet = torch.ones(cout), torch.zeros(cout)
    y = ops.conv3x3(x.to(dev), pw, tile=7, gn_cpg=cout // 32)
    assert ops.partials_of(y) is not None
    view = y.reshape(B, H * W, cout)
    view._gn_partials = y._gn_partials
    assert ops.partials_of(view) is not None, "a view shares storage address and version counter"
    fresh = ops.groupnorm(y, gam.to(dev), bet.to(dev), 1e-5, False)
    y.add_(3.0)                                          # bumps y._version (and the view's)
    y[:, : H // 2].mul_(4.0)                             # (not a per-group affine map, which GroupNorm would undo: half of the PIXELS)
    assert ops.partials_of(y) is None and ops.partials_of(view) is None
    for t in (y, view):
        out = ops.groupnorm(t, gam.to(dev), bet.to(dev), 1e-5, False).reshape(B, H, W, cout)
        ref = F.group_norm(y.float().cpu().permute(0, 3, 1, 2), 32, gam, bet, 1e-5).permute(0, 2, 3, 1)
        assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    assert rel_l2(out.float().cpu().numpy(), fresh.float().cpu().numpy()) > 0.05, "the write must have changed the normalised tensor"
    assert ops.gn_proj_fused(y, gam.to(dev), bet.to(dev), 1e-6, ops.pack_matrix(rnd((320, 320), 5, 0.05), None, dev)) is None


def test_conv3x3_statistics_request_on_a_geometry_outside_tile14(dev):
    """The tuned table's key has no image geometry: `9,24576,320,2880` (tile 14, the halo-resident 3x3) is the key of a batch-6 64 x 64 latent AND of a
    batch-4 64 x 96 one, which tile 14 cannot take (W not in 16 / 32 / 64).  The library falls back to a tap-by-tap tile for it; asking that launch
    for GroupNorm statistics used to fail with AF_E_UNSUPPORTED (round-4 advisor finding).  The request is now made only where tile 14 will run."""
    from adaface_dev_amd import ops
    B, H, W, c = 4, 64, 96, 320
    assert ops.tune_table().get(f"9,{B * H * W},{c},{9 * c},0,0,1,0", (0, 1))[0] == 14, "the collision this test is about needs the table's tile-14 entry"
    x, w = rnd((B, H, W, c), 1), rnd((c, c, 3, 3), 2, (9 * c) ** -0.5)
    y = ops.conv3x3(x.to(dev), ops.pack_conv3x3(w, None, dev), gn_cpg=c // 32)          # tile / splits from the table
    assert ops.partials_of(y) is None
    ref = F.conv2d(x.float().permute(0, 3, 1, 2)[:1], w.float(), None, padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y[:1].float().cpu().numpy(), ref.numpy()) < TOL
    out = ops.groupnorm(y, torch.ones(c, device=dev), torch.zeros(c, device=dev), 1e-5, True)
    gref = F.silu(F.group_norm(y[:1].float().cpu().permute(0, 3, 1, 2), 32, None, None, 1e-5)).permute(0, 2, 3, 1)
    assert rel_l2(out[:1].float().cpu().numpy(), gref.numpy()) < TOL


def test_conv3x3_with_skip_on_a_geometry_outside_tile14_takes_a_tap_by_tap_tile(dev):
    """Round 6: the table gives the conv2 + shortcut launches of the 64 x 64 level to the halo-resident kernel (tile 14, key `9,32768,320,3520`: no image
    geometry in it).  The same key at a latent tile 14 cannot take (8 x 32 x 128: W = 128) must not reach the library's register-staged fallback, which has
    no K tail: ops.conv3x3 picks a whole-line tap-by-tap tile there."""
    from adaface_dev_amd import ops
    B, H, W, cin, cs, cout = 8, 32, 128, 320, 640, 320
    assert ops.tune_table().get(f"9,{B * H * W},{cout},{9 * cin + cs},0,0,1,0", (0, 1))[0] == 14, "this test is about the table's tile-14 entry for the tail key"
    h, x1 = rnd((B, H, W, cin), 1), rnd((B, H, W, cs), 2)
    w3, w1 = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5), rnd((cout, cs, 1, 1), 4, cs ** -0.5)
    pw = ops.pack_conv3x3_skip(w3, None, w1, None, dev)
    out = ops.conv3x3(h.to(dev), pw, skip=(x1.to(dev), None), gn_cpg=cout // 32)
    ref = F.conv2d(h.float().permute(0, 3, 1, 2)[:1], w3.float(), None, padding=1) + F.conv2d(x1.float().permute(0, 3, 1, 2)[:1], w1.float())
    assert rel_l2(out[:1].float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    # and on the geometry tile 14 takes, the table's choice runs there and leaves the GroupNorm statistics
    h2, x2 = rnd((8, 64, 64, cin), 5), rnd((8, 64, 64, cs), 6)
    out2 = ops.conv3x3(h2.to(dev), pw, skip=(x2.to(dev), None), gn_cpg=cout // 32)
    assert ops.partials_of(out2) is not None
    ref2 = F.conv2d(h2.float().permute(0, 3, 1, 2)[:1], w3.float(), None, padding=1) + F.conv2d(x2.float().permute(0, 3, 1, 2)[:1], w1.float())
    assert rel_l2(out2[:1].float().cpu().permute(0, 3, 1, 2).numpy(), ref2.numpy()) < TOL
    gn = ops.groupnorm(out2, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), 1e-5, True)
    gref = F.silu(F.group_norm(out2[:1].float().cpu().permute(0, 3, 1, 2), 32, None, None, 1e-5)).permute(0, 2, 3, 1)
    assert rel_l2(gn[:1].float().cpu().numpy(), gref.numpy()) < TOL


@pytest.mark.parametrize("B,H,W,with_bias", [(2, 16, 16, True), (1, 64, 64, True), (3, 16, 8, False), (8, 32, 32, True)])
def test_groupnorm_proj_fused_c320(dev, B, H, W, with_bias):
    """af_gn_proj_fused: GroupNorm(32, eps 1e-6) of a tensor that carries its producer's partial statistics + the 1x1 convolution behind it
    (SpatialTransformer norm -> proj_in, attention.py:283-291) in one launch, against torch in fp32 and against the two-launch form."""
    from adaface_dev_amd import ops
    C = 320
    x0 = rnd((B, H, W, C), 1)
    pwc = ops.pack_conv3x3(rnd((C, C, 3, 3), 2, (9 * C) ** -0.5), torch.randn(C, generator=torch.Generator().manual_seed(3)), dev)
    x = ops.conv3x3(x0.to(dev), pwc, tile=7, gn_cpg=C // 32)                      # the producer leaves the partials
    assert getattr(x, "_gn_partials", None) is not None
    gam = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.3 + 1
    bet = torch.randn(C, generator=torch.Generator().manual_seed(5)) * 0.3
    w = rnd((C, C), 6, C ** -0.5)
    b = torch.randn(C, generator=torch.Generator().manual_seed(7)) if with_bias else None
    pw = ops.pack_matrix(w, b, dev)
    out = ops.gn_proj_fused(x, gam.to(dev), bet.to(dev), 1e-6, pw)
    assert out is not None and out.shape == (B * H * W, C)
    xn = F.group_norm(x.float().cpu().permute(0, 3, 1, 2), 32, gam, bet, 1e-6).permute(0, 2, 3, 1).reshape(B * H * W, C)
    ref = xn @ w.float().t() + (b if b is not None else 0)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    y = ops.groupnorm(x, gam.to(dev), bet.to(dev), 1e-6, False)
    out2 = ops.gemm(y.reshape(B * H * W, C), pw)
    assert rel_l2(out.float().cpu().numpy(), out2.float().cpu().numpy()) < 2e-3
    # a tensor without partials (or another width) is not taken: the caller falls back
    assert ops.gn_proj_fused(x.clone(), gam.to(dev), bet.to(dev), 1e-6, pw) is None


@pytest.mark.parametrize("B,H,W,c1,c2,cout,ups,splits", [(1, 64, 64, 128, 0, 128, False, 1), (2, 32, 32, 256, 0, 256, False, 2), (1, 32, 32, 128, 0, 128, True, 1),
                                                         (1, 24, 40, 64, 64, 128, False, 1), (2, 16, 16, 512, 0, 384, False, 3)])
def test_conv3x3_and_gemm_tile15_256x128(dev, B, H, W, c1, c2, cout, ups, splits):
    """Tile 15 of the whole-line kernel (256 x 128, eight waves as 4 x 2: narrow outputs over many rows, the VAE decoder's convolutions): 3x3 with
    bias / row bias / residual, two sources, nearest-x2 upsampling folded into the gather, split-K, M not a multiple of 256; and a plain GEMM."""
    from adaface_dev_amd import ops
    cin = c1 + c2
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias, rowb = torch.randn(cout, generator=torch.Generator().manual_seed(4)), rnd((B, cout), 5)
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias, padding=1) + rowb.float()[:, :, None, None]
    res = rnd(tuple(ref.permute(0, 2, 3, 1).shape), 6)
    ref = ref + res.float().permute(0, 3, 1, 2)
    out = ops.conv3x3(x1.to(dev), ops.pack_conv3x3(w, bias, dev), x2=None if x2 is None else x2.to(dev), upsample=ups, rowbias=rowb.to(dev),
                      residual=res.to(dev), tile=15, splits=splits)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    M, K = B * H * W + 24, cin
    a, wm = rnd((M, K), 7), rnd((cout, K), 8, K ** -0.5)
    og = ops.gemm(a.to(dev), ops.pack_matrix(wm, bias, dev), act=ops.AF_ACT_SILU, tile=15)
    assert rel_l2(og.float().cpu().numpy(), F.silu(a.float() @ wm.float().t() + bias).numpy()) < TOL


@pytest.mark.parametrize("M,N,K", [(388, 768, 768), (388, 3072, 768), (1552, 768, 3072), (77, 192, 320), (256, 1280, 1280), (64, 320, 192), (1, 512, 512),
                                   (8, 1280, 320), (130, 4, 320), (33, 100, 72), (1000, 320, 328)])
@pytest.mark.parametrize("mode", ["plain", "epilogue", "f32", "view"])
def test_gemm_tile18_small_direct(dev, M, N, K, mode):
    """Tile 18 (round 5): small plain GEMMs with their MFMA fragments straight from global memory, 32 x 64 outputs per workgroup, no LDS, no split-K --
    the CLIP linears at 388 / 1552 tokens and their dgrad / wgrad shapes, rank-192 adapter projections, batch-1 projections, one-row and
    eight-row cases, K that is not a multiple of 32 / 64 (zero columns of the packed weight against a clamped re-read of A), N that is not a multiple
    of 16; with bias + per-batch row bias + SiLU / quick-GELU + residual; the fp32 accumulator as output (weight gradients); an A operand that is a
    column slice of a wider buffer (lda > K) -- against the fp32 product on the same fp16 operands, and against tile 2 on the same call."""
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    pw = ops.pack_matrix(w, torch.randn(N, generator=torch.Generator().manual_seed(3)) if mode == "epilogue" else None, dev)
    kw, ref = {}, a.float() @ w.float().t()
    ad = a.to(dev)
    if mode == "epilogue":
        rpb = -(-M // 3)
        rowb, res = rnd((3, N), 6), rnd((M, N), 7)
        kw = dict(rowbias=rowb.to(dev), rows_per_batch=rpb, residual=res.to(dev), act=ops.AF_ACT_QUICKGELU if M % 2 else ops.AF_ACT_SILU)
        z = ref + pw.bias.cpu() + rowb.float()[torch.arange(M) // rpb]
        ref = (z * torch.sigmoid(1.702 * z) if M % 2 else F.silu(z)) + res.float()
    if mode == "f32":
        kw = dict(out_f32=True)
    if mode == "view":
        wide = rnd((M, K + 24), 5).to(dev)
        wide[:, 8:8 + K] = ad
        ad = wide[:, 8:8 + K]                              # a column slice: rows 16-byte aligned, lda = K + 24 (the C ABI as the strided callers use it)
        d = ops.GemmDesc()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        d.a1, d.wt, d.out = ad.data_ptr(), pw.wt.data_ptr(), out.data_ptr()
        d.M, d.N, d.K, d.kpad, d.taps, d.c1, d.lda1 = M, N, K, pw.kpad, 1, K, ad.stride(0)
        ops._launch_gemm(d, dev, "af_gemm", 18, 1)
    else:
        out = ops.gemm(ad, pw, tile=18, **kw)
    assert out.dtype == (torch.float32 if mode == "f32" else torch.float16)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < (1e-5 if mode == "f32" else TOL)
    if mode in ("plain", "epilogue"):
        out2 = ops.gemm(a.to(dev), pw, tile=2, **kw)
        assert rel_l2(out.float().cpu().numpy(), out2.float().cpu().numpy()) < 1e-3


@pytest.mark.parametrize("tile", [16, 17])
@pytest.mark.parametrize("M,N,K,splits,ln", [(8192, 640, 640, 1, False), (2048 + 40, 1280, 1280, 1, True), (512, 1280, 1280, 3, False), (4096, 320, 320, 1, True), (100, 128, 192, 1, False)])
def test_gemm_and_conv_small_tiles_16_17(dev, tile, M, N, K, splits, ln):
    """Tiles 16 / 17 of the whole-line kernel (64 x 128 / 128 x 64, four waves, three workgroups per CU): plain GEMM with bias / SiLU / residual,
    split-K, M not a multiple of the tile, the folded LayerNorm; and a 3x3 convolution with two sources."""
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    res = rnd((M, N), 4)
    if N % (128 if tile == 16 else 64) != 0:            # outside the tile's scope: a folded LayerNorm is refused (nothing else could take it), a plain GEMM falls back
        with pytest.raises(RuntimeError, match="folded LayerNorm"):
            ops.gemm(a.to(dev), ops.pack_matrix_ln(w, bias, torch.ones(K), torch.zeros(K), 1e-5, dev), tile=tile)
        return
    if ln:
        gam = torch.randn(K, generator=torch.Generator().manual_seed(5)) * 0.2 + 1
        bet = torch.randn(K, generator=torch.Generator().manual_seed(6)) * 0.2
        out = ops.gemm(a.to(dev), ops.pack_matrix_ln(w, bias, gam, bet, 1e-5, dev), residual=res.to(dev), tile=tile)
        ref = F.layer_norm(a.float(), (K,), gam, bet, 1e-5) @ w.float().t() + bias + res.float()
    else:
        out = ops.gemm(a.to(dev), ops.pack_matrix(w, bias, dev), act=ops.AF_ACT_SILU, residual=res.to(dev), tile=tile, splits=splits)
        ref = F.silu(a.float() @ w.float().t() + bias) + res.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    if M == 8192:
        B, H, W, c1, c2 = 2, 16, 16, 64, 64
        x1, x2 = rnd((B, H, W, c1), 7), rnd((B, H, W, c2), 8)
        wc = rnd((N, c1 + c2, 3, 3), 9, (9 * (c1 + c2)) ** -0.5)
        rowb = rnd((B, N), 10)
        y = ops.conv3x3(x1.to(dev), ops.pack_conv3x3(wc, bias, dev), x2=x2.to(dev), rowbias=rowb.to(dev), tile=tile)
        refc = F.conv2d(torch.cat([x1, x2], -1).float().permute(0, 3, 1, 2), wc.float(), bias, padding=1) + rowb.float()[:, :, None, None]
        assert rel_l2(y.float().cpu().permute(0, 3, 1, 2).numpy(), refc.numpy()) < TOL


def test_conv3x3_tile14_falls_back_outside_its_scope(dev):
    """stride 2 / a second source that is not a multiple of 64 channels / widths it does not take: the descriptor's fallback (tile 1) computes the same convolution."""
    from adaface_dev_amd import ops
    for (B, H, W, c1, c2, cout, stride) in ((1, 16, 16, 64, 0, 160, 2), (1, 16, 16, 64, 32, 160, 1), (1, 12, 24, 64, 0, 160, 1)):
        x1, x2 = rnd((B, H, W, c1), 1), (rnd((B, H, W, c2), 2) if c2 else None)
        w = rnd((cout, c1 + c2, 3, 3), 3, (9 * (c1 + c2)) ** -0.5)
        xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
        ref = F.conv2d(xin, w.float(), None, stride=stride, padding=1)
        out = ops.conv3x3(x1.to(dev), ops.pack_conv3x3(w, None, dev), x2=None if x2 is None else x2.to(dev), stride=stride, tile=14)
        assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("M,N,K,splits", [(300, 320, 320, 1), (4096, 640, 1280, 2), (130, 960, 64, 1)])
def test_gemm_tile4_wide(dev, M, N, K, splits):
    """128 x 320 / 8-wave variant of the LDS-DMA ring kernel (N % 320 == 0)."""
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    b, r = torch.randn(N, generator=torch.Generator().manual_seed(3)), rnd((M, N), 4)
    out = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), residual=r.to(dev), act=ops.AF_ACT_SILU * 0, tile=4, splits=splits)
    ref = a.float() @ w.float().t() + b + r.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride,splits", [(1, 32, 32, 320, 0, 320, 1, 1), (2, 16, 16, 640, 320, 640, 1, 2), (2, 16, 16, 320, 0, 320, 2, 1)])
def test_conv3x3_tile4_wide(dev, B, H, W, c1, c2, cout, stride, splits):
    from adaface_dev_amd import ops
    cin = c1 + c2
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias, rowb = torch.randn(cout, generator=torch.Generator().manual_seed(4)), rnd((B, cout), 5)
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1) + rowb.float()[:, :, None, None]
    res = rnd(tuple(ref.permute(0, 2, 3, 1).shape), 6)
    ref = ref + res.float().permute(0, 3, 1, 2)
    out = ops.conv3x3(x1.to(dev), ops.pack_conv3x3(w, bias, dev), x2=None if x2 is None else x2.to(dev), stride=stride,
                      rowbias=rowb.to(dev), residual=res.to(dev), tile=4, splits=splits)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("M,N,K,tile,act", [(700, 1280, 320, 5, 0), (700, 1280, 320, 6, 0), (300, 2560, 192, 5, 2), (520, 2560, 320, 6, 2),
                                             (256, 640, 64, 6, 0)])
def test_gemm_256_row_tiles(dev, M, N, K, tile, act):
    """256 x 256 (tile 5) / 256 x 320 (tile 6) ring tiles: plain GEMM with bias + residual (staged epilogue) and GEGLU, ragged M."""
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)) * 0.1
    if act == 2:
        wi, bi = ops.interleave_geglu(w.float(), b)
        out = ops.gemm(a.to(dev), ops.pack_matrix(wi, bi, dev), act=ops.AF_ACT_GEGLU, tile=tile)
        x, g = (a.float() @ w.float().t() + b).chunk(2, dim=-1)
        ref = x * F.gelu(g)
    else:
        r = rnd((M, N), 4)
        out = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), residual=r.to(dev), tile=tile)
        ref = a.float() @ w.float().t() + b + r.float()
    assert out.shape == ref.shape and rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("M,N,K,splits,tile", [(300, 320, 320, 1, 7), (4096, 640, 1280, 2, 7), (130, 960, 64, 1, 7), (1000, 320, 2880, 3, 7),
                                                (300, 200, 320, 1, 8), (1000, 384, 1280, 2, 8), (77, 768, 64, 1, 8),
                                                (300, 320, 320, 1, 11), (4096, 640, 1280, 2, 11), (130, 960, 64, 1, 11), (1000, 160, 2880, 3, 11), (77, 1280, 64, 1, 11),
                                                (300, 200, 320, 1, 12), (1000, 384, 1280, 2, 12), (77, 768, 64, 1, 12), (2048, 1280, 128, 1, 12), (130, 256, 192, 1, 12),
                                                (300, 320, 320, 1, 13), (4096, 640, 1280, 2, 13), (130, 960, 64, 1, 13), (1000, 160, 2880, 3, 13), (77, 1280, 128, 1, 13)])
def test_gemm_whole_line_tiles(dev, M, N, K, splits, tile):
    """Whole-line kernel (64-wide K stages, whole-cache-line LDS-DMA pieces, two slots): 128 x 320 (tile 7), 128 x 128 (tile 8) and
    128 x 160 (tile 11: four waves, two workgroups per CU; N % 160 == 0); tiles 12 / 13 = the 128 x 128 / 128 x 160 tiles with a FOUR-slot
    ring (three stages in flight; K of one, two and three stages exercise the short-ring waits); ragged M and N, split-K."""
    from adaface_dev_amd import ops
    a, w = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    b, r = torch.randn(N, generator=torch.Generator().manual_seed(3)), rnd((M, N), 4)
    out = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), residual=r.to(dev), tile=tile, splits=splits)
    ref = a.float() @ w.float().t() + b + r.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride,splits,tile", [(1, 32, 32, 320, 0, 320, 1, 1, 7), (2, 16, 16, 640, 320, 640, 1, 2, 7),
                                                                (2, 16, 16, 320, 0, 320, 2, 1, 7), (3, 8, 8, 1280, 1280, 1280, 1, 4, 7),
                                                                (2, 16, 16, 128, 64, 192, 1, 1, 8), (1, 12, 20, 64, 0, 72, 2, 2, 8),
                                                                (1, 32, 32, 320, 0, 320, 1, 1, 11), (2, 16, 16, 640, 320, 640, 1, 2, 11),
                                                                (2, 16, 16, 320, 0, 320, 2, 1, 11), (3, 8, 8, 1280, 1280, 1280, 1, 4, 11), (1, 12, 20, 64, 0, 160, 2, 2, 11),
                                                                (2, 16, 16, 128, 64, 192, 1, 1, 12), (1, 12, 20, 64, 0, 72, 2, 2, 12), (3, 8, 8, 1280, 1280, 1280, 1, 4, 12),
                                                                (1, 32, 32, 320, 0, 320, 1, 1, 13), (2, 16, 16, 640, 320, 640, 1, 2, 13), (3, 8, 8, 1280, 1280, 1280, 1, 4, 13)])
def test_conv3x3_whole_line_tiles(dev, B, H, W, c1, c2, cout, stride, splits, tile):
    from adaface_dev_amd import ops
    cin = c1 + c2
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias, rowb = torch.randn(cout, generator=torch.Generator().manual_seed(4)), rnd((B, cout), 5)
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1) + rowb.float()[:, :, None, None]
    res = rnd(tuple(ref.permute(0, 2, 3, 1).shape), 6)
    ref = ref + res.float().permute(0, 3, 1, 2)
    out = ops.conv3x3(x1.to(dev), ops.pack_conv3x3(w, bias, dev), x2=None if x2 is None else x2.to(dev), stride=stride,
                      rowbias=rowb.to(dev), residual=res.to(dev), tile=tile, splits=splits)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


def test_gemm_concat_k_and_silu(dev):
    from adaface_dev_amd import ops
    M, K1, K2, N = 300, 128, 64, 192
    a1, a2, w = rnd((M, K1), 1), rnd((M, K2), 2), rnd((N, K1 + K2), 3, 0.1)
    b = torch.randn(N, generator=torch.Generator().manual_seed(5))
    pw = ops.pack_matrix(w, b, dev)
    out = ops.gemm(a1.to(dev), pw, a2=a2.to(dev), act=ops.AF_ACT_SILU)
    ref = F.silu(torch.cat([a1, a2], 1).float() @ w.float().t() + b)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("tile,tokens,pad", [(4, 100, 0), (7, 100, 0), (7, 256, 0), (7, 384, 8), (8, 256, 16), (8, 1024, 0), (7, 128, 8),
                                             (17, 256, 0), (17, 384, 8), (17, 100, 0), (16, 256, 8), (16, 192, 0), (16, 100, 0)])
def test_gemm_split_transposed_wide_tile4(dev, tile, tokens, pad):
    """q | k | v projection with V written transposed.  tokens % 128 == 0 takes the whole-line kernel's STAGED epilogues (q | k tiles as 16-byte row
    chunks, V tiles transposed in LDS and written as 16-byte token runs); other token counts the direct one.  With a padded ld_out2 the row pad of
    V^T is part of the output: zero, although the buffer came from a NaN-poisoned torch.empty (conftest)."""
    from adaface_dev_amd import ops
    B, C, K = 2, (640 if tile == 16 else 320), 320          # tile 16's width is 128: C = 640 (q | k | v = 15 tiles)
    a, w = rnd((B * tokens, K), 1), rnd((3 * C, K), 2, K ** -0.5)
    b = torch.randn(3 * C, generator=torch.Generator().manual_seed(3)) * 0.3
    ld2 = ops.round_up(tokens, 8) + pad
    out, out2 = ops.gemm(a.to(dev), ops.pack_matrix(w, b, dev), rows_per_batch=tokens, split_col=2 * C, tile=tile, ld_out2=ld2)
    ref = a.float() @ w.float().t() + b
    assert rel_l2(out.float().cpu().numpy(), ref[:, :2 * C].numpy()) < TOL
    vt = ref[:, 2 * C:].reshape(B, tokens, C).permute(0, 2, 1)
    assert out2.shape == (B, C, ld2) and torch.isfinite(out2).all()
    assert rel_l2(out2[:, :, :tokens].float().cpu().numpy(), vt.numpy()) < TOL
    assert (out2[:, :, tokens:] == 0).all()


@pytest.mark.parametrize("tile,C", [(1, 64), (2, 64), (4, 64), (7, 64), (8, 64), (10, 64), (9, 320), (10, 320), (8, 320)])
def test_gemm_geglu(dev, tile, C):
    from adaface_dev_amd import ops
    M = 520
    a, w = rnd((M, C), 1), rnd((8 * C, C), 2, C ** -0.5)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    wi, bi = ops.interleave_geglu(w.float(), b)
    out = ops.gemm(a.to(dev), ops.pack_matrix(wi, bi, dev), act=ops.AF_ACT_GEGLU, tile=tile)
    h = a.float() @ w.float().t() + b
    x, g = h.chunk(2, dim=-1)
    ref = x * F.gelu(g)
    assert out.shape == (M, 4 * C)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("tokens,tile", [(77, 2), (64, 1), (256, 0), (77, 3), (256, 3), (77, 8), (256, 8)])
def test_gemm_split_transposed(dev, tokens, tile):
    from adaface_dev_amd import ops
    B, C, K = 3, 64, 96
    a, w = rnd((B * tokens, K), 1), rnd((3 * C, K), 2, K ** -0.5)
    out, out2 = ops.gemm(a.to(dev), ops.pack_matrix(w, None, dev), rows_per_batch=tokens, split_col=2 * C, tile=tile)
    ref = a.float() @ w.float().t()
    assert rel_l2(out.float().cpu().numpy(), ref[:, :2 * C].numpy()) < TOL
    vt = ref[:, 2 * C:].reshape(B, tokens, C).permute(0, 2, 1)
    assert out2.shape[:2] == (B, C) and out2.shape[2] % 8 == 0
    assert rel_l2(out2[:, :, :tokens].float().cpu().numpy(), vt.numpy()) < TOL


# ------------------------------------------------------------------------------- conv 3x3
@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride,ups,tile", [
    (2, 16, 16, 64, 0, 64, 1, False, 2), (2, 16, 16, 64, 0, 128, 1, False, 1), (1, 64, 64, 8, 0, 320, 1, False, 0),
    (2, 16, 16, 64, 0, 64, 2, False, 2), (2, 15, 17, 32, 0, 64, 2, False, 2), (2, 8, 8, 64, 0, 64, 1, True, 2),
    (2, 8, 8, 128, 64, 64, 1, False, 2), (1, 32, 32, 320, 0, 4, 1, False, 0), (2, 8, 8, 96, 32, 128, 1, False, 1),
    (2, 8, 8, 64, 0, 64, 1, True, 8), (2, 9, 7, 320, 0, 320, 1, True, 7), (1, 16, 16, 640, 0, 640, 1, True, 7), (3, 5, 6, 128, 0, 192, 1, True, 8),
    (2, 9, 7, 320, 0, 320, 1, True, 11), (1, 16, 16, 640, 0, 640, 1, True, 11), (3, 5, 6, 128, 0, 192, 1, True, 12), (2, 9, 7, 320, 0, 320, 1, True, 13),
])
def test_conv3x3(dev, B, H, W, c1, c2, cout, stride, ups, tile):
    from adaface_dev_amd import ops
    cin = c1 + c2
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(4))
    rowb = rnd((B, cout), 5)
    pw = ops.pack_conv3x3(w, bias, dev)
    xin = x1 if x2 is None else torch.cat([x1, x2], -1)
    xin = xin.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1) + rowb.float()[:, :, None, None]
    res = rnd(tuple(ref.permute(0, 2, 3, 1).shape), 6)
    ref = ref + res.float().permute(0, 3, 1, 2)
    out = ops.conv3x3(x1.to(dev), pw, x2=None if x2 is None else x2.to(dev), stride=stride, upsample=ups,
                      rowbias=rowb.to(dev), residual=res.to(dev), tile=tile)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


def test_conv3x3_padded_input_channels(dev):
    """first U-Net conv: 4 latent channels zero-padded to 8 (weights padded to match)."""
    from adaface_dev_amd import ops
    x = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    w = rnd((64, 4, 3, 3), 2, 1 / 6.0)
    bias = torch.randn(64, generator=torch.Generator().manual_seed(3))
    xh = ops.nchw_f32_to_nhwc_f16(x.to(dev), 8)
    assert xh.shape == (2, 16, 16, 8) and float(xh[..., 4:].abs().max()) == 0.0
    out = ops.conv3x3(xh, ops.pack_conv3x3(w, bias, dev, cin_pad=8))
    ref = F.conv2d(x.half().float(), w.float(), bias, padding=1)
    assert rel_l2(ops.nhwc_f16_to_nchw_f32(out).cpu().numpy(), ref.numpy()) < TOL


# ------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,HW,c1,c2,eps,silu", [
    (2, 64, 320, 0, 1e-5, True), (2, 256, 1280, 640, 1e-5, True), (1, 64, 2560, 0, 1e-5, True), (3, 100, 32, 0, 1e-6, False),
    (2, 4096, 320, 0, 1e-6, False), (2, 64, 640, 320, 1e-5, True), (2, 49, 64, 64, 1e-5, True),
    (8, 256, 1280, 0, 1e-5, True), (3, 256, 1280, 1280, 1e-5, True), (2, 64, 1280, 1280, 1e-6, False), (2, 144, 1280, 0, 1e-5, False),   # single-launch path
    (8, 1024, 640, 0, 1e-5, True), (3, 256, 1280, 640, 1e-5, True), (9, 256, 640, 0, 1e-6, False), (2, 1024, 320, 320, 1e-5, True),   # pair-granularity path
    (8, 4096, 320, 0, 1e-5, True), (8, 4096, 640, 0, 1e-5, True), (8, 4096, 640, 320, 1e-5, True), (8, 1024, 640, 320, 1e-5, True),     # the denoise step's
    (8, 1024, 1280, 0, 1e-5, False), (4, 4096, 640, 320, 1e-5, True), (5, 4000, 320, 0, 1e-6, True), (1, 4096, 128, 0, 1e-6, True),    # two-launch shapes ...
    (2, 16384, 256, 0, 1e-6, True), (1, 65536, 128, 0, 1e-6, True),                                                                   # ... and the VAE decoder's
    (1, 4096, 640, 320, 1e-5, True), (1, 4096, 320, 320, 1e-5, False), (2, 1024, 1280, 1280, 1e-5, True), (1, 4000, 320, 0, 1e-6, True),  # few batch items: the re-reading one-launch form
])
def test_groupnorm(dev, B, HW, c1, c2, eps, silu):
    from adaface_dev_amd import ops
    C = c1 + c2
    x1 = (rnd((B, HW, c1), 1, 2.0).float() + 0.7).half()
    x2 = rnd((B, HW, c2), 2, 0.5) if c2 else None
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.2 + 1
    b = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2
    y = ops.groupnorm(x1.to(dev), g.to(dev), b.to(dev), eps, silu, x2=None if x2 is None else x2.to(dev))
    xc = x1 if x2 is None else torch.cat([x1, x2], -1)
    ref = F.group_norm(xc.float().permute(0, 2, 1), 32, g, b, eps)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1)
    assert y.shape == (B, HW, C)
    assert rel_l2(y.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("ratio", [10.0, 100.0, 1000.0])
@pytest.mark.parametrize("B,HW,C,path", [
    (2, 64, 1280, "small"), (2, 256, 2560, "small"),          # one workgroup per (batch item, group), 16-byte chunks in registers
    (2, 1024, 640, "pair"), (3, 256, 960, "pair"),            # ... 4-byte pairs in registers
    (2, 4096, 320, "two-launch"), (1, 16384, 256, "two-launch"), (2, 4000, 640, "two-launch"),   # partial blocks + normalise
    (2, 4096, 320, "producer"), (1, 1024, 640, "producer"), (2, 4096, 320, "producer14"),         # partials from the producing GEMM's epilogue
    (1, 16384, 256, "producer14"), (2, 4096, 512, "producer14"),                                   # ... of the halo-resident kernel's 256 x 128 forms (patches / whole rows)
])
def test_groupnorm_with_group_mean_far_above_its_spread(dev, B, HW, C, path, ratio):
    """GroupNorm statistics when |mean| / sigma of a group is 10, 100, 1000 (outlier channels of real SD-1.5 checkpoints): a one-pass
    E[x^2] - mean^2 in fp32 loses log2(ratio^2) bits of the variance -- 10 % at 100, everything at 1000 -- while the reference, torch's fp32
    group_norm (util.py:195-212), does not cancel.  Every statistics path (in-register one-launch forms, partial blocks, partials left by the
    producing GEMM) is held to the ordinary tolerance against F.group_norm on the same fp16 inputs.  Values are exactly representable in fp16:
    per-group means of ratio * sigma with sigma = 1 / 4 ... 1 and a spread drawn on the fp16 grid at that magnitude."""
    from adaface_dev_amd import ops
    g = torch.Generator().manual_seed(11)
    cpg = C // 32
    sigma = 0.25 if ratio >= 1000 else 1.0
    mean = (ratio * sigma * (1 + 0.1 * torch.arange(32).float() / 32)).repeat_interleave(cpg)          # per group, |mean| / sigma ~ ratio
    sign = torch.where(torch.arange(C) // cpg % 2 == 0, 1.0, -1.0)
    x = (torch.randn((B, HW, C), generator=g) * sigma + mean * sign).half()                             # rounds to the fp16 grid at |mean|
    gam = torch.randn(C, generator=g) * 0.2 + 1
    bet = torch.randn(C, generator=g) * 0.2
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gam, bet, 1e-5).permute(0, 2, 1)
    if path.startswith("producer"):
        # an identity 1x1 GEMM / a centre-tap-identity 3x3 convolution on a statistics tile reproduces x bit for bit and leaves its partials
        H = W = int(HW ** 0.5)
        if path == "producer14":
            w = torch.zeros(C, C, 3, 3)
            w[torch.arange(C), torch.arange(C), 1, 1] = 1.0
            y = ops.conv3x3(x.reshape(B, H, W, C).to(dev), ops.pack_conv3x3(w, None, dev), tile=14, gn_cpg=cpg)
        else:
            y = ops.gemm(x.reshape(B * HW, C).to(dev), ops.pack_matrix(torch.eye(C), None, dev), rows_per_batch=HW, tile=7 if C % 320 == 0 and C < 640 else 11, gn_cpg=cpg)
            y4 = y.reshape(B, HW, C)
            y4._gn_partials = y._gn_partials
            y = y4
        assert getattr(y, "_gn_partials", None) is not None, "the launch was expected to leave its statistics"
        assert torch.equal(y.reshape(B, HW, C).cpu(), x), "identity weights must reproduce the input"
        out = ops.groupnorm(y, gam.to(dev), bet.to(dev), 1e-5, False).reshape(B, HW, C)
    else:
        out = ops.groupnorm(x.to(dev), gam.to(dev), bet.to(dev), 1e-5, False)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,HW,C,path", [
    (2, 64, 1280, "small"), (2, 1024, 640, "pair"), (2, 4096, 320, "two-launch"), (2, 4096, 320, "producer"), (1, 1024, 640, "producer"),
    (2, 4096, 320, "producer14"), (1, 16384, 128, "producer14"),
])
def test_groupnorm_with_opposite_signed_outliers_near_the_fp16_maximum(dev, B, HW, C, path):
    """Groups that hold +6e4 and -6e4 (the SD fp16 VAE / decoder trunk carries activations in the 1e4 range): |x - pivot| reaches 1.2e5, past
    the fp16 maximum, so a packed-fp16 `x - pivot` is inf, M2 = inf - inf = NaN and the whole group comes out NaN (round-5 advisor finding).
    The statistics take HALF the difference, 0.5 x - 0.5 p, which cannot overflow (af_common.h gn_half_diff).  Every statistics path, against
    torch's fp32 group_norm; the pivot (the group's first element) is one of the outliers in the odd groups."""
    from adaface_dev_amd import ops
    g = torch.Generator().manual_seed(13)
    cpg = C // 32
    x = torch.randn((B, HW, C), generator=g)
    grp = torch.arange(C) // cpg
    for b in range(B):
        for gi in range(0, 32, 3):                                 # every third group: a handful of +-6e4 values
            ch0 = gi * cpg
            x[b, 0, ch0] = 6.0e4 if gi % 2 else -6.0e4                # the pivot itself
            x[b, HW // 2, ch0 + 1] = -6.0e4 if gi % 2 else 6.0e4      # the opposite sign, same channel pair
            x[b, HW - 1, ch0 + cpg - 1] = 6.0e4
    x = x.half()
    gam = torch.randn(C, generator=g) * 0.2 + 1
    bet = torch.randn(C, generator=g) * 0.2
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gam, bet, 1e-5).permute(0, 2, 1)
    if path.startswith("producer"):
        H = W = int(HW ** 0.5)
        if path == "producer14":
            w = torch.zeros(C, C, 3, 3)
            w[torch.arange(C), torch.arange(C), 1, 1] = 1.0
            y = ops.conv3x3(x.reshape(B, H, W, C).to(dev), ops.pack_conv3x3(w, None, dev), tile=14, gn_cpg=cpg)
        else:
            y = ops.gemm(x.reshape(B * HW, C).to(dev), ops.pack_matrix(torch.eye(C), None, dev), rows_per_batch=HW, tile=7 if C % 320 == 0 and C < 640 else 11, gn_cpg=cpg)
            y4 = y.reshape(B, HW, C)
            y4._gn_partials = y._gn_partials
            y = y4
        assert getattr(y, "_gn_partials", None) is not None, "the launch was expected to leave its statistics"
        assert torch.equal(y.reshape(B, HW, C).cpu(), x), "identity weights must reproduce the input"
        out = ops.groupnorm(y, gam.to(dev), bet.to(dev), 1e-5, False).reshape(B, HW, C)
    else:
        out = ops.groupnorm(x.to(dev), gam.to(dev), bet.to(dev), 1e-5, False)
    out = out.float().cpu()
    assert torch.isfinite(out).all(), "an overflowed statistic poisons its whole group"
    assert rel_l2(out.numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,H,cin,cin2,cout,tile,splits,form", [
    (2, 16, 1280, 0, 1280, 7, 4, "small"), (8, 16, 640, 0, 1280, 14, 4, "small"), (3, 8, 1280, 1280, 1280, 1, 12, "small"),
    (2, 32, 640, 0, 640, 14, 2, "pair"), (8, 32, 320, 0, 640, 7, 2, "pair"), (2, 32, 640, 320, 640, 11, 3, "pair"), (1, 16, 320, 0, 320, 2, 5, "pair"),
    (1, 8, 128, 0, 64, 2, 3, "pair"),
])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_absorbs_the_split_k_reduce_bit_identically(dev, B, H, cin, cin2, cout, tile, splits, form, silu, monkeypatch):
    """Round 6: a split-K convolution whose next reader is a GroupNorm leaves its reduce pass to it (af_gemm_desc.defer_reduce ->
    af_groupnorm_splitk: slabs -> bias / time-embedding row bias / residual -> the convolution's fp16 output stored AND normalised, one launch).
    Against the two-launch form (af_splitk_reduce_kernel, then gn_small / gn_pair) on the same launches: the stored convolution output, the
    normalised tensor and the (mean, rstd) statistics of the training form must be BIT-IDENTICAL (same summation order, same statistics
    arithmetic); and against fp32 torch within the ordinary tolerance.  Both one-launch forms, every epilogue input, one / two sources, the
    K-concatenated shortcut, register-staged and LDS-DMA tiles."""
    from adaface_dev_amd import ops, rng
    W = H
    tag = f"dg{B}.{H}.{cin}.{cin2}.{cout}"
    x = rng.synth_input(tag + ".x", (B, H, W, cin), seed=31).half().to(dev)
    x2 = rng.synth_input(tag + ".x2", (B, H, W, cin2), seed=31).half().to(dev) if cin2 else None
    w = rng.synth_input(tag + ".w", (cout, cin + cin2, 3, 3), seed=31) * ((cin + cin2) * 9) ** -0.5
    bias = rng.synth_input(tag + ".b", (cout,), seed=31)
    pw = ops.pack_conv3x3(w, bias, dev)
    rowb = rng.synth_input(tag + ".rb", (B, cout + 8), seed=31).half().to(dev)[:, :cout]          # a strided view, as the batched emb_layers output
    res = rng.synth_input(tag + ".res", (B, H, W, cout), seed=31).half().to(dev)
    gam = (rng.synth_input(tag + ".g", (cout,), seed=31) * 0.2 + 1).to(dev)
    bet = (rng.synth_input(tag + ".bt", (cout,), seed=31) * 0.2).to(dev)
    cpg = cout // 32

    def run(defer, train):
        monkeypatch.setattr(ops, "DEFER_GN_REDUCE", defer)
        h = ops.conv3x3(x, pw, x2=x2, rowbias=rowb, residual=res, tile=tile, splits=splits, gn_cpg=cpg, defer_gn=True)
        if train:
            y, st = ops.groupnorm_train(h, gam, bet, 1e-5, silu)
        else:
            y, st = ops.groupnorm(h, gam, bet, 1e-5, silu), None
        return h, y, st

    for train in (False, True):
        before = dict(ops.pending_stats)
        h0, y0, st0 = run(False, train)
        assert ops.pending_stats == before, "the switch is off: nothing may be deferred"
        h1, y1, st1 = run(True, train)
        assert ops.pending_stats["deferred"] == before["deferred"] + 1 and ops.pending_stats["fused"] == before["fused"] + 1, \
            f"the launch was expected to defer its reduce pass and the GroupNorm to absorb it ({ops.pending_stats} vs {before})"
        assert not ops._pending
        assert torch.equal(h0, h1), "convolution output"
        assert torch.equal(y0, y1), "normalised output"
        if train:
            assert torch.equal(st0, st1), "statistics"
    xin = x if x2 is None else torch.cat([x, x2], -1)
    ref_h = F.conv2d(xin.float().cpu().permute(0, 3, 1, 2), w.half().float(), bias, padding=1) + rowb.float().cpu()[:, :, None, None] \
        + res.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(h1.float().cpu().permute(0, 3, 1, 2).numpy(), ref_h.numpy()) < TOL
    ref_y = F.group_norm(h1.float().cpu().permute(0, 3, 1, 2), 32, gam.cpu(), bet.cpu(), 1e-5)
    ref_y = F.silu(ref_y) if silu else ref_y
    assert rel_l2(y1.float().cpu().permute(0, 3, 1, 2).numpy(), ref_y.numpy()) < TOL


def test_deferred_split_k_reduce_is_materialised_for_any_other_reader(dev, monkeypatch):
    """The safety net of ops.PendingReduce: a tensor whose reduce pass was left to "the next GroupNorm" but is read by something else first -- another
    wrapper (through a VIEW), a further GEMM launch (which reuses the slab workspace), a GroupNorm outside the one-launch scope or over two sources --
    is finished by the plain reduce pass (af_splitk_reduce) before that reader runs.  Every path must give the bits of the never-deferred run."""
    from adaface_dev_amd import ops, rng
    B, H, C = 2, 16, 640
    x = rng.synth_input("dm.x", (B, H, H, C), seed=32).half().to(dev)
    pw = ops.pack_conv3x3(rng.synth_input("dm.w", (C, C, 3, 3), seed=32) * (C * 9) ** -0.5, rng.synth_input("dm.b", (C,), seed=32), dev)
    pw1 = ops.pack_matrix(rng.synth_input("dm.w1", (C, C), seed=32) * C ** -0.5, None, dev)
    gam, bet = torch.ones(2 * C, device=dev), torch.zeros(2 * C, device=dev)
    monkeypatch.setattr(ops, "DEFER_GN_REDUCE", False)
    want = ops.conv3x3(x, pw, tile=7, splits=3, gn_cpg=C // 32, defer_gn=True)
    want_g = ops.gemm(want.reshape(-1, C), pw1, tile=2, splits=2)
    want_gn2 = ops.groupnorm(want, gam, bet, 1e-5, True, x2=x)
    monkeypatch.setattr(ops, "DEFER_GN_REDUCE", True)

    def owed():
        h = ops.conv3x3(x, pw, tile=7, splits=3, gn_cpg=C // 32, defer_gn=True)
        assert ops._pending, "expected the reduce pass to be owed"
        return h
    h = owed()
    assert torch.equal(ops.add(h.reshape(B * H, H, C), h.reshape(B * H, H, C)), ops.add(want, want).reshape(B * H, H, C)) and not ops._pending     # a wrapper, through a view
    assert torch.equal(h, want)
    h = owed()
    assert torch.equal(ops.gemm(h.reshape(-1, C), pw1, tile=2, splits=2), want_g) and not ops._pending and torch.equal(h, want)      # operand of a split GEMM
    h = owed()
    assert torch.equal(ops.groupnorm(h, gam, bet, 1e-5, True, x2=x), want_gn2) and not ops._pending and torch.equal(h, want)            # two-source GroupNorm
    h = owed()
    other = ops.conv3x3(x, pw, tile=7, splits=2)                                                                                       # an unrelated launch
    assert not ops._pending and torch.equal(h, want)
    h = owed()
    ops.flush_pending()
    assert torch.equal(h, want)
    # a GroupNorm outside the one-launch forms (64 x 64 x 320: the two-launch statistics path) never defers
    xb = rng.synth_input("dm.xb", (2, 64, 64, 320), seed=32).half().to(dev)
    pwb = ops.pack_conv3x3(rng.synth_input("dm.wb", (320, 320, 3, 3), seed=32) * (320 * 9) ** -0.5, None, dev)
    hb = ops.conv3x3(xb, pwb, tile=7, splits=2, gn_cpg=10, defer_gn=True)
    assert not ops._pending
    monkeypatch.setattr(ops, "DEFER_GN_REDUCE", False)
    assert torch.equal(hb, ops.conv3x3(xb, pwb, tile=7, splits=2, gn_cpg=10, defer_gn=True))


@pytest.mark.parametrize("rows,C", [(77, 320), (1000, 640), (513, 1280), (64, 32), (10, 2048)])
def test_layernorm(dev, rows, C):
    from adaface_dev_amd import ops
    x = (rnd((rows, C), 1, 3.0).float() - 1.0).half()
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.2 + 1
    b = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2
    y = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-5)
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5)
    assert rel_l2(y.float().cpu().numpy(), ref.numpy()) < TOL


def _ln_inputs(rows, C, seed=1):
    # rows with a mean far from zero (3 sigma) and an outlier channel, as the residual stream of a transformer block has
    x = (rnd((rows, C), seed, 2.0).float() + 3.0)
    x[:, 5] *= 8.0
    x = x.half()
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.2 + 1
    b = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2
    return x, g, b


@pytest.mark.parametrize("rows,C,N,tile", [(300, 320, 320, 7), (300, 320, 320, 8), (1000, 640, 640, 8), (513, 1280, 1280, 8), (520, 320, 640, 11),
                                           (260, 320, 320, 12), (260, 320, 640, 13), (300, 320, 320, 0), (64, 1280, 1280, 0), (130, 64, 64, 0)])
def test_gemm_folded_layernorm(dev, rows, C, N, tile):
    """LayerNorm folded into the consuming GEMM (af_gemm_desc.ln_colsum, ops.pack_matrix_ln) == layer_norm then linear in fp32.
    The un-fused pair rounds LN(x) to fp16 before the GEMM; the folded form does not, so the same one-rounding tolerance holds."""
    from adaface_dev_amd import ops
    x, g, b = _ln_inputs(rows, C)
    w = rnd((N, C), 2, C ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(5)) * 0.3
    r = rnd((rows, N), 6)
    pw = ops.pack_matrix_ln(w, bias, g, b, 1e-5, dev)
    out = ops.gemm(x.to(dev), pw, residual=r.to(dev), tile=tile)
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5) @ w.float().t() + bias + r.float()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("ratio", [10.0, 30.0])
@pytest.mark.parametrize("C,tile", [(320, 7), (640, 8), (1280, 16)])
def test_gemm_folded_layernorm_rows_with_mean_above_their_spread(dev, C, tile, ratio):
    """The folded LayerNorm takes a row's variance as q / K - mean^2 from fp32 sums of the A fragments (one pass, inside the GEMM's main loop).  Unlike a
    GroupNorm group, a token's channel vector has |mean| / sigma below ~1 in this network (an outlier CHANNEL widens the row's spread, it does not move
    its mean), but the form's margin is stated and held here: rows with |mean| / sigma = 10 and 30 stay inside the ordinary tolerance (the fp32 sums of
    320 ... 1280 squares lose ~6e-8 K^0.5 (mean / sigma)^2 of the variance: 1e-4 at 10, 1e-3 at 30; 100 would be past it -- af_norm.hip's shifted
    partials are what a group statistic needs, §4.12)."""
    from adaface_dev_amd import ops
    g0 = torch.Generator().manual_seed(21)
    rows = 260
    x = (torch.randn((rows, C), generator=g0) + ratio * torch.where(torch.arange(rows) % 2 == 0, 1.0, -1.0)[:, None]).half()
    g = torch.randn(C, generator=g0) * 0.2 + 1
    b = torch.randn(C, generator=g0) * 0.2
    w = rnd((C, C), 2, C ** -0.5)
    out = ops.gemm(x.to(dev), ops.pack_matrix_ln(w, None, g, b, 1e-5, dev), tile=tile)
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5) @ w.float().t()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("tile,C,M", [(7, 64, 520), (8, 64, 520), (9, 320, 520), (10, 320, 520), (8, 320, 520), (0, 320, 520),
                                      (9, 320, 128), (9, 1280, 128), (9, 1280, 520), (10, 1280, 128)])
def test_gemm_folded_layernorm_geglu(dev, tile, C, M):
    """(The M = 128 / C = 1280 cases of tile 9 -- the 256 x 320 tile at its register limit -- faulted while the weight prefetch parked its
    loads in VGPRs the compiler could spill and re-use before they landed; tools/probes/r03ae_ln_geglu_fault.py.)"""
    from adaface_dev_amd import ops
    from adaface_dev_amd.ldm.modules.attention import GEGLU
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm
    x, g, b = _ln_inputs(M, C)
    m, ln = GEGLU(C, 4 * C).to(dev), LayerNorm(C).to(dev)
    with torch.no_grad():
        m.proj.weight.copy_(rnd((8 * C, C), 2, C ** -0.5).float())
        m.proj.bias.copy_(torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1)
        ln.weight.copy_(g)
        ln.bias.copy_(b)
    out = ops.gemm(x.to(dev), m.packed_ln(ln), act=ops.AF_ACT_GEGLU, tile=tile)
    h = F.layer_norm(x.float(), (C,), g, b, 1e-5) @ m.proj.weight.detach().float().cpu().t() + m.proj.bias.detach().float().cpu()
    xv, gv = h.chunk(2, dim=-1)
    ref = xv * F.gelu(gv)
    assert out.shape == (M, 4 * C)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("tokens,tile,C", [(100, 7, 320), (77, 8, 64), (256, 8, 640), (100, 0, 320)])
def test_gemm_folded_layernorm_split_transposed(dev, tokens, tile, C):
    """q | k | v of a self-attention layer from the un-normalised rows: q | k row-major, v transposed."""
    from adaface_dev_amd import ops
    B = 2
    x, g, b = _ln_inputs(B * tokens, C)
    w = rnd((3 * C, C), 2, C ** -0.5)
    out, out2 = ops.gemm(x.to(dev), ops.pack_matrix_ln(w, None, g, b, 1e-5, dev), rows_per_batch=tokens, split_col=2 * C, tile=tile)
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5) @ w.float().t()
    assert rel_l2(out.float().cpu().numpy(), ref[:, :2 * C].numpy()) < TOL
    vt = ref[:, 2 * C:].reshape(B, tokens, C).permute(0, 2, 1)
    assert rel_l2(out2[:, :, :tokens].float().cpu().numpy(), vt.numpy()) < TOL


@pytest.mark.parametrize("M,with_ln,with_res", [(256, True, True), (1000, True, True), (4096, True, False), (130, False, True), (24576, True, True)])
def test_ff_fused_c320(dev, M, with_ln, with_res):
    """af_ff_fused: LayerNorm -> GEGLU projection -> output projection (+ bias, + residual) of a C = 320 transformer block in one
    launch == the same chain in fp32 (attention.py:31-58, 242-252).  Ragged token counts, with / without the LayerNorm and residual."""
    from adaface_dev_amd import ops
    from adaface_dev_amd.ldm.modules.attention import FeedForward
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm
    C = 320
    x, g, b = _ln_inputs(M, C)
    ff, ln = FeedForward(C, glu=True).to(dev), LayerNorm(C).to(dev)
    with torch.no_grad():
        ff.net[0].proj.weight.copy_(rnd((8 * C, C), 2, C ** -0.5).float())
        ff.net[0].proj.bias.copy_(torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1)
        ff.net[2].weight.copy_(rnd((C, 4 * C), 4, (4 * C) ** -0.5).float())
        ff.net[2].bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(5)) * 0.1)
        ln.weight.copy_(g)
        ln.bias.copy_(b)
    res = x.to(dev) if with_res else None
    if with_ln:
        pw1 = ff.net[0].packed_ln(ln)
    else:
        pw1 = ff.net[0].packed()
    out = ops.ff_fused(x.to(dev), pw1, ff.net[2].packed(), residual=res)
    xin = F.layer_norm(x.float(), (C,), g, b, 1e-5) if with_ln else x.float()
    h = xin @ ff.net[0].proj.weight.detach().float().cpu().t() + ff.net[0].proj.bias.detach().float().cpu()
    xv, gv = h.chunk(2, dim=-1)
    ref = (xv * F.gelu(gv)) @ ff.net[2].weight.detach().float().cpu().t() + ff.net[2].bias.detach().float().cpu()
    if with_res:
        ref = ref + x.float()
    assert out.shape == (M, C)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    # and it is what FeedForward.hip launches for a 64 x 64-level batch
    if with_ln and with_res and M >= 24576:
        y = ff.hip(x.to(dev), residual=res, ln=ln)
        assert torch.equal(y, out)


@pytest.mark.parametrize("B,N,with_gn", [(2, 256, True), (1, 384, False), (3, 128, True), (8, 4096, True)])
def test_ff_chain_c320_proj_out_tail(dev, B, N, with_gn):
    """af_ff_chain: the one-launch C = 320 feed-forward with the SpatialTransformer's proj_out + residual as its tail (attention.py:287-304) against the same chain
    in fp32, and BIT FOR BIT against the two launches it replaces (af_ff_fused, then the proj_out GEMM with x_in as residual: the intermediate takes the same fp16
    rounding in LDS as in memory); the GroupNorm partials it leaves give the same GroupNorm as the tensor's own statistics.  The 8 x 4096 case is what the
    SpatialTransformer of the 64 x 64 level launches."""
    from adaface_dev_amd import ops
    from adaface_dev_amd.ldm.modules import attention as A
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm
    C, M = 320, B * N
    x, g, b = _ln_inputs(M, C)
    x_in = rnd((M, C), 21)
    ff, ln = A.FeedForward(C, glu=True).to(dev), LayerNorm(C).to(dev)
    wp, bp = rnd((C, C), 22, C ** -0.5), torch.randn(C, generator=torch.Generator().manual_seed(23)) * 0.1
    with torch.no_grad():
        ff.net[0].proj.weight.copy_(rnd((8 * C, C), 2, C ** -0.5).float())
        ff.net[0].proj.bias.copy_(torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1)
        ff.net[2].weight.copy_(rnd((C, 4 * C), 4, (4 * C) ** -0.5).float())
        ff.net[2].bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(5)) * 0.1)
        ln.weight.copy_(g)
        ln.bias.copy_(b)
    pw1, pw2, pwp = ff.net[0].packed_ln(ln), ff.net[2].packed(), ops.pack_matrix(wp, bp, dev)
    out = ops.ff_chain(x.to(dev), pw1, pw2, x.to(dev), pwp, x_in.to(dev), rows_per_batch=N, gn_cpg=10 if with_gn else 0)
    # two launches
    x3 = ops.ff_fused(x.to(dev), pw1, pw2, residual=x.to(dev))
    out2 = ops.gemm(x3, pwp, residual=x_in.to(dev), tile=7)
    assert torch.equal(out, out2) or rel_l2(out.float().cpu().numpy(), out2.float().cpu().numpy()) < 2e-4      # (same products; the GEMM's K order may differ by tile)
    # fp32
    if M <= 4096:
        h = F.layer_norm(x.float(), (C,), g, b, 1e-5) @ ff.net[0].proj.weight.detach().float().cpu().t() + ff.net[0].proj.bias.detach().float().cpu()
        xv, gv = h.chunk(2, dim=-1)
        r3 = (xv * F.gelu(gv)) @ ff.net[2].weight.detach().float().cpu().t() + ff.net[2].bias.detach().float().cpu() + x.float()
        ref = r3 @ wp.float().t() + bp + x_in.float()
        assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    if with_gn:
        gn = ops.partials_of(out)
        assert gn is not None and gn.cpg == 10
        gam, bet = torch.randn(C, generator=torch.Generator().manual_seed(7)) * 0.2 + 1, torch.randn(C, generator=torch.Generator().manual_seed(8)) * 0.2
        y_p = ops.groupnorm(out.reshape(B, N, C), gam.to(dev), bet.to(dev), 1e-5, True)            # from the partials
        y_s = ops.groupnorm(out.clone().reshape(B, N, C), gam.to(dev), bet.to(dev), 1e-5, True)    # its own statistics pass
        assert rel_l2(y_p.float().cpu().numpy(), y_s.float().cpu().numpy()) < 1e-3
    else:
        assert ops.partials_of(out) is None
    for _ in range(3):
        assert torch.equal(ops.ff_chain(x.to(dev), pw1, pw2, x.to(dev), pwp, x_in.to(dev), rows_per_batch=N, gn_cpg=0), out)


@pytest.mark.parametrize("C", [320, 640])
@pytest.mark.parametrize("B,N,L,with_ln,with_res,ld_extra", [(2, 256, 77, True, True, 0), (1, 128, 77, False, False, 0), (3, 384, 80, True, False, 0),
                                                             (2, 128, 40, True, True, 0), (2, 128, 37, True, True, 0), (2, 256, 77, True, True, 16),
                                                             (1, 128, 5, False, True, 8), (2, 128, 64, True, False, 8), (3, 64, 77, True, True, 8)])
def test_xattn_fused_c320(dev, B, N, L, with_ln, with_res, ld_extra, C):
    """af_xattn_fused (the whole C = 320 / C = 640 cross-attention block in one launch: LayerNorm-folded q projection, 77-key softmax attention on
    the context projection's K / V^T slices, to_out, bias, residual) against fp32 torch AND against the three-launch path of
    CrossAttention.hip on the same weights; K / V^T handed over exactly as the U-Net does (column / row slices of a wider batched
    projection, so the strides are not the layer's own)."""
    from adaface_dev_amd import ops
    from adaface_dev_amd.ldm.modules import attention as A
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm
    if C == 320 and N % 128:
        pytest.skip("C = 320: 128-token workgroups")
    Cc, heads, dh = 768, 8, C // 8
    m = A.CrossAttention(C, Cc, heads=heads, dim_head=dh).to(dev)
    ln = LayerNorm(C).to(dev)
    with torch.no_grad():
        for i, w in enumerate((m.to_q.weight, m.to_k.weight, m.to_v.weight, m.to_out[0].weight)):
            w.copy_(rnd(tuple(w.shape), 30 + i, w.shape[1] ** -0.5).float())
        m.to_out[0].bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.1)
        ln.weight.copy_(torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2 + 1)
        ln.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(5)) * 0.2)
    x, _, _ = _ln_inputs(B * N, C, seed=7)
    ctx = rnd((B, L, Cc), 8)
    res = rnd((B * N, C), 9) if with_res else None
    # reference in fp32
    xf = x.float()
    xn = F.layer_norm(xf, (C,), ln.weight.detach().float().cpu(), ln.bias.detach().float().cpu(), 1e-5) if with_ln else xf
    W = [w.detach().float().cpu() for w in (m.to_q.weight, m.to_k.weight, m.to_v.weight, m.to_out[0].weight)]
    q = (xn @ W[0].t()).reshape(B, N, heads, dh).permute(0, 2, 1, 3)
    kk = (ctx.float() @ W[1].t()).reshape(B, L, heads, dh).permute(0, 2, 1, 3)
    vv = (ctx.float() @ W[2].t()).reshape(B, L, heads, dh).permute(0, 2, 1, 3)
    o = torch.softmax(q @ kk.transpose(-1, -2) * dh ** -0.5, dim=-1) @ vv
    ref = o.permute(0, 2, 1, 3).reshape(B * N, C) @ W[3].t() + m.to_out[0].bias.detach().float().cpu()
    if with_res:
        ref = ref + res.float()
    # K / V^T as slices of a wider batched projection (another layer's 640 columns in front, 320 behind)
    pad_w = [rnd((640, Cc), 40, Cc ** -0.5), rnd((320, Cc), 41, Cc ** -0.5)]
    wk = torch.cat([pad_w[0], m.to_k.weight.detach().cpu().half(), pad_w[1]], 0)
    wv = torch.cat([pad_w[0], m.to_v.weight.detach().cpu().half(), pad_w[1]], 0)
    pack = ops.pack_matrix(torch.cat([wk, wv], 0), None, dev)
    ldv = (L + 7) // 8 * 8 + ld_extra
    k_all, vt_all = ops.gemm(ctx.to(dev).reshape(B * L, Cc), pack, rows_per_batch=L, split_col=wk.shape[0], ld_out2=ldv)
    # the projection's transposed output owns its row pad: zero, although the buffer came from a (NaN-poisoned, see conftest) torch.empty
    assert vt_all.shape[-1] == ldv and torch.isfinite(vt_all).all() and (vt_all[..., L:] == 0).all()
    k, vt = k_all[:, 640:640 + C], vt_all[:, 640:640 + C, :]
    pq = m._packed_q_ln(ln) if with_ln else m.to_q.packed()
    out = ops.xattn_fused(x.to(dev), pq, k, vt, m.to_out[0].packed(), B=B, N=N, L=L, heads=heads, scale=dh ** -0.5, ldk=wk.shape[0],
                          residual=None if res is None else res.to(dev))
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < TOL
    # the three-launch path of the module on the same operands
    m._kv_pre = (k, vt, wk.shape[0])
    old = A.FUSE_XATTN
    if C == 640:                                        # and the module takes the one-launch form when its switch is on
        old640, A.FUSE_XATTN640, min640, A.XATTN640_FUSE_MIN_TOKENS = A.FUSE_XATTN640, True, A.XATTN640_FUSE_MIN_TOKENS, 64
        try:
            if with_ln and A.FUSE_XATTN:
                assert m.xattn_fusable(C, B, N, L, ln) and not m.xattn_fusable(C, B, N, L, ln, chain=True)
                out1 = m.hip(x.to(dev), B, N, context=ctx.to(dev), residual=None if res is None else res.to(dev), ln=ln)
                assert torch.equal(out1, out)
        finally:
            A.FUSE_XATTN640, A.XATTN640_FUSE_MIN_TOKENS = old640, min640
    A.FUSE_XATTN = False
    try:
        out3 = m.hip(x.to(dev), B, N, context=ctx.to(dev), residual=None if res is None else res.to(dev), ln=ln if with_ln else None)
    finally:
        A.FUSE_XATTN = old
        m._kv_pre = None
    assert rel_l2(out.float().cpu().numpy(), out3.float().cpu().numpy()) < TOL
    # masked keys contribute NOTHING whatever the pad holds (attention.py:196-202; 0 x NaN = NaN in an MFMA): NaN / Inf behind key L - 1 of every V^T
    # row, in a caller-owned buffer that no producer of ours has cleaned -- both the one-launch kernel and the three-launch core must not see it
    if ldv > L:
        for bad in (float("nan"), float("inf"), -65504.0):
            vt_h = vt_all.clone()
            vt_h[..., L:] = bad
            out_h = ops.xattn_fused(x.to(dev), pq, k, vt_h[:, 640:640 + C, :], m.to_out[0].packed(), B=B, N=N, L=L, heads=heads, scale=dh ** -0.5,
                                    ldk=wk.shape[0], residual=None if res is None else res.to(dev))
            assert torch.isfinite(out_h).all(), f"pad = {bad}"
            assert torch.equal(out_h, out), f"pad = {bad}"
            m._kv_pre = (k, vt_h[:, 640:640 + C, :], wk.shape[0])
            A.FUSE_XATTN = False
            try:
                out3_h = m.hip(x.to(dev), B, N, context=ctx.to(dev), residual=None if res is None else res.to(dev), ln=ln if with_ln else None)
            finally:
                A.FUSE_XATTN = old
                m._kv_pre = None
            assert torch.isfinite(out3_h).all() and torch.equal(out3_h, out3), f"pad = {bad} (three launches)"


def test_gemm_folded_layernorm_refuses_other_kernels(dev):
    from adaface_dev_amd import ops
    x, g, b = _ln_inputs(64, 64)
    pw = ops.pack_matrix_ln(rnd((64, 64), 2), None, g, b, 1e-5, dev)
    d_ok = ops.gemm(x.to(dev), pw, tile=8)
    assert torch.isfinite(d_ok.float()).all()
    import adaface_dev_amd.ops as O
    # an explicit register-staged tile cannot carry the fold: the wrapper moves it to a whole-line tile, the C ABI itself refuses
    from adaface_dev_amd import _lib
    import ctypes as C
    d = _lib.GemmDesc()
    out = torch.empty((64, 64), dtype=torch.float16, device=dev)
    xd = x.to(dev)
    d.a1, d.wt, d.out, d.ln_colsum, d.ln_eps = xd.data_ptr(), pw.wt.data_ptr(), out.data_ptr(), pw.ln_cs.data_ptr(), 1e-5
    d.M, d.N, d.K, d.kpad, d.taps, d.c1, d.lda1, d.tile = 64, 64, 64, pw.kpad, 1, 64, 64, 2
    d.zeros = O._zero_page(dev).data_ptr()
    rc = _lib.lib().af_gemm(C.byref(d), torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and "LayerNorm" in _lib.lib().af_last_error().decode()


# ------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, heads, mask=None):
    from oracle.unet_oracle import attention_core
    return attention_core(q.float(), k.float(), v.float(), heads, mask)


@pytest.mark.parametrize("B,N,L,heads,d,masked", [
    (2, 64, 64, 8, 8, False), (2, 256, 256, 8, 40, False), (1, 200, 77, 8, 40, False), (2, 48, 77, 2, 40, False),
    (1, 1024, 1024, 8, 80, False), (1, 256, 256, 8, 160, False), (2, 96, 97, 4, 64, False), (2, 64, 64, 8, 16, True),
    (1, 300, 300, 8, 40, True), (2, 128, 77, 8, 32, False), (1, 4096, 4096, 2, 40, False),
    (2, 640, 192, 3, 40, False), (1, 520, 128, 2, 24, False), (3, 1024, 1024, 1, 48, False),      # two-chain kernel incl. ragged query tails
])
def test_attention(dev, B, N, L, heads, d, masked):
    from adaface_dev_amd import ops
    C = heads * d
    q, k, v = rnd((B, N, C), 1), rnd((B, L, C), 2), rnd((B, L, C), 3)
    mask = None
    kb = None
    if masked:
        mask = torch.rand(B, L, generator=torch.Generator().manual_seed(9)) > 0.4
        mask[0, : L // 2] = False
        kb = ops.make_keybias(mask.to(dev), L)
    ldv = ops.round_up(L, 8)
    vt = torch.full((B, C, ldv), float("nan"), dtype=torch.float16)  # padding must never be read as data
    vt[:, :, :L] = v.permute(0, 2, 1)
    o = ops.attention(q.reshape(B * N, C).to(dev), k.reshape(B * L, C).to(dev), vt.to(dev), B=B, Nq=N, L=L, heads=heads,
                      d=d, ldq=C, ldk=C, keybias=kb)
    ref = _attn_ref(q, k, v, heads, mask)
    assert rel_l2(o.float().cpu().reshape(B, N, C).numpy(), ref.numpy()) < TOL


def test_attention_all_keys_masked_is_uniform(dev):
    """masked_fill(-finfo.max) on every key gives a uniform softmax, not NaN (attention.py:183-194)."""
    from adaface_dev_amd import ops
    B, N, heads, d = 1, 64, 8, 40
    C = heads * d
    q, k, v = rnd((B, N, C), 1), rnd((B, N, C), 2), rnd((B, N, C), 3)
    kb = ops.make_keybias(torch.zeros(B, N, dtype=torch.bool, device=dev), N)
    o = ops.attention(q.reshape(N, C).to(dev), k.reshape(N, C).to(dev), v.permute(0, 2, 1).contiguous().to(dev), B=B, Nq=N,
                      L=N, heads=heads, d=d, ldq=C, ldk=C, keybias=kb)
    ref = v.float().mean(dim=1, keepdim=True).expand(B, N, C)
    assert torch.isfinite(o).all()
    assert rel_l2(o.float().cpu().reshape(B, N, C).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("N", [128, 512])
def test_attention_online_softmax_rescale_branch(dev, N):
    """Force the running max to jump late (a spiked key in the last stage, far beyond the 2^8 lazy-reference margin) -- the
    rescale of the O accumulator must be exact (cdna guide rule 26: rare data-dependent branch needs its own test).
    N = 128: one-chain kernel; N = 512: two-chain kernel."""
    from adaface_dev_amd import ops
    B, heads, d = 1, 2, 40
    C = heads * d
    L = 320
    q, k, v = rnd((B, N, C), 1), rnd((B, L, C), 2, 0.3), rnd((B, L, C), 3)
    k[0, L - 3] = (q[0, 5] * 4).half()  # large positive score against query 5 (and others) in the final stage
    if N >= 512:
        # the pipelined two-chain kernel (d = 40) keeps two skewed chains per wave: also move the reference of a chain-B query (40) in a MIDDLE
        # stage, and start a query (70) from a strongly NEGATIVE first stage (its reference must start at that maximum, not at 0)
        k[0, 130] = (q[0, 40] * 4).half()
        k[0, :64] = -(q[0, 70] * 2).half()
    ldv = ops.round_up(L, 8)
    vt = torch.zeros((B, C, ldv), dtype=torch.float16)
    vt[:, :, :L] = v.permute(0, 2, 1)
    o = ops.attention(q.reshape(N, C).to(dev), k.reshape(L, C).to(dev), vt.to(dev), B=B, Nq=N, L=L, heads=heads, d=d, ldq=C,
                      ldk=C)
    ref = _attn_ref(q, k, v, heads)
    assert rel_l2(o.float().cpu().reshape(B, N, C).numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("B,N,L", [(2, 4096, 77), (8, 1024, 77), (1, 8192, 80), (3, 3072, 50)])
def test_xattn_chain_c320_transformer_block(dev, B, N, L, monkeypatch):
    """Round 6, af_xattn_chain: the self-attention's output projection + residual as phase 0 of the one-launch C = 320 cross-attention block.  A whole
    BasicTransformerBlock.hip (attention.py:242-252: LayerNorm-folded q | k | v, self-attention, to_out + x, cross-attention on 77 keys + x, GEGLU
    feed-forward + x) with the chain against the same block without it (one more GEMM launch, x1 through memory) and against fp32 torch; ragged key
    counts; B * N from one 128-token tile per CU downwards; repeated launches bit-stable."""
    from adaface_dev_amd.ldm.modules import attention as A
    C, Cc = 320, 768
    blk = A.BasicTransformerBlock(C, 8, 40, context_dim=Cc).to(dev)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g) * p.shape[1] ** -0.5)
            elif "norm" in n and n.endswith("weight"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2 + 1)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    x, _, _ = _ln_inputs(B * N, C, seed=11)
    ctx = rnd((B, L, Cc), 12)
    monkeypatch.setattr(A, "XATTN_FUSE_MIN_TOKENS", 0)

    def run(chain):
        monkeypatch.setattr(A, "CHAIN_XATTN", chain)
        return blk.hip(x.to(dev), B, N, ctx.to(dev), None)
    y0 = run(False)
    calls = {"n": 0}
    from adaface_dev_amd import ops
    real = ops.xattn_chain

    def spy(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(ops, "xattn_chain", spy)
    y1 = run(True)
    assert calls["n"] == 1, "the chained launch was expected to run"
    for _ in range(3):
        assert torch.equal(run(True), y1)
    assert rel_l2(y1.float().cpu().numpy(), y0.float().cpu().numpy()) < TOL
    # fp32 reference of the block
    P = {n: p.detach().float().cpu() for n, p in blk.named_parameters()}
    xf = x.float()

    def attn(xq, kv, pre):
        q = (xq @ P[pre + "to_q.weight"].t()).reshape(B, -1, 8, 40).permute(0, 2, 1, 3)
        kk = (kv @ P[pre + "to_k.weight"].t()).reshape(B, -1, 8, 40).permute(0, 2, 1, 3)
        vv = (kv @ P[pre + "to_v.weight"].t()).reshape(B, -1, 8, 40).permute(0, 2, 1, 3)
        o = torch.softmax(q @ kk.transpose(-1, -2) * 40 ** -0.5, dim=-1) @ vv
        return o.permute(0, 2, 1, 3).reshape(-1, C) @ P[pre + "to_out.0.weight"].t() + P[pre + "to_out.0.bias"]
    ln = lambda v, n: F.layer_norm(v, (C,), P[n + ".weight"], P[n + ".bias"], 1e-5)
    h = ln(xf, "norm1").reshape(B, N, C)
    x1 = attn(h, h, "attn1.") + xf
    x2 = attn(ln(x1, "norm2").reshape(B, N, C), ctx.float(), "attn2.") + x1
    hp = ln(x2, "norm3") @ P["ff.net.0.proj.weight"].t() + P["ff.net.0.proj.bias"]
    a, gte = hp.chunk(2, dim=-1)
    ref = (a * F.gelu(gte)) @ P["ff.net.2.weight"].t() + P["ff.net.2.bias"] + x2
    assert rel_l2(y1.float().cpu().numpy(), ref.numpy()) < 2 * TOL      # three chained fp16 sub-blocks


@pytest.mark.parametrize("B,N,L,heads,d", [(1, 4096, 4096, 2, 40), (2, 1024, 1024, 8, 40), (3, 512, 320, 4, 40), (1, 1536, 256, 3, 40)])
def test_self_attention_eight_wave_workgroups_bit_identical(dev, B, N, L, heads, d, monkeypatch):
    """Round 6: the two-chain self-attention kernel with 8-wave workgroups (512 queries, one workgroup per CU, the SIMD partners at a static
    raised priority so that they run out of phase: AF_ATTN_NW=8) against the 4-wave form on the same inputs: the same per-chain arithmetic in the
    same order -> bit-identical outputs and log-sum-exps, incl. the lazy-reference rescale branch (a spiked key in the last stage) -- and repeated
    launches stay identical (the partners share K / V^T tiles through LDS behind one barrier per stage, as before)."""
    from adaface_dev_amd import ops
    C = heads * d
    q, k, v = rnd((B, N, C), 1), rnd((B, L, C), 2, 0.5), rnd((B, L, C), 3)
    k[0, L - 3] = (q[0, 5] * 4).half()
    k[0, 130] = (q[0, 40] * 4).half()
    vt = v.permute(0, 2, 1).contiguous()
    run = lambda: ops.attention(q.reshape(B * N, C).to(dev), k.reshape(B * L, C).to(dev), vt.to(dev), B=B, Nq=N, L=L, heads=heads, d=d, ldq=C, ldk=C,
                                want_lse=True)
    monkeypatch.setenv("AF_ATTN_NW", "4")
    o4, l4 = run()
    monkeypatch.setenv("AF_ATTN_NW", "8")
    o8, l8 = run()
    assert torch.equal(o4, o8) and torch.equal(l4, l8)
    for _ in range(5):
        o, l = run()
        assert torch.equal(o, o8) and torch.equal(l, l8)
    assert rel_l2(o8.float().cpu().reshape(B, N, C).numpy(), _attn_ref(q, k, v, heads).numpy()) < TOL


def test_attention_scores_capture(dev):
    from adaface_dev_amd import ops
    B, N, L, heads, d = 2, 100, 77, 8, 40
    C = heads * d
    q, k = rnd((B, N, C), 1), rnd((B, L, C), 2)
    score, prob = ops.attention_scores(q.reshape(B * N, C).to(dev), k.reshape(B * L, C).to(dev), B=B, Nq=N, L=L, heads=heads, d=d)
    _, attn, sc = __import__("oracle.unet_oracle", fromlist=["x"]).attention_core(q.float(), k.float(), k.float(), heads, None, True)
    assert rel_l2(score.cpu().numpy(), sc.numpy()) < 1e-5
    assert rel_l2(prob.cpu().numpy(), attn.numpy()) < 1e-5


# ------------------------------------------------------------------------------- element-wise
def test_timestep_embedding(dev, golden_dir):
    import os
    from adaface_dev_amd import ops
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    out = ops.timestep_embedding(torch.from_numpy(g["temb_t"]).to(dev), 320)
    # fp16 output of values in [-1, 1]: absolute tolerance 1e-3 (args up to 999 rad in fp32)
    assert np.abs(out.float().cpu().numpy() - g["temb_320"]).max() < 1.5e-3


def test_layout_roundtrip_and_silu(dev):
    from adaface_dev_amd import ops
    x = torch.randn(2, 5, 7, 9, generator=torch.Generator().manual_seed(1))
    y = ops.nchw_f32_to_nhwc_f16(x.to(dev), 8)
    back = ops.nhwc_f16_to_nchw_f32(y, 5)
    assert torch.equal(back.cpu(), x.half().float())
    s = ops.silu(y)
    assert rel_l2(s.float().cpu().numpy(), F.silu(y.float().cpu()).numpy()) < TOL


def test_cfg_ddim_step_and_q_sample(dev):
    from adaface_dev_amd import ops
    from oracle import diffusion_oracle as D
    g = torch.Generator().manual_seed(0)
    x, ec, eu = torch.randn(4, 4, 8, 8, generator=g), torch.randn(4, 4, 8, 8, generator=g), torch.randn(4, 4, 8, 8, generator=g)
    a_t, a_prev = 0.5215, 0.5552
    xp, p0 = ops.cfg_ddim_step(torch.cat([ec, eu]).to(dev), x.to(dev), 4.0, a_t, a_prev, True)
    rx, rp = D.ddim_update(x, D.cfg_combine(ec, eu, 4.0), a_t, a_prev)
    assert rel_l2(xp.cpu().numpy(), rx.numpy()) < 1e-6 and rel_l2(p0.cpu().numpy(), rp.numpy()) < 1e-6
    xp1, _ = ops.cfg_ddim_step(ec.to(dev), x.to(dev), 1.0, a_t, a_prev, False)
    rx1, _ = D.ddim_update(x, ec, a_t, a_prev)
    assert rel_l2(xp1.cpu().numpy(), rx1.numpy()) < 1e-6
    tabs = D.register_schedule(D.make_beta_schedule_linear())
    t = torch.tensor([0, 10, 500, 999])
    xt = ops.q_sample(x.to(dev), ec.to(dev), torch.from_numpy(tabs["sqrt_alphas_cumprod"])[t].to(dev),
                      torch.from_numpy(tabs["sqrt_one_minus_alphas_cumprod"])[t].to(dev))
    assert rel_l2(xt.cpu().numpy(), D.q_sample(tabs, x, t, ec).numpy()) < 1e-6


def test_error_reporting(dev):
    """Bad arguments come back as RuntimeError with the library's message (never abort, never fall back)."""
    from adaface_dev_amd import ops
    a = rnd((16, 24), 1).to(dev)
    pw = ops.pack_matrix(rnd((8, 24), 2), None, dev)
    pw.N = 6  # not a multiple of 4
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.gemm(a, pw)
    with pytest.raises(RuntimeError, match="head dim"):
        ops.attention(a, a, a.reshape(1, 24, 16), B=1, Nq=16, L=16, heads=2, d=12, ldq=24, ldk=24)


def test_prefetch_reads_without_side_effects(dev):
    """af_prefetch only reads: the range is unchanged, odd sizes / empty ranges / misaligned pointers are handled."""
    from adaface_dev_amd import _lib
    L = _lib.lib()
    x = torch.arange(1 << 20, dtype=torch.float16, device=dev)
    ref = x.clone()
    st = torch.cuda.current_stream().cuda_stream
    assert L.af_prefetch(x.data_ptr(), x.numel() * 2, st) == 0
    assert L.af_prefetch(x.data_ptr(), 100, st) == 0 and L.af_prefetch(x.data_ptr(), 0, st) == 0
    assert L.af_prefetch(x.data_ptr() + 2, 1024, st) < 0 and b"aligned" in L.af_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x, ref)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 7, 8, 11, 12, 13])
def test_split_k_in_kernel_reduction_is_bit_identical_and_repeatable(dev, tile, monkeypatch):
    """Split-K with the reduction inside the GEMM launch (af_gemm_desc.splitk_fused: the last-arriving K-slice of every output tile sums the
    slabs in slice order and runs the epilogue) against the two-launch form: bit-identical outputs (same summation order), for
    every epilogue input (bias, per-batch row bias, SiLU, residual), ragged M, 2 - 4 slices -- and again over many back-to-back
    launches of different shapes sharing the workspace (the arrival counters must come back to zero every time; a stale or lost
    hand-off would show as a wrong tile)."""
    from adaface_dev_amd import ops, rng
    shapes = [(1000, 320, 1280, 2), (4096, 640, 1280, 3), (616, 640, 2560, 4), (2048, 1280, 1280, 2), (130, 320, 640, 2)]
    if tile in (4, 7, 11, 13):
        shapes = [sh for sh in shapes if sh[1] % 320 == 0]
    cases = []
    for M, N, K, sp in shapes:
        a = rng.synth_input(f"skf.a{M}", (M, K), seed=5).half().to(dev)
        pw = ops.pack_matrix(rng.synth_input(f"skf.w{N}x{K}", (N, K), seed=5) * K ** -0.5, rng.synth_input(f"skf.b{N}", (N,), seed=5), dev)
        res = rng.synth_input(f"skf.r{M}x{N}", (M, N), seed=5).half().to(dev)
        rowb = rng.synth_input(f"skf.rb{N}", (4, N), seed=5).half().to(dev)
        cases.append((a, pw, res, rowb, sp, -(-M // 4)))

    def run_all(fused_max):
        monkeypatch.setattr(ops, "SPLITK_FUSED_MAX", fused_max)
        return [ops.gemm(a, pw, residual=res, rowbias=rowb, rows_per_batch=rpb, act=ops.AF_ACT_SILU, tile=tile, splits=sp)
                for a, pw, res, rowb, sp, rpb in cases]
    want = run_all(0)                                   # two launches: GEMM slabs + af_splitk_reduce_kernel
    for rep in range(25):
        got = run_all(4)
        for g, w, c in zip(got, want, cases):
            assert torch.equal(g, w), (tile, rep, tuple(c[0].shape), c[4])
    ws = ops._splitk_workspace(dev)
    assert int(ws[-(ops._lib.AF_SPLITK_COUNTER_BYTES // 4):].view(torch.int32).abs().sum()) == 0      # counters restored
    # 3x3 convolution with two sources and split-K through the same path
    x = rng.synth_input("skf.x", (2, 16, 16, 640), seed=5).half().to(dev)
    x2 = rng.synth_input("skf.x2", (2, 16, 16, 320), seed=5).half().to(dev)
    pc = ops.pack_conv3x3(rng.synth_input("skf.wc", (640, 960, 3, 3), seed=5) * (960 * 9) ** -0.5, rng.synth_input("skf.bc", (640,), seed=5), dev)
    if tile in (1, 2):
        # round 5: chosen by SIZE (ops.SPLITK_FUSED_BYTES) for the register-staged tiles, up to 16 slices: the training legs' small outputs
        small = []
        for M, N, K, sp in [(256, 256, 2304, 6), (64, 512, 4608, 8), (1024, 128, 1152, 3), (77, 768, 3072, 12), (256, 1280, 5120, 16)]:
            a = rng.synth_input(f"sks.a{M}x{K}", (M, K), seed=6).half().to(dev)
            pw = ops.pack_matrix(rng.synth_input(f"sks.w{N}x{K}", (N, K), seed=6) * K ** -0.5, rng.synth_input(f"sks.b{N}", (N,), seed=6), dev)
            small.append((a, pw, sp))
        monkeypatch.setattr(ops, "SPLITK_FUSED_MAX", 0)
        monkeypatch.setattr(ops, "SPLITK_FUSED_BYTES", 0)
        want_small = [ops.gemm(a, pw, tile=tile, splits=sp) for a, pw, sp in small]
        monkeypatch.setattr(ops, "SPLITK_FUSED_BYTES", 4 << 20)
        for rep in range(10):
            for (a, pw, sp), w in zip(small, want_small):
                assert torch.equal(ops.gemm(a, pw, tile=tile, splits=sp), w), (tile, rep, tuple(a.shape), sp)
        assert int(ws[-(ops._lib.AF_SPLITK_COUNTER_BYTES // 4):].view(torch.int32).abs().sum()) == 0
        monkeypatch.setattr(ops, "SPLITK_FUSED_BYTES", 0)
    monkeypatch.setattr(ops, "SPLITK_FUSED_MAX", 0)
    wc = ops.conv3x3(x, pc, x2=x2, tile=tile, splits=3)
    monkeypatch.setattr(ops, "SPLITK_FUSED_MAX", 4)
    for rep in range(5):
        assert torch.equal(ops.conv3x3(x, pc, x2=x2, tile=tile, splits=3), wc), rep


@pytest.mark.parametrize("B,H,W,c1,c2,cout,up,splits,extras,variant", [
    (1, 64, 64, 128, 0, 128, False, 1, True, 2), (2, 32, 32, 64, 0, 256, False, 1, False, 2), (1, 16, 16, 128, 0, 128, False, 2, True, 2), (4, 8, 8, 128, 64, 128, False, 1, True, 2),
    (1, 128, 128, 64, 0, 128, False, 1, True, 3), (2, 128, 128, 128, 0, 256, False, 1, False, 3), (1, 256, 256, 64, 0, 128, False, 1, True, 3),
    (1, 64, 64, 128, 0, 128, True, 1, False, 3), (1, 144, 176, 64, 0, 128, False, 1, True, 3), (1, 128, 128, 64, 64, 512, False, 1, True, 3),
    (2, 64, 32, 64, 0, 128, True, 1, True, 2), (3, 128, 144, 64, 0, 128, False, 1, True, 3)])
def test_conv3x3_tile14_256x128_tile_and_patches(dev, B, H, W, c1, c2, cout, up, splits, extras, variant):
    """Round 6: the halo-resident kernel on the VAE's channel counts and image sizes (model.py:136-175, 536-567) -- a 256 x 128 tile where N is no
    160-multiple (af_gemm_halo_variant 2), and 16 x 16-pixel patches on images wider than 64 pixels (variant 3: borders on all four sides of a patch,
    nearest-x2 sources, two sources, non-square images) -- against torch conv2d in fp32 and the tap-by-tap tile."""
    from adaface_dev_amd import ops, _lib
    import ctypes as C
    x1 = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 2) if c2 else None
    cin = c1 + c2
    w = rnd((cout, cin, 3, 3), 3, (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(4))
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias, padding=1)
    Ho, Wo = ref.shape[2:]
    rowb = res = None
    if extras:
        rowb = rnd((B, cout), 5)
        res = rnd((B, Ho, Wo, cout), 6)
        ref = ref + rowb.float()[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    d = _lib.GemmDesc()
    d.taps, d.c1, d.c2, d.N, d.B, d.H, d.W, d.Ho, d.Wo, d.M, d.upsample, d.stride = 9, c1, c2, cout, B, H, W, Ho, Wo, B * Ho * Wo, int(up), 1
    d.K, d.kpad = 9 * cin, 9 * cin
    assert _lib.lib().af_gemm_halo_variant(C.byref(d)) == variant
    d.splits = splits
    assert ops.conv_halo_eligible(d)
    pw = ops.pack_conv3x3(w, bias, dev)
    kw = dict(x2=None if x2 is None else x2.to(dev), upsample=up, rowbias=None if rowb is None else rowb.to(dev), residual=None if res is None else res.to(dev))
    out = ops.conv3x3(x1.to(dev), pw, tile=14, splits=splits, **kw)
    assert rel_l2(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    out8 = ops.conv3x3(x1.to(dev), pw, tile=8, splits=1, **kw)
    assert rel_l2(out.float().cpu().numpy(), out8.float().cpu().numpy()) < 2e-3
    assert (out.float() - out8.float()).abs().max().item() < 0.05 * max(1.0, out8.float().abs().max().item())      # no misplaced patch / row anywhere
    for _ in range(3):
        assert torch.equal(ops.conv3x3(x1.to(dev), pw, tile=14, splits=splits, **kw), out)
