"""SD VAE decoder on a real MI355X (`pytest -m gpu`): row softmax vs torch, the decoder against the REFERENCE module's outputs
(tests/golden/vae.npz; reduced width in full, SD-1.5 size by probes + a crop) and `decode` (post_quant_conv + decoder) against the CPU
oracle.  fp16 activations across ~30 convolutions: 1e-2 rel-L2."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from test_vae_oracle import VAE_SMALL, _probes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("rows,L", [(37, 64), (300, 256), (64, 1024), (130, 4096), (5, 1800)])
def test_softmax_rows_vs_torch(dev, rows, L):
    from adaface_dev_amd import ops
    x = (torch.randn(rows, L, generator=torch.Generator().manual_seed(L)) * 4).half()
    y = ops.softmax_rows(x.to(dev))
    ref = torch.softmax(x.float(), dim=1)
    assert rel_l2(y.float().cpu().numpy(), ref.numpy()) < 2e-3
    assert float((y.float().sum(1) - 1).abs().max()) < 5e-3


@pytest.mark.parametrize("rows,L", [(37, 64), (64, 1024), (130, 4096), (5, 1800)])
def test_softmax_rows_bwd_vs_torch(dev, rows, L):
    from adaface_dev_amd import ops
    g = torch.Generator().manual_seed(L + 1)
    p = torch.softmax(torch.randn(rows, L, generator=g) * 3, dim=1).half()
    dp = torch.randn(rows, L, generator=g).half()
    ds = ops.softmax_rows_bwd(p.to(dev), dp.to(dev))
    pf, df = p.float(), dp.float()
    ref = pf * (df - (pf * df).sum(1, keepdim=True))
    assert rel_l2(ds.float().cpu().numpy(), ref.numpy()) < 2e-3


def _decoder(cfg, dev):
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import Decoder
    m = Decoder(**cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(rng.synth_tensor("decoder." + n, p.shape, seed=90))
    return m.to(dev).eval()


def test_vae_decoder_reduced_width_vs_reference(dev):
    from adaface_dev_amd import rng
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    m = _decoder(VAE_SMALL, dev)
    z = rng.synth_input("vae.z.small", (2, 4, 16, 16), seed=90)
    with torch.no_grad():
        y = m(z.to(dev))
    assert y.shape == (2, 3, 128, 128) and y.dtype == torch.float32
    e = rel_l2(y.cpu().numpy(), g["small_out"])
    print(f"VAE decoder (reduced width) rel-L2 vs reference: {e:.3e}")
    assert e < 1e-2


def test_vae_decoder_sd15_size_vs_reference_and_decode_vs_oracle(dev):
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
    from oracle import vae_oracle as VO
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    ae = AutoencoderKLDecoder()
    with torch.no_grad():
        for n, p in ae.named_parameters():
            p.copy_(rng.synth_tensor(n, p.shape, seed=90))
    ae = ae.to(dev).eval()
    z = rng.synth_input("vae.z.full", (1, 4, 64, 64), seed=90)
    with torch.no_grad():
        y = ae.decoder(z.to(dev))
    assert y.shape == (1, 3, 512, 512)
    e = rel_l2(y[0, :, 200:264, 300:364].cpu().numpy(), g["full_crop"])
    print(f"VAE decoder (SD-1.5 size, 64x64 latent -> 512x512) crop rel-L2 vs reference: {e:.3e}")
    assert e < 1e-2
    assert np.allclose(_probes(y.cpu()), g["full_probes"], rtol=5e-2, atol=2e-2)
    # decode = post_quant_conv + decoder, batch 2, small latent, against the oracle on the same weights
    sd = {k: v.detach().float().cpu() for k, v in ae.state_dict().items()}
    z2 = rng.synth_input("vae.z2", (2, 4, 16, 16), seed=91)
    with torch.no_grad():
        img = ae.decode(z2.to(dev)).cpu()
        ref = VO.decode(sd, z2)
    assert rel_l2(img.numpy(), ref.numpy()) < 1e-2


def test_vae_encoder_vs_reference_and_encode_decode_roundtrip_shapes(dev):
    """Encoder (asymmetric-padding stride-2 Downsample = +1 tap shift in the conv loader) against the REFERENCE module's output;
    AutoencoderKL.encode -> (mean, logvar) against the oracle; decode(encode(x).mean) has the image's shape."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKL, Encoder
    from oracle import vae_oracle as VO
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    e = Encoder(**dict(VAE_SMALL, double_z=True))
    with torch.no_grad():
        for n, p in e.named_parameters():
            p.copy_(rng.synth_tensor("encoder." + n, p.shape, seed=90))
    e = e.to(dev).eval()
    img = rng.synth_input("vae.img.small", (2, 3, 128, 128), seed=90)
    with torch.no_grad():
        y = e(img.to(dev))
    err = rel_l2(y.cpu().numpy(), g["enc_small_out"])
    print(f"VAE encoder (reduced width) rel-L2 vs reference: {err:.3e}")
    assert y.shape == (2, 8, 16, 16) and err < 1e-2
    ae = AutoencoderKL(dict(VAE_SMALL, double_z=True))
    with torch.no_grad():
        for n, p in ae.named_parameters():
            p.copy_(rng.synth_tensor(n, p.shape, seed=92))
    sd = {k: v.detach().float().clone() for k, v in ae.state_dict().items()}
    ae = ae.to(dev).eval()
    mean, logvar = ae.encode(img.to(dev))
    rm, rl = VO.encode(sd, img)
    assert rel_l2(mean.cpu().numpy(), rm.numpy()) < 1e-2 and rel_l2(logvar.cpu().numpy(), rl.numpy()) < 1e-2
    rec = ae.decode(mean)
    assert rec.shape == img.shape and bool(torch.isfinite(rec).all())


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 64, 64), (1, 32, 30, 128, 96), (2, 9, 11, 32, 64)])
def test_conv3x3_tap_shift_is_asymmetric_padding(dev, B, H, W, cin, cout):
    """ops.conv3x3(stride=2, tap_shift=1) == F.conv2d(F.pad(x, (0,1,0,1)), w, stride=2) incl. odd sizes."""
    import torch.nn.functional as F
    from adaface_dev_amd import ops, rng
    x = rng.synth_input("ts.x", (B, cin, H, W), seed=93)
    w = rng.synth_input("ts.w", (cout, cin, 3, 3), seed=93, scale=(cin * 9) ** -0.5)
    b = rng.synth_input("ts.b", (cout,), seed=93, scale=0.1)
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    # the op derives Ho from the symmetric formula; for odd sizes the reference's padded conv gives floor((H + 1 - 3) / 2) + 1
    ref = F.conv2d(F.pad(xh.float().cpu().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.half().float(), b, 2, 0)
    y = ops.conv3x3(xh, ops.pack_conv3x3(w, b, dev), stride=2, tap_shift=1)
    y = y[:, :ref.shape[2], :ref.shape[3]]
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < 2e-3


def test_encoder_masked_attention_vs_reference(dev):
    """fg / aug masks zero the post-softmax weights between foreground and background tokens (model.py:191-232): af_mask_pairs on
    the [N, N] probabilities, against the REFERENCE Encoder's masked outputs; fractional masks against the oracle."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import Encoder
    from oracle import vae_oracle as VO
    from test_vae_oracle import vae_test_masks
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    e = Encoder(**dict(VAE_SMALL, double_z=True))
    with torch.no_grad():
        for n, p in e.named_parameters():
            p.copy_(rng.synth_tensor("encoder." + n, p.shape, seed=90))
    sd = {"encoder." + n: p.detach().clone() for n, p in e.named_parameters()}
    e = e.to(dev).eval()
    img = rng.synth_input("vae.img.small", (2, 3, 128, 128), seed=90)
    fg, aug = vae_test_masks(128)
    with torch.no_grad():
        ym = e(img.to(dev), {"fg_mask": fg.to(dev), "aug_mask": aug.to(dev)}).cpu().numpy()
        yf = e(img.to(dev), {"fg_mask": fg, "aug_mask": None}).cpu().numpy()
        yn = e(img.to(dev), {"fg_mask": None, "aug_mask": aug}).cpu().numpy()
    errs = rel_l2(ym, g["enc_small_masked_out"]), rel_l2(yf, g["enc_small_fgonly_out"]), rel_l2(yn, g["enc_small_out"])
    print("masked VAE encoder rel-L2 vs reference (fg+aug, fg only, aug only):", ["%.2e" % v for v in errs])
    assert max(errs) < 1e-2 and rel_l2(ym, g["enc_small_out"]) > 1e-2
    # fractional masks: a pixel with 0 < fg < 1 is both foreground and background (bit mask 3)
    fgf = torch.rand(2, 1, 128, 128, generator=torch.Generator().manual_seed(3)).round(decimals=0) * 0.5 + fg * 0.5
    with torch.no_grad():
        want = VO.encoder(sd, img, mask={"fg_mask": fgf, "aug_mask": aug})
        got = e(img.to(dev), {"fg_mask": fgf, "aug_mask": aug}).cpu().numpy()
    assert rel_l2(got, want.numpy()) < 1e-2


@pytest.mark.parametrize("size", ["reduced", "sd15"])
def test_vae_decode_input_gradient_vs_oracle_autograd(dev, size):
    """d(loss)/dz through the frozen decoder (decode_first_stage_with_grad, ddpm.py:899-908 -- what the ArcFace alignment loss
    back-propagates into the x0 prediction) against torch autograd through the CPU oracle on the same weights: reduced width on a
    16x16 latent, the SD-1.5 decoder on a 16x16 latent (128x128 image; the mid attention over 256 tokens) and, SD-1.5 only, a gradient
    that lives in one 24x24 image window, as a face crop's does.  fp16 through ~30 convolutions each way: 2e-2 rel-L2."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
    from oracle import vae_oracle as VO
    cfg = dict(AutoencoderKLDecoder.SD15_DDCONFIG)
    kw = {}
    if size == "reduced":
        cfg.update({k: v for k, v in VAE_SMALL.items() if k in cfg})
        kw = dict(num_resolutions=len(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"])
    ae = AutoencoderKLDecoder(cfg)
    with torch.no_grad():
        for n, p in ae.named_parameters():
            p.copy_(rng.synth_tensor(n, p.shape, seed=92))
    ae = ae.to(dev).eval()
    sd = {k: v.detach().float().cpu() for k, v in ae.state_dict().items()}
    z = rng.synth_input("vae.gz", (2, 4, 16, 16), seed=92)
    up = 2 ** (len(cfg["ch_mult"]) - 1)
    wgt = rng.synth_input("vae.gw", (2, 3, 16 * up, 16 * up), seed=92)
    cases = [wgt]
    if size == "sd15":
        win = torch.zeros_like(wgt)
        win[:, :, 40:64, 70:94] = wgt[:, :, 40:64, 70:94] * 1e-4             # a small, local gradient: exercises the power-of-two rescale
        cases.append(win)
    for w in cases:
        zr = z.clone().requires_grad_(True)
        ref_img = VO.decode(sd, zr, **kw)
        (ref_img * w).sum().backward()
        zd = z.to(dev).requires_grad_(True)
        img = ae.decode(zd)
        assert img.requires_grad and rel_l2(img.detach().cpu().numpy(), ref_img.detach().numpy()) < 1e-2
        (img * w.to(dev)).sum().backward()
        e = rel_l2(zd.grad.cpu().numpy(), zr.grad.numpy())
        print(f"VAE decode ({size}) latent-gradient rel-L2 vs oracle autograd: {e:.3e}")
        assert e < 2e-2
    assert all(p.grad is None for p in ae.parameters())
    with torch.no_grad():
        assert not ae.decode(z.to(dev)).requires_grad
