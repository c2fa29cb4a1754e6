"""Pin oracle/vae_oracle.py against outputs of the REFERENCE's VAE Decoder (tests/golden/vae.npz): reduced width in full, SD-1.5
size by probes; and the host mirror's state-dict layout.  CPU only."""
import os

import numpy as np
import torch

from conftest import GOLDEN, rel_l2
from adaface_dev_amd import rng
from oracle import vae_oracle as VO

VAE_SMALL = dict(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=128, z_channels=4)


def vae_test_masks(r):
    """The masks tests/golden/gen_golden.py::vae_test_masks built the masked-encoder vectors with."""
    fg = torch.zeros(2, 1, r, r)
    fg[0, :, r // 4: 3 * r // 4, r // 4: 3 * r // 4] = 1
    fg[1, :, :, : r // 2] = 1
    aug = torch.ones(2, 1, r, r)
    aug[1, :, : r // 8] = 0
    return fg, aug


def _probes(t):
    f = t.detach().float().reshape(-1)
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761) % f.numel()
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()]).astype(np.float32)


def decoder_state_dict(cfg, seed=90):
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import Decoder
    with torch.device("meta"):
        m = Decoder(**cfg)
    return {"decoder." + n: rng.synth_tensor("decoder." + n, p.shape, seed=seed) for n, p in m.named_parameters()}


def test_vae_decoder_oracle_vs_reference_reduced_width():
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    sd = decoder_state_dict(VAE_SMALL)
    z = rng.synth_input("vae.z.small", (2, 4, 16, 16), seed=90)
    with torch.no_grad():
        y = VO.decoder(sd, z)
    assert rel_l2(y.numpy(), g["small_out"]) < 1e-5


def test_vae_decoder_oracle_vs_reference_sd15_size():
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    sd = decoder_state_dict(dict(VAE_SMALL, ch=128, resolution=256))
    assert sum(v.numel() for v in sd.values()) == 49_490_179                       # SD-1.5 KL-f8 decoder
    z = rng.synth_input("vae.z.full", (1, 4, 64, 64), seed=90)
    torch.set_num_threads(8)
    with torch.no_grad():
        y = VO.decoder(sd, z)
    assert np.allclose(_probes(y), g["full_probes"], rtol=1e-3, atol=1e-4)
    assert rel_l2(y[0, :, 200:264, 300:364].numpy(), g["full_crop"]) < 1e-4


def test_host_mirror_layout_and_no_cpu_fallback():
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
    m = AutoencoderKLDecoder()
    names = set(m.state_dict().keys())
    for k in ("post_quant_conv.weight", "decoder.conv_in.weight", "decoder.mid.block_1.norm1.weight", "decoder.mid.attn_1.q.weight",
              "decoder.mid.attn_1.proj_out.bias", "decoder.up.3.block.2.conv2.weight", "decoder.up.1.block.0.nin_shortcut.weight",
              "decoder.up.1.upsample.conv.weight", "decoder.norm_out.weight", "decoder.conv_out.weight"):
        assert k in names, k
    assert "decoder.up.0.upsample.conv.weight" not in names
    try:
        m.decode(torch.zeros(1, 4, 8, 8))
    except RuntimeError as e:
        assert "MI355X" in str(e) or "no CPU" in str(e)
    else:
        raise AssertionError("the VAE decoder ran on the CPU: there must be no fallback path")


def test_vae_encoder_oracle_vs_reference_reduced_width():
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import Encoder
    g = np.load(os.path.join(GOLDEN, "vae.npz"))
    with torch.device("meta"):
        m = Encoder(**dict(VAE_SMALL, double_z=True))
    sd = {"encoder." + n: rng.synth_tensor("encoder." + n, p.shape, seed=90) for n, p in m.named_parameters()}
    img = rng.synth_input("vae.img.small", (2, 3, 128, 128), seed=90)
    with torch.no_grad():
        y = VO.encoder(sd, img)
    assert y.shape == (2, 8, 16, 16) and rel_l2(y.numpy(), g["enc_small_out"]) < 1e-5
    # masked mid-block attention (model.py:191-232), pinned on the reference Encoder's own masked outputs
    fg, aug = vae_test_masks(128)
    with torch.no_grad():
        ym = VO.encoder(sd, img, mask={"fg_mask": fg, "aug_mask": aug})
        yf = VO.encoder(sd, img, mask={"fg_mask": fg, "aug_mask": None})
        yn = VO.encoder(sd, img, mask={"fg_mask": None, "aug_mask": aug})           # no fg mask: the reference ignores aug
    assert rel_l2(ym.numpy(), g["enc_small_masked_out"]) < 1e-5
    assert rel_l2(yf.numpy(), g["enc_small_fgonly_out"]) < 1e-5
    assert rel_l2(ym.numpy(), g["enc_small_out"]) > 1e-2 and torch.equal(yn, y)
