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
