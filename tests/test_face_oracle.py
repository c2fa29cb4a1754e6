"""Pin oracle/face_oracle.py against outputs of the REFERENCE's ResNetFace-18 IR-SE module (tests/golden/arcface.npz) and
check the host mirror's module tree is state-dict compatible with the reference's.  CPU only."""
import os

import numpy as np
import torch

from conftest import GOLDEN, rel_l2
from adaface_dev_amd import rng
from oracle import face_oracle as FO


def _probes(t):
    f = t.detach().float().reshape(-1)
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761) % f.numel()
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()]).astype(np.float32)


def _model_and_sd():
    from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
    m = resnet_face18(use_se=True).eval()
    sd = rng.synth_face_state_dict(m.state_dict(), seed=50)
    return m, sd


def test_face_oracle_vs_reference_outputs():
    g = np.load(os.path.join(GOLDEN, "arcface.npz"))
    _, sd = _model_and_sd()
    x = rng.synth_input("face.x", (3, 1, 128, 128), seed=50)
    with torch.no_grad():
        y, feats = FO.resnet_face18(sd, x, return_features=True)
    assert rel_l2(y.numpy(), g["emb"]) < 1e-5
    for i, f in enumerate(feats):
        assert np.allclose(_probes(f), g[f"layer{i + 1}_probes"], rtol=1e-4, atol=1e-5), i
    assert rel_l2(feats[3].numpy(), g["layer4"]) < 1e-5


def test_host_mirror_state_dict_layout_and_no_cpu_fallback():
    m, sd = _model_and_sd()
    names = set(m.state_dict().keys())
    for k in ("conv1.weight", "bn1.running_var", "prelu.weight", "layer1.0.bn0.weight", "layer1.0.se.fc.0.weight", "layer1.0.se.fc.1.weight",
              "layer2.0.downsample.0.weight", "layer2.0.downsample.1.running_mean", "layer4.1.conv2.weight", "bn4.bias", "fc5.weight",
              "bn5.num_batches_tracked"):
        assert k in names, k
    assert m.fc5.weight.shape == (512, 512 * 8 * 8) and m.layer2[0].downsample[0].weight.shape == (128, 64, 1, 1)
    m.load_state_dict(sd)                                   # strict: every key present with the right shape
    try:
        m(torch.zeros(1, 1, 128, 128))
    except RuntimeError as e:
        assert "no CPU" in str(e) or "MI355X" in str(e)
    else:
        raise AssertionError("ResNetFace ran on the CPU: there must be no fallback path")
    try:
        m.train()(torch.zeros(1, 1, 128, 128))
    except NotImplementedError:
        pass
    else:
        raise AssertionError("training-mode BatchNorm must not be silently approximated")
