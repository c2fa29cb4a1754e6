#!/usr/bin/env python
"""VAE decode time of one image batch (BASELINE configs[1]: 4 latents 64x64 -> 4 x 3 x 512 x 512), hipGraph replay.
    python tools/bench_vae.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from adaface_dev_amd import ops, rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    ae = AutoencoderKLDecoder()
    with torch.no_grad():
        for n, p in ae.named_parameters():
            p.copy_(rng.synth_tensor(n, p.shape, seed=90))
    ae = ae.to(dev).eval()
    z = rng.synth_input("vae.bench", (B, 4, 64, 64), seed=1).to(dev)
    with torch.no_grad():
        for _ in range(2):
            img = ae.decode(z)
        torch.cuda.synchronize()
        ops.prof_reset()
        ops.prof_enable(True)
        ae.decode(z)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        from adaface_dev_amd import _lib
        fam = {n: ops.prof_read(f) for n, f in (("gemm", _lib.AF_FAM_GEMM), ("gnorm", _lib.AF_FAM_GNORM), ("elem", _lib.AF_FAM_ELEM))}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            img = ae.decode(z)
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tf = 2.51e12 * B / (ms * 1e-3) / 1e12          # 2.51 TFLOP per decoded image
    print(f"VAE decode batch {B}: {ms:.2f} ms ({tf:.0f} TFLOP/s over 2.51 TFLOP/img), finite={bool(torch.isfinite(img).all())}, families (launches, ms): {fam}")


if __name__ == "__main__":
    main()
