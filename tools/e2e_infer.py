#!/usr/bin/env python
"""BASELINE configs[1] end to end at full size through the AdaFaceWrapper surface (seeded random weights): face IDs -> Arc2Face
image prompt -> AdaFace token embeddings -> token table -> rewritten prompt -> CLIP-L text encoder -> 50 DDIM steps with CFG on the
SD-1.5 U-Net (batch 4 + 4) -> VAE decoder -> 4 PIL images.  Prints the time of each phase (eager launches; bench.py measures the
denoise step under hipGraph replay).      python tools/e2e_infer.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.adaface_wrapper import AdaFaceWrapper
    from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    w = AdaFaceWrapper(device=dev, num_inference_steps=steps)
    rng.load_synth_weights(w.text_encoder, seed=60)
    rng.load_synth_weights(w.id2ada_prompt_encoder.text_to_image_prompt_encoder, seed=61)
    rng.load_synth_weights(w.id2ada_prompt_encoder.subj_basis_generator.prompt2token_proj, seed=62)
    rng.load_synth_weights(w.ldm.model.diffusion_model, seed=0)
    vae = AutoencoderKLDecoder()
    with torch.no_grad():
        for n, p in vae.named_parameters():
            p.copy_(rng.synth_tensor(n, p.shape, seed=90))
    w.vae = vae
    w = w.to(dev)
    w.vae.to(dev).eval()
    w.ldm.model.diffusion_model.prepare()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0

    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t

    ids = rng.synth_input("e2e.ids", (3, 512), seed=7).to(dev)
    noise = rng.synth_input("e2e.noise", (4, 4, 64, 64), seed=7)
    prompt = "portrait of a z, in a garden"
    w.vae = None
    for it in range(2):                                      # first pass packs weights / warms kernels
        embs, t_emb = timed(lambda: w.prepare_adaface_embeddings(None, face_id_embs=ids, avg_at_stage="id_emb"))
        (pe, ne, _, _), t_enc = timed(lambda: w.encode_prompt(prompt, device=dev))
        lat, t_ddim = timed(lambda: w(noise, None, prompt_embeds=(pe, ne), guidance_scale=6.0, out_image_count=4))
        img, t_vae = timed(lambda: vae.decode(lat / 0.18215))
    w.vae = vae
    imgs, t_all = timed(lambda: w(noise, prompt, guidance_scale=6.0, out_image_count=4))
    print(f"build+weights {t_build:.1f} s | AdaFace embeddings {t_emb * 1e3:.1f} ms | prompt encode (pos+neg) {t_enc * 1e3:.1f} ms | "
          f"{steps} DDIM steps (U-Net batch 8, eager) {t_ddim * 1e3:.1f} ms = {t_ddim / steps * 1e3:.2f} ms/step | VAE decode x4 {t_vae * 1e3:.1f} ms | "
          f"whole forward() incl. PIL {t_all * 1e3:.1f} ms | {len(imgs)} images {imgs[0].size}, latents finite={bool(torch.isfinite(lat).all())}, "
          f"embs {tuple(embs.shape)}")


if __name__ == "__main__":
    main()
