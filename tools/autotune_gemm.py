#!/usr/bin/env python
"""In-situ autotune of af_gemm's (tile, split-K) per shape on an MI355X.

Runs U-Net forwards at the benchmark shapes with a recorder hooked into ops._launch_gemm: the
first time a GEMM shape is seen, every candidate (tile in {128x128, 64x64 register-staged; 128x128, 128x320
LDS-DMA ring} x split-K in {1..16}) is timed on the live operands with HIP events, and the fastest is written to
adaface-dev_amd/tuning/gfx950_gemm.json.  Usage (GPU box):

    python tools/autotune_gemm.py [--batches 8,2] [--train] [--fresh]

--vae additionally runs the VAE decoder; --train additionally runs three Stage-1 distillation micro-batches (denoising steps 2, 3, 4: teacher batches 1-2, the
batched student pass at 4-6 samples, the whole backward and the CLIP encoders' forward/dgrad/wgrad GEMMs) so the
training shapes are tuned as well.  Existing entries are kept unless --fresh is given.
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def candidate_ok(d, tile, splits, _lib, ops):
    """Whether af_gemm takes descriptor ``d`` on (tile, splits): the scope rules of include/adaface_hip.h, as the tuners apply them."""
    geglu = d.act == _lib.AF_ACT_GEGLU
    nk = d.kpad // 64
    if tile == 14 and not ops.conv_halo_eligible(d):
        return False                        # halo-resident 3x3 kernel: its split-K slices are 64-channel chunks
    if tile >= 7 and (d.upsample not in (0, 1) or (d.upsample and tile not in (7, 8, 11, 12, 13, 14, 15)) or d.c1 % 64 or d.c2 % 64):
        return False                        # whole-line kernel: 64-multiples of channels; nearest-x2 upsample in the non-GEGLU tiles
    if tile in (16, 17) and (geglu or d.N % (128 if tile == 16 else 64) != 0 or d.c3 or d.c4 or (d.out_mode == _lib.AF_OUT_SPLIT_T and d.taps != 1)):
        return False                        # 64 x 128 / 128 x 64, four waves, three workgroups per CU: standard epilogue, no K tail
    if tile in (11, 13) and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T or d.N % 160 != 0):
        return False                        # 128 x 160, four waves (11: two workgroups per CU; 13: four-slot ring)
    if tile == 15 and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T or d.N % 128 != 0 or d.M < 16384):
        return False                        # 256 x 128, eight waves: narrow outputs over many rows
    if tile == 12 and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T):
        return False                        # 128 x 128 with a four-slot ring
    if tile == 7 and (d.N % (256 if geglu else 320) != 0):
        return False
    if tile == 8 and geglu and d.N % 128 != 0:
        return False                        # GEGLU on the 128 x 128 tile (two workgroups per CU): value | gate pairs of 64 output columns
    if tile in (9, 10) and (not geglu or d.N % (320 if tile == 9 else 256) != 0):
        return False
    if tile in (5, 6) and (d.taps != 1 or d.out_mode != 0 or d.M * d.N < 256 * 256 * 128 or splits > 1):
        return False                        # 256-row ring tiles: plain / GEGLU 1x1 GEMMs with at least ~128 tiles
    if (d.c3 or d.c4) and not (7 <= tile <= 15):
        return False                        # the K-concatenated 1x1 tail lives in the whole-line tiles and (round 6) the halo-resident kernel
    if d.ln_colsum and (tile < 7 or tile in (14, 15) or splits > 1 or (tile in (9, 10) and not geglu)):
        return False                        # folded LayerNorm: whole-line tiles, unsplit
    f32 = d.out_mode == _lib.AF_OUT_F32     # weight gradients: fp32 from the reduce pass, or unsplit from tiles 1 / 2
    if f32 and splits == 1 and tile > 2:
        return False
    if splits > 1 and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T or nk < 4 * splits):
        return False
    if tile == 14 and splits > (d.c1 + d.c2) // 64:
        return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,2")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--latents", default="64x64", help="comma-separated latent sizes HxW to run the U-Net at (64x64 = 512^2 images; 96x64 = 768x512)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--bench-train", action="store_true", help="tune the shapes of bench.py's Stage-1 training leg (97 context tokens, FFN adapters)")
    ap.add_argument("--bench-train2", action="store_true", help="tune the shapes of bench.py's Stage-2 (compositional distillation) leg")
    ap.add_argument("--vae", action="store_true", help="also tune the VAE decoder's shapes (batch 4 and 1, 64x64 latent)")
    ap.add_argument("--fresh", action="store_true", help="ignore the existing table instead of extending it")
    ap.add_argument("--cold-weights", action="store_true", help="time single launches after evicting the caches (weights cold, 1x1 activations re-read): what a GEMM meets inside the real step")
    ap.add_argument("--retune", action="store_true", help="re-time every shape this run meets; entries of shapes it does not meet are kept")
    ap.add_argument("--tiles", default="", help="comma-separated tile ids to time (default: all)")
    ap.add_argument("--trace", action="store_true", help="print every configuration before it is launched (to find one that faults)")
    ap.add_argument("--try-tile", type=int, default=0, help="for every shape this run meets that is ALREADY in the table: time the table's choice against this tile (splits 1, 2) and take the tile only where it is > 2 %% faster")
    ap.add_argument("--new-halo-forms", action="store_true", help="with --try-tile 14: only the shapes that the kernel's round-6 forms took into its scope (256 x 128 tile, 16 x 16-pixel patches: the VAE's shapes)")
    ap.add_argument("--retune-halo", action="store_true", help="re-time only the 3x3 shapes in the scope of the halo-resident kernel (tile 14)")
    ap.add_argument("--protect", default="", help="comma-separated logs of tools/autotune_instep.py: the shapes they decided inside a step are left alone")
    args = ap.parse_args()
    from adaface_dev_amd import SD15_UNET_CONFIG, _lib, ops, rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel

    dev = torch.device("cuda:0")
    L = _lib.lib()
    table, log = {}, []
    if not args.fresh and os.path.exists(ops._TUNE_PATH):
        with open(ops._TUNE_PATH) as f:
            table = {k: tuple(v) for k, v in json.load(f).items()}
        print(f"extending {len(table)} existing entries")

    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if args.cold_weights else None

    def timed(d, device, tile, splits):
        d.tile, d.splits = tile, splits
        d.zeros = ops._zero_page(device).data_ptr()
        if splits > 1:
            ws = ops._splitk_workspace(device)
            if splits * d.M * d.N * 4 > ws.numel() * 4 - _lib.AF_SPLITK_COUNTER_BYTES:
                return None
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
            d.splitk_fused = int(splits <= ops.SPLITK_FUSED_MAX)      # the rule ops._launch_gemm applies: narrow splits reduce in-kernel
        else:
            d.splitk_fused = 0
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            if L.af_gemm(C.byref(d), st) < 0:
                if args.trace:
                    print("  refused:", _lib.last_error() if hasattr(_lib, "last_error") else "", flush=True)
                return None
        if args.cold_weights:
            # the real step meets every weight cold in HBM (1.7 GB read once per step) while activations were just produced: evict
            # everything, re-read the activation operands, then time ONE launch; median of 5
            ts = []
            for _ in range(5):
                flush.fill_(1.0)
                for ptr, nb in ((d.a1, d.M * max(d.lda1, d.c1) * 2), (d.a2, d.M * max(d.lda2, d.c2) * 2 if d.a2 else 0)):
                    if ptr and nb and d.taps == 1:
                        L.af_prefetch(ptr, int(nb), st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.af_gemm(C.byref(d), st)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[2]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            L.af_gemm(C.byref(d), st)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / args.reps

    retuned = set()

    protected = set()
    for path in filter(None, args.protect.split(",")):
        import ast
        for line in open(path):
            if line.startswith("('"):
                protected.add(ast.literal_eval(line)[0])

    def recorder(key, d, device):
        if key in protected and key in table:
            return table[key]
        if args.try_tile and key in table and key not in retuned:
            retuned.add(key)
            tl = args.try_tile
            ok = not (tl == 15 and (d.act == _lib.AF_ACT_GEGLU or d.out_mode != 0 or d.N % 128 != 0 or d.M < 16384 or d.c1 % 64 or d.c2 % 64 or d.upsample not in (0, 1)))
            ok = ok and candidate_ok(d, tl, 1, _lib, ops)           # (outside its scope the library falls back to another tile silently)
            if args.new_halo_forms:
                ok = ok and tl == 14 and L.af_gemm_halo_variant(C.byref(d)) in (2, 3)
            if ok:
                cur = table[key]
                t_cur = timed(d, device, cur[0], cur[1])
                res = {f"{cur[0]}x{cur[1]}": None if t_cur is None else round(t_cur * 1e3, 1)}
                best, best_t = cur, t_cur
                for sp in sorted({1, 2, cur[1], 2 * cur[1]}):
                    if sp > 16 or not candidate_ok(d, tl, sp, _lib, ops):
                        continue
                    if sp > 1 and d.kpad // 64 < 8:
                        continue
                    t = timed(d, device, tl, sp)
                    if t is None:
                        continue
                    res[f"{tl}x{sp}"] = round(t * 1e3, 1)
                    if best_t is None or t < 0.98 * best_t:
                        best, best_t = (tl, sp), t
                if best != cur:
                    table[key] = best
                log.append((key, best, None if best_t is None else round(best_t * 1e3, 1), 0.0, res))
            return table[key]
        again = args.retune or (args.retune_halo and ops.conv_halo_eligible(d))
        if key in table and (not again or key in retuned):
            return table[key]
        retuned.add(key)
        nk = d.kpad // 64
        best, best_t, res = (0, 1), None, {}
        for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17):
            geglu = d.act == _lib.AF_ACT_GEGLU
            if args.tiles and str(tile) not in args.tiles.split(","):
                continue
            if tile == 14 and not ops.conv_halo_eligible(d):
                continue                        # halo-resident 3x3 kernel: its split-K slices are 64-channel chunks
            if tile >= 7 and (d.upsample not in (0, 1) or (d.upsample and tile not in (7, 8, 11, 12, 13, 14, 15)) or d.c1 % 64 or d.c2 % 64):
                continue                        # whole-line kernel: 64-multiples of channels; nearest-x2 upsample in the non-GEGLU tiles
            if tile in (11, 13) and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T or d.N % 160 != 0):
                continue                        # 128 x 160, four waves (11: two workgroups per CU; 13: four-slot ring)
            if tile == 15 and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T or d.N % 128 != 0 or d.M < 32768):
                continue                        # 256 x 128, eight waves: narrow outputs over many rows
            if tile == 12 and (geglu or d.out_mode == _lib.AF_OUT_SPLIT_T):
                continue                        # 128 x 128 with a four-slot ring
            if tile in (16, 17) and not candidate_ok(d, tile, 1, _lib, ops):
                continue                        # 64 x 128 / 128 x 64: three workgroups per CU
            if tile == 7 and (d.N % (256 if geglu else 320) != 0):
                continue
            if tile == 8 and geglu:
                continue
            if tile in (9, 10) and (not geglu or d.N % (320 if tile == 9 else 256) != 0):
                continue
            if tile in (5, 6) and (d.taps != 1 or d.out_mode != 0 or d.M * d.N < 256 * 256 * 128):
                continue                        # 256-row ring tiles: plain / GEGLU 1x1 GEMMs with at least ~128 tiles
            for splits in (1, 2, 3, 4, 6, 8, 12, 16):
                if tile in (5, 6) and splits > 1:
                    continue
                f32 = d.out_mode == _lib.AF_OUT_F32                      # weight gradients: fp32 from the reduce pass, or unsplit from tiles 1 / 2
                if f32 and splits == 1 and tile > 2:
                    continue
                if splits > 1 and (d.act == _lib.AF_ACT_GEGLU or d.out_mode == _lib.AF_OUT_SPLIT_T or nk < 4 * splits):
                    continue
                if tile == 14 and splits > (d.c1 + d.c2) // 64:
                    continue
                if args.trace:
                    print("timing", key, tile, splits, flush=True)
                t = timed(d, device, tile, splits)
                if t is None:
                    continue
                res[f"{tile}x{splits}"] = round(t * 1e3, 1)
                if best_t is None or t < best_t:
                    best, best_t = (tile, splits), t
        if best_t is None:                      # nothing in the requested tile set takes this launch: keep what the table has
            return table.get(key, (0, 1))
        table[key] = best
        gf = 2.0 * d.M * d.N * d.K / (best_t * 1e-3) / 1e12
        log.append((key, best, round(best_t * 1e3, 1), round(gf, 1), res))
        return best

    unet = UNetModel(**SD15_UNET_CONFIG)
    rng.load_synth_weights(unet, seed=0)
    unet = unet.to(dev).eval()
    ops._tune_recorder = recorder
    with torch.no_grad():
        for b, (lh, lw) in [(int(v), tuple(int(q) for q in hw.split("x"))) for hw in args.latents.split(",") for v in args.batches.split(",")]:
            x = rng.synth_input("bench.x", (b, 4, lh, lw), seed=1).to(dev)
            ctx = rng.synth_input("bench.ctx", (b, 77, 768), seed=1).to(dev)
            unet(x, torch.full((b,), 500, device=dev), ctx, extra_info=None)
            torch.cuda.synchronize()
    if args.train:
        from adaface_dev_amd.adaface.arc2face_models import CLIPTextModelWrapper
        from adaface_dev_amd.adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
        from adaface_dev_amd.adaface.unet_teachers import Arc2FaceTeacher
        from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
        from adaface_dev_amd.ldm.trainer import DistillTrainer
        ldm = LatentDiffusion(SD15_UNET_CONFIG)
        rng.load_synth_weights(ldm.model.diffusion_model, seed=0)
        id2ada = Arc2Face_ID2AdaPrompt()
        rng.load_synth_weights(id2ada.text_to_image_prompt_encoder, seed=2)
        rng.load_synth_weights(id2ada.subj_basis_generator.prompt2token_proj, seed=3)
        text_enc = CLIPTextModelWrapper()
        rng.load_synth_weights(text_enc, seed=4)
        ldm = ldm.to(dev)
        for p in ldm.model.diffusion_model.parameters():
            p.requires_grad_(False)
        ldm.unet_teacher = Arc2FaceTeacher(unet)                      # shapes only: the teacher can share the tuned U-Net
        ldm.model.set_up_ffn_loras()                                  # the FFN DoRA adapters' GEMM shapes as well
        tr = DistillTrainer(ldm, id2ada.to(dev), text_enc.to(dev), batch_size=4, accumulate_grad_batches=2)
        for i in range(4):
            b = dict(x_start=rng.synth_input("tb.x", (4, 4, 64, 64), seed=5).to(dev),
                     face_id_embs=rng.synth_input("tb.id", (4, 512), seed=5).to(dev), fg_mask=torch.ones(4, 1, 64, 64, device=dev))
            tr.training_step(b, i)
            torch.cuda.synchronize()
    for stage in [s for s, on in ((1, args.bench_train), (2, args.bench_train2)) if on]:
        # exactly the micro-batches bench.py times (its own model construction), eager, six of them = two 2,3,4-step cycles
        import bench
        ns = argparse.Namespace(batch=4, no_ffn_lora=False, no_train_graphs=True, train_steps=6, train_warmup=2, no_roofline=True, distill_only=False)
        bench.run_train(ns, (1, 0, 0, False), dev, stage=stage)
        torch.cuda.synchronize()
    if args.vae:
        from adaface_dev_amd.ldm.modules.diffusionmodules.model import AutoencoderKLDecoder
        ae = AutoencoderKLDecoder()
        with torch.no_grad():
            for n, p in ae.named_parameters():
                p.copy_(rng.synth_tensor(n, p.shape, seed=90))
        ae = ae.to(dev).eval()
        for b in (4, 1):
            ae.decode(rng.synth_input("vae.bench", (b, 4, 64, 64), seed=1).to(dev))
            torch.cuda.synchronize()
    ops._tune_recorder = None
    out = args.out or ops._TUNE_PATH
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({k: list(v) for k, v in sorted(table.items())}, f, indent=0)
    for row in log:
        print(row)
    print(f"wrote {len(table)} entries to {out}")


if __name__ == "__main__":
    main()
