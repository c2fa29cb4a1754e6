#!/usr/bin/env python
"""bench.py -- denoise-steps/s of the AdaFace SD-1.5 hot path on MI355X.

Workload (BASELINE.json configs[1], the one `metric` is quoted on): AdaFace txt2img DDIM
inference, 512x512 (latent 64x64), bs = 4 images/GPU with classifier-free guidance, i.e. one
*denoise step* = one U-Net epsilon-prediction on a batch of 8 (4 cond + 4 uncond, 77 context
tokens) + guidance combine + DDIM update.  Synthetic latents / context, seeded random weights
of the SD-1.5 architecture (no network for checkpoints).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

N > 1: one process per GPU.  Started WITHOUT a launcher (`python bench.py --gpus 8`), this process touches no GPU and
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (never an exec), relays its rank-0
JSON line and exits with its code.  Inference has no exchange step: independent replicas, weak scaling,
`value` = N*K steps / max-over-ranks time.  The same JSON line carries a `train` object: the Stage-1 distillation
micro-batch (BASELINE configs[2]/[3], reference main.py:618, 911-915) on the same ranks, data parallel with the bucketed
RCCL gradient all-reduce overlapped with the backward -- `train.value` (images/s) at N = 1 and N = 8 is what north_star's
">= 6x training images/s" is read from -- and a `train_stage2` object: the compositional-distillation micro-batch (configs[4]).
`--mode denoise` / `--mode train` / `--mode train2` run one leg only.  Defaults: 10 warm-up + 100 timed denoise steps (two whole
50-step samplings), 12 + 12 micro-batches per training leg; ~45 s in all on one MI355X (wall time per leg is printed on stderr).

The JSON line carries, besides the driver contract:
  roofline     -- dominant kernel family (the MFMA GEMM / implicit-conv template): algorithmic
                  FLOPs (conv3x3 400.33 + Linear 233.28 + conv1x1 43.62 GFLOP per
                  U-Net sample at 77 tokens) / summed launch time measured with hipEvents on the
                  launch stream in an instrumented pass of the same steps, against 2.5 PFLOP/s
                  dense fp16 MFMA (MI355X_MICROARCH.md);
  cpu_baseline -- the CPU oracle (torch-CPU fp32 restatement of the reference U-Net, "port")
                  timed on this host: ONE U-Net sample (bs 1, 64x64 latent, 77 tokens) = 1/8 of a
                  denoise step, all host threads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GEMM_GFLOP_PER_SAMPLE = 400.33 + 233.28 + 43.62   # SURVEY.md 8d / BASELINE.md section 2 (T = 77)
ATTN_GFLOP_PER_SAMPLE = 122.49 + 3.56
UNET_GFLOP_PER_SAMPLE = 803.27
MFMA_PEAK_TFLOPS = 2500.0                          # dense fp16/bf16, MI355X_MICROARCH.md
VAE_DEC_TFLOP = 2.51                               # SD-1.5 KL-f8 decoder, one 64 x 64 latent -> 512 x 512 image, forward (conv / matmul FLOPs of the reference Decoder)
FACE_TFLOP = 0.0034                                # ResNetFace-18 IR-SE on one 128 x 128 crop, forward
HBM_PEAK_GBS = 8000.0
# cross-attention core algorithmic bytes per U-Net sample (fp16): 2 B * (q + o: 2 N C, k + v: 2 T C) per layer,
# layers: 5 x (N 4096, C 320), 5 x (1024, 640), 5 x (256, 1280), 1 x (64, 1280); T = 77
XATTN_BYTES_PER_SAMPLE = 2 * sum(n_l * (2 * N * C + 2 * 77 * C) for n_l, N, C in ((5, 4096, 320), (5, 1024, 640), (5, 256, 1280), (1, 64, 1280)))


def dist_context():
    """(world, rank, local_rank, launched): launched = under torch.distributed.run (also with one rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    return world, rank, local_rank, launched


def init_device(ctx):
    """Bind this rank to its GPU and, under a launcher, join the RCCL process group (backend "nccl" IS RCCL on ROCm)."""
    world, rank, local_rank, launched = ctx
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or launched:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=600))
    return dev


def build_train(args, ctx, dev, stage=1):
    """The models, the trainer and the synthetic batches of a training leg (see run_train): -> (trainer, batches, step_kw, B, n_train, ldm, teacher, id2ada, text_enc).
    Shared with tools/autotune_instep.py, which times GEMM configurations inside these micro-batches."""
    world, rank, local_rank, launched = ctx
    import torch.distributed as dist
    from adaface_dev_amd import SD15_UNET_CONFIG, _lib, ops, rng
    from adaface_dev_amd.adaface.arc2face_models import CLIPTextModelWrapper
    from adaface_dev_amd.adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
    from adaface_dev_amd.adaface.unet_teachers import Arc2FaceTeacher
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from adaface_dev_amd.ldm.trainer import DistillTrainer

    _lib.lib()
    B = args.batch
    # random-init weights of the architecture (no checkpoints offline): the two U-Nets skip torch's default init and draw their
    # synthetic values on the GPU (rng.load_synth_weights(on_device=True): same rules, generated where they live)
    with rng.skip_default_init():
        ldm = LatentDiffusion(SD15_UNET_CONFIG)
        teacher = UNetModel(**SD15_UNET_CONFIG)
    ldm, teacher = ldm.to(dev), teacher.to(dev)
    rng.load_synth_weights(ldm.model.diffusion_model, seed=0, on_device=True)
    rng.load_synth_weights(teacher, seed=1, on_device=True)
    id2ada = Arc2Face_ID2AdaPrompt().to(dev)
    rng.load_synth_weights(id2ada.text_to_image_prompt_encoder, seed=2, on_device=True)
    rng.load_synth_weights(id2ada.subj_basis_generator.prompt2token_proj, seed=3, on_device=True)
    text_enc = CLIPTextModelWrapper().to(dev)
    rng.load_synth_weights(text_enc, seed=4, on_device=True)
    text_enc.extend_position_embeddings(97)          # --clip_prompt_max_length 97: the training context length (main.py:272)
    ldm = ldm.to(dev)
    for p in ldm.model.diffusion_model.parameters():
        p.requires_grad_(False)
    ldm.unet_teacher = Arc2FaceTeacher(teacher.to(dev))
    if not args.no_ffn_lora:
        ldm.model.set_up_ffn_loras()         # rank-192 DoRA on up_blocks.3 convs; 'unet_distill' is always on in Stage 1 (ddpm.py:3130-3134)
    step_kw = {}
    if stage == 2:
        from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
        B = 3                                                              # configs[4]: bs 3 / GPU
        ldm.comp_distill_priming_unet = UNetTeacher(ldm.unet_teacher.unet, cfg_scale_range=(2, 4), p_uses_cfg=1.0, name="comp_priming")
        ldm.uncond_context = (rng.synth_input("bench.uncond", (1, 97, 768), seed=5).to(dev), [""], {})
        ldm.model.set_up_attn_loras()                                      # rank-192 DoRA on q / k / v / out of layers 22-24
        # the face-gated terms: x0 predictions are decoded by the SD-1.5 KL-f8 decoder (synthetic weights) and handed to the ArcFace
        # wrapper; a fixed face box (28 x 28 latent pixels = 19 % of the image, inside the reference's 'good' range) stands in for the
        # RetinaFace detector network, which is an external package -- so the whole loss assembly (face alignment values,
        # subject-attention suppression, re-denoising of the subject-single instance, feature-matching loss) runs every micro-batch
        from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
        from adaface_dev_amd.ldm.modules.arcface_wrapper import ArcFaceWrapper, FaceCropper
        with rng.skip_default_init():
            vae = ldm.instantiate_first_stage()
            face_net = resnet_face18()
        vae.to(dev)
        rng.load_synth_weights(vae, seed=6, on_device=True)
        rng.load_synth_weights(face_net.to(dev), seed=7, on_device=True)
        ldm.arcface = ArcFaceWrapper(face_net.eval(), FaceCropper(lambda img, T=20: [(144.0, 128.0, 224.0, 224.0, 0.995)]))
    if stage == 1 and not args.distill_only:
        # the reference's Stage-1 iteration mix (v1-distill-arc2face-ada.yaml:28 unet_distill_iter_gap: 2): micro-batches alternate
        # do_normal_recon / do_unet_distill.  A recon iteration needs the null-prompt embedding (CFG), and -- arcface_align_loss_weight
        # 0.01, the reference default, without which its per-step losses do not exist (ddpm.py:2702) -- the VAE decoder and the
        # ArcFace wrapper; a fixed face box stands in for the RetinaFace detector network (an external package)
        from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
        from adaface_dev_amd.ldm.modules.arcface_wrapper import ArcFaceWrapper, FaceCropper
        ldm.uncond_context = (rng.synth_input("bench.uncond", (1, 97, 768), seed=5).to(dev), [""], {})
        with rng.skip_default_init():
            vae = ldm.instantiate_first_stage()
            face_net = resnet_face18()
        vae.to(dev)
        rng.load_synth_weights(vae, seed=6, on_device=True)
        rng.load_synth_weights(face_net.to(dev), seed=7, on_device=True)
        ldm.arcface = ArcFaceWrapper(face_net.eval(), FaceCropper(lambda img, T=20: [(144.0, 128.0, 224.0, 224.0, 0.995)]))
    if getattr(args, "reference_pass_structure", False):
        # every pass the reference runs, run: the null-prompt pass of a recon step twice, the class-prompt pass on priming steps too, the SS / SR
        # instances as separate calls, and (below) all 16 x0 predictions of a compositional step decoded as the reference's image logger has them.
        # The default legs leave that work out (results identical: DESIGN 8.5); this switch is the like-for-like run against BASELINE configs[2..4].
        ldm.cache_uncond_in_step = False
        ldm.skip_unread_cls_priming = False
        ldm.batch_no_grad_instances = False
        ldm.share_no_grad_trunk = False          # (round 5's pass batching too: every gradient-free instance pass walks its own trunk,
        ldm.batch_cond_with_uncond = False       #  prompt rows and null-prompt rows are separate U-Net calls, as the reference makes them)
    tr = DistillTrainer(ldm, id2ada.to(dev), text_enc.to(dev), batch_size=B, accumulate_grad_batches=2, prompt_len=97, stage=stage,
                        use_graphs=not args.no_train_graphs)
    if getattr(args, "reference_pass_structure", False):
        tr.decode_all_blocks_for_logging = True
    if stage == 1 and not args.distill_only:
        tr.unet_distill_iter_gap = 2
    n_train = sum(a.numel for a in tr.arenas)

    def batch(i):
        seed = 42 + i + rank * 10 ** 8
        return dict(x_start=rng.synth_input(f"tb.x{i % 4}", (B, 4, 64, 64), seed=seed).to(dev),
                    face_id_embs=rng.synth_input(f"tb.id{i % 4}", (B, 512), seed=seed).to(dev),
                    fg_mask=torch.ones(B, 1, 64, 64, device=dev))
    batches = [batch(i) for i in range(4)]
    return tr, batches, step_kw, B, n_train, ldm, teacher, id2ada, text_enc


def run_train(args, ctx, dev, stage=1):
    """stage 1: BASELINE configs[2] (1 GPU) / configs[3] (DDP); stage 2: configs[4] (see the end of this docstring).  One *step* = one training
    micro-batch of bs images/GPU: face IDs -> Arc2Face encoder -> trainable SubjBasisGenerator -> frozen text encoder ->
    teacher multi-step targets + student eps per step (HALF_BS = ceil(bs/steps) instances, steps cycling 2,3,4 as
    ddpm.py:1270-1289) -> masked MSE -> backward to the 85 M SubjBasisGenerator weights; every 2nd micro-batch the
    bucketed gradient all-reduce (overlapped with the backward), unscale and fused CAdamW.  Full-size models: 2 x SD-1.5
    U-Net (student, teacher) + 3 x CLIP-L text transformers, seeded random weights.  Returns the result dict on rank 0.

    stage 2 (BASELINE configs[4], reference ddpm.py:2371-2480): one *step* = one compositional-distillation micro-batch: BLOCK_SIZE 1 of
    the bs-3 batch (the reference fixes it, :2372-2374), latents primed from pure noise by the second (teacher) U-Net with
    classifier-free guidance over 3-4 steps, then 4 subject-compos denoising steps of the student on the four-prompt batch with
    activation capture (explicit attention in layers 22-24, score mixing / normalisation, trainable attention + FFN DoRA adapters),
    guidance passes, the subject-single x0 predictions of every step decoded for the face pipeline (what the loss consumes; the reference also decodes the
    other three blocks, for its image logger, ddpm.py:2454-2465; a fixed face box stands in for the RetinaFace detector network), the
    whole loss assembly incl. the re-denoising of the subject-single instance and the feature-matching loss, backward, CAdamW."""
    world, rank, local_rank, launched = ctx
    import torch.distributed as dist
    from adaface_dev_amd import _lib, ops
    tr, batches, step_kw, B, n_train, ldm, teacher, id2ada, text_enc = build_train(args, ctx, dev, stage)
    # what the leg runs besides U-Net passes, counted as it runs (the FLOP count of the leg's roofline fraction below): images through the VAE decoder
    # (forward; and those decoded WITH an input gradient, i.e. whose backward runs too) and face crops through ResNetFace-18
    extra = {"vae_fwd_images": 0, "vae_grad_images": 0, "face_fwd_images": 0, "face_grad_images": 0}
    if getattr(ldm, "first_stage_model", None) is not None:
        _dec = ldm.first_stage_model.decode

        def counting_decode(z, *a, **k):
            extra["vae_fwd_images"] += int(z.shape[0])
            if torch.is_grad_enabled() and z.requires_grad:
                extra["vae_grad_images"] += int(z.shape[0])
            return _dec(z, *a, **k)
        ldm.first_stage_model.decode = counting_decode
    if getattr(ldm, "arcface", None) is not None:
        _face = ldm.arcface.arcface.forward

        def counting_face(x, *a, **k):
            extra["face_fwd_images"] += int(x.shape[0])
            if torch.is_grad_enabled() and x.requires_grad:
                extra["face_grad_images"] += int(x.shape[0])
            return _face(x, *a, **k)
        ldm.arcface.arcface.forward = counting_face
    steps = args.train_steps + (args.train_steps % 2)            # whole accumulation windows
    warm = max(2, args.train_warmup + (args.train_warmup % 2))
    losses = []
    for i in range(warm):
        tr.training_step(batches[i % 4], i, **step_kw)
    if world > 1:
        dist.barrier()
    # one all-legs run in four showed a ~0.4 s stall somewhere in this loop that the per-micro-batch timings below did not; a generational
    # collection of the host's object graph (models, graphs, autograd nodes of three legs) is the likely cause: collect now, collector off while timing
    import gc
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    sync_debug = getattr(args, "sync_debug", None)
    if sync_debug:
        import traceback
        import warnings
        sync_log = open(sync_debug, "w")

        def show(message, category, filename, lineno, file=None, line=None):
            stack = [f for f in traceback.extract_stack()[:-1] if "adaface" in f.filename or "bench.py" in f.filename]
            sync_log.write(f"SYNC {message}\n" + "".join(f"    {f.filename.split('/')[-1]}:{f.lineno} {f.name}: {f.line}\n" for f in stack[-6:]))
        warnings.showwarning = show
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
    # The caching allocator hands a freed block out again only when the stream work that used it has finished; a host that runs ahead of the GPU
    # therefore sometimes finds no free block and calls hipMalloc for a new segment inside a micro-batch (timing-dependent: 0.2 - 0.4 s in one
    # micro-batch of a run now and then, profiles/r05ai).  Slack in the pool removes that: one large block allocated and freed here stays cached and
    # is split on demand.  segment counts per micro-batch are reported so that a stall can be told from an allocation.
    reserve_gb = float(os.environ.get("AF_BENCH_ALLOC_RESERVE_GB", "16"))
    if reserve_gb > 0:
        free_b, _ = torch.cuda.mem_get_info()
        nbytes = int(min(reserve_gb * (1 << 30), 0.5 * free_b))
        if nbytes > (1 << 28):
            _slack = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            del _slack
    seg0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    for k in extra:
        extra[k] = 0
    t0 = time.perf_counter()
    host_ms, seg_allocs = [], []
    for i in range(steps):
        th = time.perf_counter()
        losses.append(tr.training_step(batches[i % 4], warm + i, **step_kw))
        host_ms.append(round((time.perf_counter() - th) * 1e3, 1))
        seg1 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
        seg_allocs.append(int(seg1 - seg0))
        seg0 = seg1
    if sync_debug:
        torch.cuda.set_sync_debug_mode("default")
        sync_log.close()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    extra_timed = dict(extra)                         # counts of the timed region only
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    per_type = None
    if stage == 1 and not args.distill_only:
        # the mix's two iteration types timed one micro-batch at a time (synchronised), after the timed region: 8 more micro-batches
        acc = {}
        for i in range(8):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tr.training_step(batches[i % 4], warm + steps + i)
            torch.cuda.synchronize()
            acc.setdefault(tr.last_iter_type, []).append((time.perf_counter() - t1) * 1e3)
        per_type = {k: {"micro_batches": len(v), "ms_per_micro_batch": round(sum(v) / len(v), 2)} for k, v in acc.items()}
    fam = None
    if not args.no_roofline:
        # every rank runs the instrumented micro-batches (the gradient exchange is a collective); rank 0 reports its families
        tr.set_graphs_enabled(False)                   # per-launch events need eager launches
        ops.prof_reset()
        ops.prof_enable(True)
        for i in range(6):                             # one full 2,3,4-step cycle twice over (6 micro-batches)
            tr.training_step(batches[i % 4], warm + steps + 8 + i, **step_kw)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        fam = {}
        for name, f in (("gemm", _lib.AF_FAM_GEMM), ("attn", _lib.AF_FAM_ATTN), ("gnorm", _lib.AF_FAM_GNORM),
                        ("lnorm", _lib.AF_FAM_LNORM), ("elem", _lib.AF_FAM_ELEM)):
            n, ms = ops.prof_read(f)
            fam[name] = {"launches_per_step": round(n / 6, 1), "ms_per_step": round(ms / 6, 3)}
        ops.prof_reset()
    out = None
    if rank == 0:
        ms = elapsed / steps * 1e3
        # algorithmic FLOPs of one micro-batch (SURVEY.md 8d, T = 97: 804.96 GFLOP per U-Net sample forward; student = fwd +
        # activation-gradient bwd ~ 2x fwd, teacher = 1x fwd): the 2,3,4-step cycle runs HALF_BS x steps = 4, 6, 4 student and
        # as many teacher sample-passes => 14/3 of each per micro-batch on average at bs 4
        per_mb = sum(-(-B // s) * s for s in (2, 3, 4)) / 3.0
        train_tflop = per_mb * 3 * 0.80496
        what = (f"whole micro-batch: {per_mb:.2f} student fwd+bwd + {per_mb:.2f} teacher fwd U-Net sample-passes "
                f"= {train_tflop:.2f} TFLOP algorithmic (conv/matmul only, encoders not counted) / wall time")
        metric = "train-images/sec Stage-1 Arc2Face distillation bs=4/GPU"
        workload = (f"stage1_unet_distill micro-batch: bs={B}/GPU, 512x512 (latent 64x64), 97 context tokens, denoising steps cycle 2,3,4 "
                    "with HALF_BS=ceil(bs/steps), teacher+student SD-1.5 U-Nets, 3 CLIP-L encoders, "
                    f"{n_train} trainable fp32 params, accumulate_grad_batches=2, CAdamW")
        if stage == 1 and not args.distill_only:
            # a recon micro-batch on images: 2 denoising steps x (student fwd + bwd on bs 4, the CFG null pass, the class-prompt pass) = 2 x 16
            # sample-forward equivalents; on pure noise (p = 0.4) four no-grad priming steps (8 each: the pass + its null pass) come first.
            # (The reference runs the null-prompt pass twice per step with identical arguments -- after the subject pass and after the
            # class-prompt pass; here the second request takes the first one's tensor, ddpm.guided_denoise(uncond_cache=...) -- and a
            # class-prompt pass in every priming step whose result nothing reads; the count below is of the passes that run.)
            recon_fwd = 0.6 * 32 + 0.4 * (4 * 8 + 32)
            if getattr(args, "reference_pass_structure", False):
                # ... and with every pass the reference runs: a second null pass per step (20 per step), class-prompt pass + its null pass on priming steps (16 each)
                recon_fwd = 0.6 * 40 + 0.4 * (4 * 16 + 40)
            train_tflop = 0.5 * train_tflop + 0.5 * recon_fwd * 0.80496
            what = (f"mean over the reference's Stage-1 iteration mix (micro-batches alternate normal recon / U-Net distillation): "
                    f"{train_tflop:.1f} TFLOP algorithmic per micro-batch on average (U-Net passes only; the recon iterations' VAE decodes and "
                    "ResNetFace-18 passes are not counted) / wall time")
            workload = ("the reference's Stage-1 iteration mix, unet_distill_iter_gap = 2 (v1-distill-arc2face-ada.yaml:28): micro-batches alternate "
                        f"do_normal_recon (bs {B}: 2 denoising steps with CFG + class-prompt passes + capture of layers 22-24 -- the null-prompt pass of a step computed once for both guided passes --, on the images or, p = 0.4, "
                        "from pure noise after 4 priming steps; x0 decoded for the face pipeline) and do_unet_distill (" + workload + ")")
        if stage == 2:
            # priming: 3.5 steps x (positive + negative pass at batch 2) = 14 sample forwards; student: 4 steps x (SS + SR + SC + MC
            # + 4 unconditional = 8 sample forwards, the SC pass -- SC and MC when the scores are mixed -- also backward: ~1.5)
            fwd_equiv = 3.5 * 4 + 4 * (8 + 1.5)
            train_tflop = fwd_equiv * 0.80496
            what = (f"whole micro-batch: ~{fwd_equiv:.0f} U-Net sample-forward equivalents (14 priming, 32 student forward incl. guidance, "
                    f"~6 backward) = {train_tflop:.1f} TFLOP algorithmic / wall time")
            metric = "train-images/sec Stage-2 compositional distillation bs=3/GPU"
            workload = ("stage2_comp_distill micro-batch: bs=3/GPU of which BLOCK_SIZE=1 is denoised (reference ddpm.py:2372-2374), 512x512, 97 tokens, "
                        "priming U-Net 3-4 CFG steps + student 4 subject-compos steps x 4 prompts with capture of layers 22-24 + guidance passes + re-denoising of the subject-single instance, "
                        "x0 of the subject-single block decoded per step (the blocks the loss reads; image logging is out of scope), VAE decoder + ResNetFace-18 with input gradients, "
                        f"attention + FFN DoRA adapters, {n_train} trainable fp32 params, accumulate_grad_batches=2, CAdamW")
        # + the VAE decoder and ResNetFace-18 passes the micro-batches ran (counted above): VAE_DEC_TFLOP per decoded 512 x 512 image forward (the
        # reference Decoder under FLOP hooks), as much again where the input gradient is taken (dgrad convolutions only: the VAE is
        # frozen); ResNetFace-18 IR-SE at 128 x 128: FACE_TFLOP per crop and direction
        unet_tflop = train_tflop
        vae_tflop = VAE_DEC_TFLOP * (extra_timed["vae_fwd_images"] + extra_timed["vae_grad_images"]) / steps
        face_tflop = FACE_TFLOP * (extra_timed["face_fwd_images"] + extra_timed["face_grad_images"]) / steps
        train_tflop = unet_tflop + vae_tflop + face_tflop
        what += (f"; + {vae_tflop:.2f} TFLOP of VAE decoding ({extra_timed['vae_fwd_images'] / steps:.2f} images forward, {extra_timed['vae_grad_images'] / steps:.2f} with the "
                 f"input-gradient backward, per micro-batch; {VAE_DEC_TFLOP} TFLOP per image and direction) + {face_tflop:.3f} TFLOP of ResNetFace-18 "
                 f"({extra_timed['face_fwd_images'] / steps:.2f} crops forward per micro-batch) = {train_tflop:.2f} TFLOP per micro-batch: the fraction below is of the whole leg")
        out = {"metric": metric, "value": round(world * B * steps / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": workload,
                          "pass_structure": ("reference: every pass the reference runs (--reference-pass-structure)" if getattr(args, "reference_pass_structure", False) else
                                             "default: passes whose result nothing reads, or that repeat an identical pass of the same step, do not run "
                                             "(DESIGN 8.5); --reference-pass-structure is the like-for-like run"),
                          "parallelism": f"dp{world} (RCCL bucketed all-reduce overlapped with backward)" if world > 1 else "single GPU",
                          "hipgraph_segments": [f"{g.name}: {sum(1 for e in g.entries.values() if e.get('state') == 'graph')} captured" for g in tr.graph_segments],
                          "optimizer_steps": tr.global_step, "skipped_steps": tr.skipped_steps, "loss_scale": tr.scaler.scale,
                          "last_loss": float(losses[-1]), "finite": bool(all(torch.isfinite(l) for l in losses)),
                          "per_iteration_type": per_type,
                          "host_ms_per_micro_batch_in_timed_region": host_ms,
                          "allocator_segments_per_micro_batch_in_timed_region": seg_allocs},
               "roofline": {"bound": "mfma", "achieved": round(train_tflop / (ms * 1e-3), 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(train_tflop / (ms * 1e-3) / MFMA_PEAK_TFLOPS, 4), **train_traffic(stage, args),
                            "tflop_per_micro_batch": {"unet": round(unet_tflop, 3), "vae_decoder": round(vae_tflop, 3), "resnetface18": round(face_tflop, 4)},
                            "what": what},
               "families_per_micro_batch": fam}
    del tr, ldm, teacher, id2ada, text_enc
    torch.cuda.empty_cache()
    return out


def train_traffic(stage, args):
    """HBM-side bytes per Stage-1 distillation micro-batch, per kernel family, from the committed rocprofv3 --pmc passes of the distill-only
    train leg (tools/profile_round.sh -> profiles/r*_train_traffic.json; FETCH_SIZE x 2 + WRITE_SIZE as for the denoise leg).  Reported only
    for the tree it was measured on (sources_sha) and for the leg it was measured on; null otherwise."""
    import glob
    from adaface_dev_amd import _lib
    if stage == 1:
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_traffic.json")), reverse=True):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("sources_sha") == _lib.sources_sha():
                fams = {k: round(v["bytes_per_step"]) for k, v in tj.items() if isinstance(v, dict)}
                return {"traffic": float(sum(fams.values())), "traffic_unit": "bytes per distillation micro-batch (families below; the mix's recon "
                        "micro-batches are not in this measurement)", "traffic_families": fams, "traffic_source": os.path.basename(tpath)}
    return {"traffic": None}


XATTN_BLOCK_GFLOP_PER_SAMPLE = 32.1        # to_q + to_k + to_v + core + to_out of the 16 attn2 blocks


def cross_attn_block_time(unet, ctx2, B, dev):
    """north_star's "MFMA utilisation on U-Net cross-attention", on the WHOLE attn2 block as SURVEY.md 8d defines it: for each of the
    16 cross-attention layers the q projection, the 77-key attention core and the output projection with its residual (what
    ``CrossAttention.hip`` launches per layer) are timed in isolation -- 10 calls captured in a hipGraph, inputs resident -- plus ONE
    launch of the batched k / v projection of all layers.  The algorithmic work (32.1 GFLOP per U-Net sample) over the summed time
    is priced against the dense fp16 MFMA peak; the core alone is the HBM-bound ``cross_attn_core`` entry above."""
    from adaface_dev_amd.ldm.modules.attention import FOLD_LAYERNORM, SpatialTransformer

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    with torch.no_grad():
        t_kv = timed(lambda: [setattr(m, "_kv_pre", None) for m in unet._project_context_all(ctx2)])
        layers = unet._project_context_all(ctx2)          # leave the projections in place for the per-layer calls
        per_layer, tot = [], t_kv
        n = 2 * B
        for blocks in (unet.input_blocks, [unet.middle_block], unet.output_blocks):
            for module in blocks:
                for layer in module:
                    if isinstance(layer, SpatialTransformer):
                        a2 = layer.transformer_blocks[0].attn2
                        C = a2.inner_dim
                        N = {320: 4096, 640: 1024}.get(C, 256)
                        if C == 1280 and module is unet.middle_block:
                            N = 64
                        x = torch.randn(n * N, C, device=dev).half()
                        ln = layer.transformer_blocks[0].norm2 if FOLD_LAYERNORM else None       # what the step launches: norm2 folded into to_q
                        ms = timed(lambda a2=a2, x=x, N=N, ln=ln: a2.hip(x, n, N, ctx2, None, residual=x, ln=ln))
                        per_layer.append((C, N, round(ms * 1e3, 1)))
                        tot += ms
        for m in layers:
            m._kv_pre = None
    tf = XATTN_BLOCK_GFLOP_PER_SAMPLE * 1e9 * n / (tot * 1e-3) / 1e12
    return {"ms_per_step": round(tot, 4), "kv_projection_ms": round(t_kv, 4), "tflops": round(tf, 1), "mfma_frac": round(tf / MFMA_PEAK_TFLOPS, 4),
            "definition": "sum over the 16 attn2 blocks of [to_q (norm2 folded in) + 77-key attention core + to_out with residual: ONE launch (af_xattn_fused) at "
                          "C = 320 / U-Net batch >= 2 (tiled kernel, round 4), three launches otherwise] + one batched k/v projection, each block timed in isolation (hipGraph of 10 calls); "
                          "32.1 GFLOP per U-Net sample (SURVEY.md 8d)",
            "layers_C_N_us": per_layer}


def run_denoise(args, ctx, dev):
    """BASELINE configs[1]: the headline line.  Returns the result dict on rank 0 (None elsewhere)."""
    world, rank, local_rank, launched = ctx
    from adaface_dev_amd import SD15_UNET_CONFIG, _lib, ops, rng
    from adaface_dev_amd.ldm.models.diffusion.ddim import DDIMSampler
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion

    _lib.lib()  # no fallback: fail here if the HIP extension is absent
    B = args.batch
    with rng.skip_default_init():
        ldm = LatentDiffusion(SD15_UNET_CONFIG)
    ldm = ldm.to(dev).eval()
    unet = ldm.model.diffusion_model
    rng.load_synth_weights(unet, seed=0, on_device=True)     # random-init weights, drawn on the GPU (same rules as the fixtures' Philox values)
    unet.prepare()
    for p in unet.parameters():
        p.requires_grad_(False)

    seed = 42 + rank * 10 ** 8   # per-rank stream, as ldm/util.py:524-530
    x = rng.synth_input("bench.x", (B, 4, 64, 64), seed=seed).to(dev)
    c_c = rng.synth_input("bench.ctx", (B, 77, 768), seed=seed).to(dev)
    c_u = rng.synth_input("bench.uctx", (B, 77, 768), seed=seed).to(dev)
    sampler = DDIMSampler(ldm)
    sampler.make_schedule(50, verbose=False)
    ts_desc = sampler.ddim_timesteps[::-1].copy()     # 981 ... 1
    scales = sampler.guide_scales(50, 4.0)

    # static buffers of one denoise step
    x_in = torch.empty((2 * B, 4, 64, 64), dtype=torch.float32, device=dev)
    t_in = torch.empty((2 * B,), dtype=torch.int64, device=dev)
    ctx2 = torch.cat([c_c, c_u]).to(torch.float16).contiguous()   # (cond, uncond) order, ddim.py:244
    state = {"x": x.clone(), "eps": None, "graph": None}

    def unet_eps():
        return unet(x_in, t_in, ctx2, extra_info=None)

    def step(i):
        idx = i % 50
        index = 50 - idx - 1
        if idx == 0:
            state["x"].copy_(x)      # a new 50-step sampling starts from the initial noise (runs longer than 50 steps cycle through whole samplings)
        x_in[:B].copy_(state["x"])
        x_in[B:].copy_(state["x"])
        t_in.fill_(int(ts_desc[idx]))
        if state["graph"] is not None:
            state["graph"].replay()
            eps = state["eps"]
        else:
            eps = unet_eps()
        xp, _ = ops.cfg_ddim_step(eps, state["x"], scales[idx], float(sampler.ddim_alphas[index]),
                                  float(sampler.ddim_alphas_prev[index]), True)
        state["x"] = xp

    def barrier():
        if world > 1 or launched:
            import torch.distributed as dist
            dist.barrier()

    with torch.no_grad():
        step(0)                      # eager warm-up: packs weights, sets kernel attributes
        torch.cuda.synchronize()
        if not args.no_graph:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            pf = None
            if args.prefetch_stream > 0:
                # weight prefetch beside the GEMMs on a second stream of the SAME graph (ops.WeightPrefetcher): record the launch order
                # in this eager pass, replay it depth launches ahead during the capture
                pf = ops.WeightPrefetcher(depth=args.prefetch_stream, max_workgroups=args.prefetch_wgs)
                ops.set_weight_prefetcher(pf.record())
            with torch.cuda.stream(s):
                unet_eps()
            torch.cuda.current_stream().wait_stream(s)
            side = torch.cuda.Stream() if pf is not None else None
            # thread_local: with N > 1 the RCCL watchdog thread may touch the HIP runtime while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                if pf is not None:
                    pf.play(side)
                state["eps"] = unet_eps()
                if pf is not None:
                    pf.join()
            if pf is not None:
                print(f"[bench] weight prefetch stream: {pf.issued} prefetch launches for {len(pf.plan)} weight-consuming launches, "
                      f"depth {pf.depth}, <= {pf.max_wg} workgroups each", file=sys.stderr)
                ops.set_weight_prefetcher(None)
            state["graph"] = g
        state["x"] = x.clone()
        for i in range(args.warmup):
            step(i)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        finite = bool(torch.isfinite(state["x"]).all())

    if world > 1 or launched:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    roofline = None
    if not args.no_roofline and rank == 0:
        # instrumented pass: same steps, eager launches, hipEvents around every launch of each family.  The C = 320 cross-attention blocks
        # run here in their three-launch form (to_q GEMM, core, to_out GEMM) so that the families keep their definitions -- every projection
        # in the GEMM family, the 16 cores in `xattn`; the timed region above ran them as one launch each (af_xattn_fused, tiled form: profiles/r04g)
        from adaface_dev_amd.ldm.modules import attention as _attn_mod
        state["graph"] = None
        fused_xattn, _attn_mod.FUSE_XATTN = _attn_mod.FUSE_XATTN, False
        with torch.no_grad():
            ops.prof_reset()
            ops.prof_enable(True)
            nprof = min(args.steps, 5)
            for i in range(nprof):
                step(args.warmup + i)
            torch.cuda.synchronize()
            ops.prof_enable(False)
        _attn_mod.FUSE_XATTN = fused_xattn
        fam = {}
        for name, f in (("gemm", _lib.AF_FAM_GEMM), ("attn", _lib.AF_FAM_ATTN), ("gnorm", _lib.AF_FAM_GNORM),
                        ("lnorm", _lib.AF_FAM_LNORM), ("elem", _lib.AF_FAM_ELEM), ("xattn", _lib.AF_FAM_XATTN)):
            n, ms = ops.prof_read(f)
            fam[name] = {"launches_per_step": n / nprof, "ms_per_step": ms / nprof}
        ops.prof_reset()
        g_ms = fam["gemm"]["ms_per_step"]
        g_n = fam["gemm"]["launches_per_step"]
        flops_step = GEMM_GFLOP_PER_SAMPLE * 1e9 * 2 * B
        achieved = flops_step / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        a_ms = fam["attn"]["ms_per_step"]
        x_ms = fam["xattn"]["ms_per_step"]
        # HBM bytes per launch of the GEMM family: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (separate passes, gfx950 x2 fetch correction; tools/pmc_traffic.py), committed under profiles/ -- counters cannot be
        # read from inside the process, so the latest committed measurement is reported (null if absent)
        traffic = None
        traffic_src = None
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
            with open(tpath) as f:
                tj = json.load(f)
            # a measurement is only reported for the kernels it was taken on: the file records the hash of the GEMM sources
            # + tuning table (tools/pmc_traffic.py writes it); anything else is stale => null
            if tj.get("sources_sha") == _lib.sources_sha() and tj.get("batch", 4) == B and g_n:
                traffic = round(tj["gemm"]["bytes_per_step"] / g_n, 1)   # per af_gemm call, like `achieved`
                traffic_src = os.path.basename(tpath)
                break
        # the same fraction from the kernels that produced ms_per_step: the rocprofv3 --kernel-trace summary of the GRAPH-REPLAYED leg (this pass is eager
        # and instrumented, and runs the C = 320 cross-attention blocks as three launches: it reads ~5 % slower).  Counters / traces cannot be taken
        # in-process: the committed measurement (tools/profile_round.sh -> tools/rocprof_frac.py) is reported while its sources hash matches this build
        rocprof = None
        for rpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_frac.json")), reverse=True):
            with open(rpath) as f:
                rj = json.load(f)
            if rj.get("sources_sha") == _lib.sources_sha():
                rocprof = {k: rj[k] for k in ("file", "gemm_family_ms_per_step", "kernel_ms_per_step", "achieved_tflops", "frac", "what")}
                rocprof["json"] = os.path.basename(rpath)
                break
        roofline = {
            "kernel": "af_gemm family: af_conv3hd_kernel / af_gemm3w_kernel / af_gemm3_kernel / af_gemm_kernel (conv3x3 implicit GEMM + linear + conv1x1)",
            "rocprof": rocprof,
            "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "flops_per_launch": flops_step / g_n if g_n else None,
            "avg_launch_ms": g_ms / g_n if g_n else None,
            "families_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in fam.items()},
            "families_launches_per_step": {k: v["launches_per_step"] for k, v in fam.items()},
            "self_attn_tflops": round(122.49e9 * 2 * B / (a_ms * 1e-3) / 1e12, 2) if a_ms > 0 else None,
            # U-Net cross-attention CORE (QK^T + softmax + PV over 77 keys, 16 layers): HBM-bound by construction (76 FLOP per
            # algorithmic byte at C = 320), so it is priced against BOTH roofs; kernel boundary = af_attention only
            # (the q / kv / out projections run in the GEMM family)
            "cross_attn_core": None if x_ms <= 0 else {
                "ms_per_step": round(x_ms, 4), "launches_per_step": fam["xattn"]["launches_per_step"],
                "tflops": round(3.56e9 * 2 * B / (x_ms * 1e-3) / 1e12, 2),
                "mfma_frac": round(3.56e9 * 2 * B / (x_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                "algorithmic_GBps": round(XATTN_BYTES_PER_SAMPLE * 2 * B / (x_ms * 1e-3) / 1e9, 1),
                "hbm_frac": round(XATTN_BYTES_PER_SAMPLE * 2 * B / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        roofline["cross_attn_block"] = cross_attn_block_time(unet, ctx2, B, dev)

    cpu_baseline = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import unet_oracle as O
        # 32 threads measured fastest for this op mix on the GPU box's 256-thread host (16: 3.1 s, 32: 2.8 s,
        # 64: 3.9 s, 256: 137 s per U-Net sample); the count actually used is what is reported.
        ncores = min(32, len(os.sched_getaffinity(0)))
        torch.set_num_threads(ncores)
        sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
        xc = rng.synth_input("full.x", (1, 4, 64, 64), seed=0)
        cc = rng.synth_input("full.ctx", (1, 77, 768), seed=0)
        nrep = 4
        with torch.no_grad():
            O.unet_forward(sd, SD15_UNET_CONFIG, xc, torch.tensor([500]), cc, {})   # warm-up (thread pool, allocator)
            t1 = time.perf_counter()
            for _ in range(nrep):
                O.unet_forward(sd, SD15_UNET_CONFIG, xc, torch.tensor([500]), cc, {})
            dt = (time.perf_counter() - t1) / nrep
        cpu_baseline = {
            "value": round(1.0 / (dt * 2 * B), 5), "unit": "denoise-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nrep} timed U-Net sample forwards (bs 1, 64x64 latent, 77 tokens, fp32; 1 sample = 1/{2 * B} of a denoise "
                      f"step), mean {dt:.2f} s each; value = 1 / ({2 * B} x that)",
        }

    out = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        out = {
            "metric": "denoise-steps/sec SD-1.5 U-Net 512px bs=4/GPU",
            "value": round(value, 3), "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "adaface_txt2img_ddim: SD-1.5 U-Net eps-prediction + CFG + DDIM update, 512x512 (latent 64x64), "
                                   f"bs={B}/GPU, U-Net batch {2 * B} (cond+uncond), 77 tokens, DDIM-50 timesteps, random-init weights",
                       "parallelism": f"replicas x{world}", "hipgraph": not args.no_graph,
                       "unet_samples_per_s": round(value * 2 * B, 2),
                       "step_mfma_frac": round(UNET_GFLOP_PER_SAMPLE * 1e9 * 2 * B / (ms * 1e-3) / (MFMA_PEAK_TFLOPS * 1e12), 4),
                       "finite": finite},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
    del unet, ldm, sampler
    state.clear()
    torch.cuda.empty_cache()
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as a CHILD process (this parent never touches the GPU:
    torch.cuda.device_count() does not initialise it), relay the rank-0 JSON line, exit with the child's code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return r.returncode if line is not None or r.returncode else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed denoise steps (default: two whole 50-step DDIM samplings, 1.3 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (U-Net batch is 2x with CFG)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--prefetch-stream", type=int, default=int(os.environ.get("AF_PREFETCH_STREAM", "0")), metavar="DEPTH",
                    help="denoise leg: read the weights of launch i + DEPTH on a second stream of the captured graph while launch i runs (0 = off)")
    ap.add_argument("--prefetch-wgs", type=int, default=int(os.environ.get("AF_PREFETCH_WGS", "64")), help="workgroups per prefetch launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-ffn-lora", action="store_true", help="train leg: without the U-Net's trainable FFN DoRA adapters")
    ap.add_argument("--train-steps", type=int, default=12, help="timed micro-batches of the train leg")
    ap.add_argument("--train-warmup", type=int, default=24, help="untimed micro-batches (the hipGraph segments of every signature are captured in here: the from-noise recon variant, p = 0.4, is captured at its SECOND draw -- 12 were not always enough, profiles/r05ai)")
    ap.add_argument("--distill-only", action="store_true", help="train leg: every micro-batch a U-Net distillation iteration (rounds 1-2's leg) instead of the reference's recon / distill mix")
    ap.add_argument("--reference-pass-structure", action="store_true",
                    help="train legs: run every pass the reference runs (duplicate null-prompt pass, class-prompt pass on priming steps, SS / SR as separate "
                         "calls, all 16 x0 predictions decoded): the like-for-like run against BASELINE configs[2..4]; the default legs skip that work")
    ap.add_argument("--no-reference-leg", action="store_true", help="train legs: do not also time the --reference-pass-structure form (the default line carries both)")
    ap.add_argument("--no-train-graphs", action="store_true", help="train leg: launch every kernel from Python instead of replaying captured segments")
    ap.add_argument("--sync-debug", default=None, metavar="FILE", help="train legs: write the Python stack of every host<->device synchronisation "
                    "of the timed micro-batches to FILE (torch.cuda.set_sync_debug_mode); the timing of such a run is not a result")
    ap.add_argument("--mode", choices=["all", "denoise", "train", "train2"], default="all",
                    help="all (default): the headline denoise line (BASELINE configs[1]) carrying the Stage-1 training leg "
                         "(configs[2]/[3]) as its `train` object; denoise / train: one leg only (train prints its own line)")
    args = ap.parse_args()
    ctx = dist_context()
    world, rank, local_rank, launched = ctx
    if args.gpus > 1 and not launched:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if launched and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} under a launcher with WORLD_SIZE {world}; using {world}", file=sys.stderr)
    dev = init_device(ctx)
    out = None
    t_leg = time.perf_counter()

    def lap(what):                                   # wall time per leg on stderr (model construction + warm-up + timed region)
        nonlocal t_leg
        now = time.perf_counter()
        if rank == 0:
            print(f"[bench] {what}: {now - t_leg:.1f} s", file=sys.stderr, flush=True)
        t_leg = now
    if args.mode in ("all", "denoise"):
        out = run_denoise(args, ctx, dev)
        lap("denoise leg")
    if args.mode in ("all", "train", "train2"):
        # the train leg must never cost the headline line: an exception is reported inside the line, and a hang (a stuck
        # collective) is cut by a watchdog on rank 0 that prints what it has and leaves
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(1200.0) and rank == 0:
                res = dict(out or {}, train={"error": "train legs exceeded 1200 s"})
                print(json.dumps(res), flush=True)
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        def leg(stage):
            try:
                return run_train(args, ctx, dev, stage=stage)
            except Exception as e:                      # noqa: BLE001  (reported, not swallowed)
                import traceback
                traceback.print_exc()
                return {"error": f"{type(e).__name__}: {e}"}
        tr = leg(1) if args.mode in ("all", "train") else None
        lap("train leg")
        tr2 = leg(2) if args.mode in ("all", "train2") else None
        lap("train_stage2 leg")
        if not args.reference_pass_structure and not args.no_reference_leg and not args.distill_only and not args.sync_debug:
            # the like-for-like number beside the default one, in the same line (round-5 review): the same legs with every pass the reference runs,
            # a shorter timed region, no instrumented pass
            import copy
            ref_args = copy.copy(args)
            ref_args.reference_pass_structure, ref_args.no_roofline, ref_args.train_steps = True, True, min(args.train_steps, 8)
            for res, stage in ((tr, 1), (tr2, 2)):
                if isinstance(res, dict) and "error" not in res:
                    try:
                        r = run_train(ref_args, ctx, dev, stage=stage)
                    except Exception as e:              # noqa: BLE001  (reported, not swallowed)
                        r = {"error": f"{type(e).__name__}: {e}"}
                    if rank == 0:
                        res["reference_pass_structure"] = r if "error" in (r or {}) else {
                            k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")} | {
                            "pass_structure": r["config"]["pass_structure"], "per_iteration_type": r["config"]["per_iteration_type"],
                            "tflop_per_micro_batch": r["roofline"]["tflop_per_micro_batch"], "frac": r["roofline"]["frac"]}
                    lap(f"reference-pass-structure leg (stage {stage})")
        done.set()
        if args.mode == "train":
            out = tr
        elif args.mode == "train2":
            out = tr2
        elif rank == 0:
            out["train"], out["train_stage2"] = tr, tr2
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or launched:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
