#!/usr/bin/env python
"""Second-stage tuner of af_gemm's (tile, split-K) table: choose among each shape's best few configurations by what they cost INSIDE the denoise step.

tools/autotune_gemm.py times a configuration back to back on live operands -- everything it reads is then L2-resident.  Inside the step a GEMM meets its
weights cold (1.7 GB of them stream from HBM once per step) and its activations as the previous launch left them, and the per-shape sums show it: the
family takes 7.0 ms back to back, 8.8 ms with every operand evicted, 8.6 ms in the step (profiles/r04v_gemm_breakdown_*.txt).  Deeper rings and other
split counts can win there although they lose warm.  This tool therefore
  1. runs the U-Net forward (bench.py's denoise shapes: batch 8, 64 x 64 latent, 77 tokens) once and, per GEMM shape, times every admissible configuration
     warm on the live operands (as the first-stage tuner does) to get a short list: the table's choice + the best --keep others;
  2. for every shape, puts each short-listed configuration into the live table and runs the eager forward --reps times with HIP events around THAT shape's
     launches (the real path: GroupNorm statistics from the producer, fused blocks and all); the median of the per-forward sums decides.
Shapes whose launches write GroupNorm partial statistics only try configurations that still can (af_gemm_gn_stats_ok).
--leg distill | recon | train2 does the same inside micro-batches of bench.py's training legs (eager launches), leaving the shapes an earlier run
tuned alone (--protect its log).  Every shape's j-th candidate is live at the same time and each shape is timed by its own events, so the number of
workload runs does not grow with the number of shapes.
    python tools/autotune_instep.py [--leg denoise] [--keep 4] [--reps 5] [--out adaface-dev_amd/tuning/gfx950_gemm.json] [--log gpurun_out/instep.log]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=["denoise", "distill", "recon", "train2"], default="denoise",
                    help="the workload whose launches are tuned: bench.py's denoise step, or micro-batches of its training legs (eager, no hipGraph segments)")
    ap.add_argument("--keep", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--min-gain", type=float, default=0.02, help="a configuration replaces the table's only if it is this fraction faster in the step")
    ap.add_argument("--min-us", type=float, default=1.0, help="... and saves at least this many microseconds per unit of the workload")
    ap.add_argument("--protect", default="", help="comma-separated logs of earlier runs of this tool: the shapes they tuned are left alone (a training leg must not re-tune the denoise step's shapes)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    from adaface_dev_amd import SD15_UNET_CONFIG, _lib, ops, rng
    from autotune_gemm import candidate_ok

    dev = torch.device("cuda:0")
    L = _lib.lib()
    table = ops.tune_table()
    protected = set()
    for path in filter(None, args.protect.split(",")):
        import ast
        for line in open(path):
            if line.startswith("('"):
                protected.add(ast.literal_eval(line)[0])

    if args.leg == "denoise":
        from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
        with rng.skip_default_init():
            unet = UNetModel(**SD15_UNET_CONFIG)
        unet = unet.to(dev).eval()
        rng.load_synth_weights(unet, seed=0, on_device=True)
        x = rng.synth_input("bench.x", (args.batch, 4, 64, 64), seed=1).to(dev)
        ctx = rng.synth_input("bench.ctx", (args.batch, 77, 768), seed=1).to(dev)
        ts = torch.full((args.batch,), 500, device=dev)

        def unit():
            with torch.no_grad():
                unet(x, ts, ctx, extra_info=None)
    else:
        import bench
        ns = argparse.Namespace(batch=4, no_ffn_lora=False, no_train_graphs=True, train_steps=2, train_warmup=2, no_roofline=True,
                                distill_only=args.leg == "distill")
        tr, batches, step_kw, B, n_train, *_keep = bench.build_train(ns, (1, 0, 0, False), dev, stage=2 if args.leg == "train2" else 1)
        idx = [0]

        def unit():
            # one repeatable unit of the leg: the 2,3,4-step cycle of the distillation iteration; one recon iteration on the images and one from pure noise;
            # one compositional iteration with normalised and one with mixed scores
            if args.leg == "distill":
                for _ in range(3):
                    tr.training_step(batches[idx[0] % 4], idx[0])
                    idx[0] += 1
            elif args.leg == "train2":
                for aug in ("normalize_cross_attn", "mix_sc_mc_attn"):
                    tr.training_step(batches[idx[0] % 4], idx[0], attn_aug=aug)
                    idx[0] += 1
            else:
                for noise in (False, True):
                    torch.manual_seed(1234)
                    tr.optimizer.zero_grad()
                    loss = tr.normal_recon_step(batches[idx[0] % 4], on_pure_noise=noise)
                    (loss * tr.scaler.scale).backward()
                    idx[0] += 1
                tr.optimizer.zero_grad()

    def key_of(d):
        k = f"{d.taps},{d.M},{d.N},{d.K},{d.act},{d.out_mode},{d.stride},{d.upsample}"
        return k + ",ln" if d.ln_colsum else k

    real = L.af_gemm
    mode = {"what": "off"}
    shortlist, warm, gn_keys, launches, events = {}, {}, set(), {}, {}

    def warm_time(d, tile, splits, reps=12):
        saved = (d.tile, d.splits, d.workspace, d.workspace_bytes, d.splitk_fused, d.gn_partials)
        d.tile, d.splits, d.splitk_fused, d.gn_partials = tile, splits, 0, 0
        t = None
        ok = True
        if splits > 1:
            ws = ops._splitk_workspace(dev)
            ok = splits * d.M * d.N * 4 <= ws.numel() * 4 - _lib.AF_SPLITK_COUNTER_BYTES
            if ok:
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        st = torch.cuda.current_stream().cuda_stream
        if ok and all(real(C.byref(d), st) >= 0 for _ in range(2)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                real(C.byref(d), st)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / reps
        d.tile, d.splits, d.workspace, d.workspace_bytes, d.splitk_fused, d.gn_partials = saved
        return t

    def hooked(dref, st):
        d = dref._obj
        key = key_of(d)
        if mode["what"] == "census":
            launches[key] = launches.get(key, 0) + 1
            if d.gn_partials:
                gn_keys.add(key)
            if key not in shortlist and key not in protected:
                cur = (d.tile, d.splits)                           # what the live path chose for this launch (table, LayerNorm-fold rule, heuristic)
                res = {}
                for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17):
                    for splits in (1, 2, 3, 4, 6, 8, 12, 16):
                        if (tile, splits) == cur or not candidate_ok(d, tile, splits, _lib, ops):
                            continue
                        if d.gn_partials and L.af_gemm_gn_stats_ok(tile, splits, d.taps, d.act, d.out_mode, d.N, d.gn_cpg,
                                                                  d.rows_per_batch if d.rows_per_batch > 0 else d.M) != 1:
                            continue
                        t = warm_time(d, tile, splits, reps=12 if args.leg == "denoise" else 6)
                        if t is not None:
                            res[(tile, splits)] = t
                best = sorted(res, key=res.get)[:args.keep]
                shortlist[key] = [cur] + best
                warm[key] = {f"{c[0]}x{c[1]}": round(res[c] * 1e3, 1) for c in best}
            return real(dref, st)
        if mode["what"] == "time" and key in shortlist:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = real(dref, st)
            e1.record()
            events.setdefault(key, []).append((e0, e1))
            return rc
        return real(dref, st)

    L.af_gemm = hooked
    unit()
    torch.cuda.synchronize()
    mode["what"] = "census"
    unit()
    torch.cuda.synchronize()
    keys = [k for k in launches if k in shortlist]
    print(f"{args.leg}: {len(keys)} shapes to tune ({len(launches) - len(keys)} protected), {sum(launches[k] for k in keys)} launches per unit; "
          f"{len(gn_keys)} of the shapes write GroupNorm statistics", flush=True)
    # every shape's j-th candidate goes into the live table at once: the shapes are timed separately (events per shape), so one series of units serves all
    mode["what"] = "time"
    orig = {k: table.get(k) for k in keys}
    res = {k: {} for k in keys}
    for j in range(args.keep + 1):
        for k in keys:
            table[k] = shortlist[k][j] if j < len(shortlist[k]) else shortlist[k][0]
        sums = {k: [] for k in keys}
        for r in range(args.reps + 1):
            events.clear()
            unit()
            torch.cuda.synchronize()
            if r:                                                   # the first unit with new configurations also allocates
                for k in keys:
                    sums[k].append(sum(a.elapsed_time(b) for a, b in events.get(k, [])) * 1e3)
        for k in keys:
            if j < len(shortlist[k]) and sums[k]:
                res[k][shortlist[k][j]] = sorted(sums[k])[len(sums[k]) // 2]
        print(f"candidate {j}: {sum(sorted(v)[len(v) // 2] for v in sums.values() if v) / 1e3:.2f} ms of tuned GEMM launches per unit", flush=True)
    log, changed, saved_us = [], 0, 0.0
    for k in sorted(keys, key=lambda k: -res[k].get(shortlist[k][0], 0.0)):
        cur = shortlist[k][0]
        best = min(res[k], key=res[k].get)
        if best != cur and res[k][best] < (1 - args.min_gain) * res[k][cur] and res[k][cur] - res[k][best] >= args.min_us:
            table[k] = best
            changed += 1
            saved_us += res[k][cur] - res[k][best]
        elif orig[k] is None:
            table.pop(k, None)                                       # no entry before (the library's heuristic, or a LayerNorm key served by its base key)
        else:
            table[k] = orig[k]
        log.append((k, launches[k], f"{cur[0]}x{cur[1]}", "x".join(map(str, table.get(k, cur))), {f"{c[0]}x{c[1]}": round(v, 1) for c, v in res[k].items()}, warm.get(k)))
    for line in log[:60]:
        print(line, flush=True)
    tail = f"{args.leg}: {changed} of {len(keys)} shapes changed; {saved_us:.1f} us per unit by the in-step sums"
    print(tail, flush=True)
    L.af_gemm = real
    out = args.out or ops._TUNE_PATH
    with open(out, "w") as f:
        json.dump({k: list(v) for k, v in table.items()}, f, indent=0, sort_keys=True)
    if args.log:
        with open(args.log, "w") as f:
            for line in log:
                f.write(repr(line) + "\n")
            f.write("# " + tail + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
