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
    python tools/autotune_instep.py [--keep 4] [--reps 5] [--out adaface-dev_amd/tuning/gfx950_gemm.json] [--log gpurun_out/instep.log]
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
    ap.add_argument("--keep", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--min-gain", type=float, default=0.02, help="a configuration replaces the table's only if it is this fraction faster in the step")
    ap.add_argument("--out", default=None)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    from adaface_dev_amd import SD15_UNET_CONFIG, _lib, ops, rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from autotune_gemm import candidate_ok

    dev = torch.device("cuda:0")
    L = _lib.lib()
    table = ops.tune_table()
    with rng.skip_default_init():
        unet = UNetModel(**SD15_UNET_CONFIG)
    unet = unet.to(dev).eval()
    rng.load_synth_weights(unet, seed=0, on_device=True)
    x = rng.synth_input("bench.x", (args.batch, 4, 64, 64), seed=1).to(dev)
    ctx = rng.synth_input("bench.ctx", (args.batch, 77, 768), seed=1).to(dev)
    ts = torch.full((args.batch,), 500, device=dev)

    def forward():
        with torch.no_grad():
            unet(x, ts, ctx, extra_info=None)

    def key_of(d):
        k = f"{d.taps},{d.M},{d.N},{d.K},{d.act},{d.out_mode},{d.stride},{d.upsample}"
        return k + ",ln" if d.ln_colsum else k

    real = L.af_gemm
    mode = {"what": "off", "key": None}
    shortlist, gn_keys, launches, events = {}, set(), {}, []

    def warm_time(d, tile, splits, reps=12):
        saved = (d.tile, d.splits, d.workspace, d.workspace_bytes, d.splitk_fused, d.gn_partials)
        d.tile, d.splits, d.splitk_fused, d.gn_partials = tile, splits, 0, 0
        if splits > 1:
            ws = ops._splitk_workspace(dev)
            if splits * d.M * d.N * 4 > ws.numel() * 4 - _lib.AF_SPLITK_COUNTER_BYTES:
                d.tile, d.splits, d.workspace, d.workspace_bytes, d.splitk_fused, d.gn_partials = saved
                return None
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        st = torch.cuda.current_stream().cuda_stream
        ok = all(real(C.byref(d), st) >= 0 for _ in range(2))
        t = None
        if ok:
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
            if key not in shortlist:
                cur = (d.tile, d.splits)                           # what the live path chose for this launch (table, LayerNorm-fold rule, heuristic)
                res = {}
                for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15):
                    for splits in (1, 2, 3, 4, 6, 8, 12, 16):
                        if (tile, splits) == cur or not candidate_ok(d, tile, splits, _lib, ops):
                            continue
                        if d.gn_partials and L.af_gemm_gn_stats_ok(tile, splits, d.taps, d.act, d.out_mode, d.N, d.gn_cpg,
                                                                  d.rows_per_batch if d.rows_per_batch > 0 else d.M) != 1:
                            continue
                        t = warm_time(d, tile, splits)
                        if t is not None:
                            res[(tile, splits)] = t
                best = sorted(res, key=res.get)[:args.keep]
                shortlist[key] = [cur] + best
                shortlist[key + "/warm"] = {f"{c[0]}x{c[1]}": round(res[c] * 1e3, 1) for c in best}
            return real(dref, st)
        if mode["what"] == "time" and key == mode["key"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = real(dref, st)
            e1.record()
            events.append((e0, e1))
            return rc
        return real(dref, st)

    L.af_gemm = hooked
    forward()
    torch.cuda.synchronize()
    mode["what"] = "census"
    forward()
    torch.cuda.synchronize()
    keys = [k for k in launches if k in shortlist]
    print(f"{len(keys)} shapes, {sum(launches[k] for k in keys)} launches per step; {len(gn_keys)} of the shapes write GroupNorm statistics", flush=True)
    mode["what"] = "time"
    log, changed, saved_us = [], 0, 0.0
    for key in sorted(keys, key=lambda k: -launches[k]):
        cur = shortlist[key][0]
        res = {}
        for cand in shortlist[key]:
            table[key] = cand
            sums = []
            for r in range(args.reps + 1):
                events.clear()
                mode["key"] = key
                forward()
                torch.cuda.synchronize()
                if r:                                               # the first forward with a new configuration also allocates
                    sums.append(sum(a.elapsed_time(b) for a, b in events) * 1e3)
            res[cand] = sorted(sums)[len(sums) // 2]
        best = min(res, key=res.get)
        if best != cur and res[best] < (1 - args.min_gain) * res[cur]:
            table[key] = best
            changed += 1
            saved_us += res[cur] - res[best]
        else:
            table[key] = cur
        line = (key, launches[key], f"{cur[0]}x{cur[1]}", f"{table[key][0]}x{table[key][1]}", {f"{c[0]}x{c[1]}": round(v, 1) for c, v in res.items()},
                shortlist.get(key + "/warm"))
        log.append(line)
        print(line, flush=True)
    print(f"{changed} of {len(keys)} shapes changed; {saved_us:.1f} us per step by the in-step sums", flush=True)
    L.af_gemm = real
    out = args.out or ops._TUNE_PATH
    with open(out, "w") as f:
        json.dump({k: list(v) for k, v in table.items()}, f, indent=0, sort_keys=True)
    if args.log:
        with open(args.log, "w") as f:
            for line in log:
                f.write(repr(line) + "\n")
            f.write(f"# {changed} of {len(keys)} shapes changed; {saved_us:.1f} us per step by the in-step sums\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
