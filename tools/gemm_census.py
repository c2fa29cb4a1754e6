#!/usr/bin/env python
"""Census of the GEMM launches of bench.py's training legs: how many times each (shape, epilogue) key is launched per micro-batch,
which (tile, split-K) the table gives it, and the time the autotune log measured for that choice.
    python tools/gemm_census.py [--stage 0|1|2] [--log profiles/r02s_autotune_all.log]        (stage 0 = one denoise step at U-Net batch 8)"""
import argparse
import ast
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--log", default=os.path.join(ROOT, "profiles", "r02s_autotune_all.log"))
    args = ap.parse_args()
    from adaface_dev_amd import ops
    import bench
    tuned = {}
    if os.path.exists(args.log):
        for line in open(args.log):
            if line.startswith("('"):
                try:
                    key, best, us, tf, _ = ast.literal_eval(line.strip())
                    tuned[key] = (us, tf)
                except Exception:
                    pass
    counts = collections.Counter()
    on = [False]

    def recorder(key, d, device):
        if on[0]:
            counts[key] += 1
        return ops.tune_table().get(key, (0, 1))
    ops._tune_recorder = recorder
    orig = bench.time.perf_counter
    # count only the timed micro-batches: run_train calls perf_counter right before and after them
    calls = [0]

    def pc():
        calls[0] += 1
        on[0] = 1 <= calls[0] <= 2 * n_mb + 1        # t0, then one call before and one after every timed micro-batch, then the closing call
        return orig()
    n_mb = 6
    bench.time.perf_counter = pc
    if args.stage == 0:                                  # the denoise leg: one U-Net forward at batch 8 = one step
        from adaface_dev_amd import SD15_UNET_CONFIG, rng
        from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
        dev = torch.device("cuda:0")
        with rng.skip_default_init():
            unet = UNetModel(**SD15_UNET_CONFIG)
        unet = unet.to(dev).eval()
        rng.load_synth_weights(unet, seed=0, on_device=True)
        x, ctx = rng.synth_input("bench.x", (8, 4, 64, 64), seed=1).to(dev), rng.synth_input("bench.ctx", (8, 77, 768), seed=1).to(dev)
        with torch.no_grad():
            unet(x, torch.full((8,), 500, device=dev), ctx, extra_info=None)
            on[0], n_mb = True, 1
            unet(x, torch.full((8,), 500, device=dev), ctx, extra_info=None)
            on[0] = False
    else:
        ns = argparse.Namespace(batch=4, no_ffn_lora=False, no_train_graphs=True, train_steps=n_mb, train_warmup=2, no_roofline=True, distill_only=False)
        bench.run_train(ns, (1, 0, 0, False), torch.device("cuda:0"), stage=args.stage)
    tot_us = 0.0
    rows = []
    for key, n in counts.items():
        us = tuned.get(key, (None, None))[0]
        rows.append((n / n_mb * (us or 0.0), key, n / n_mb, ops.tune_table().get(key), us))
        tot_us += n / n_mb * (us or 0.0)
    rows.sort(reverse=True)
    print(f"# stage {args.stage}: {sum(counts.values()) / n_mb:.0f} GEMM launches per micro-batch over {len(counts)} keys; "
          f"sum of tuned (warm, back-to-back) times {tot_us / 1e3:.2f} ms per micro-batch")
    print("#   us/mb  launches/mb  tuned_us  (tile, splits)  key = taps,M,N,K,act,out_mode,stride,upsample")
    for t, key, n, best, us in rows:
        print(f"{t:9.1f} {n:9.1f} {us if us is not None else float('nan'):9.1f}  {str(best):10s}  {key}")


if __name__ == "__main__":
    main()
