#!/usr/bin/env python
"""Per-shape breakdown of the GEMM family in one denoise step (U-Net batch 8): launches, time at the tuned (tile, splits),
TFLOP/s, share of the family.  GPU box:  python tools/gemm_breakdown.py [--batch 8]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--warm-weights", action="store_true", help="with --cold: af_prefetch the packed weights after the flush (what a side-stream prefetcher would achieve)")
    ap.add_argument("--cold", action="store_true", help="evict L2 / Infinity Cache (a 1 GB fill) before every timed launch: cold operands, as in the real step where 1.7 GB of weights stream from HBM")
    args = ap.parse_args()
    from adaface_dev_amd import SD15_UNET_CONFIG, _lib, ops, rng
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    dev = torch.device("cuda:0")
    L = _lib.lib()
    table = ops.tune_table()
    seen = {}

    def recorder(key, d, device):
        ts = table.get(key, (0, 0))
        if key not in seen:
            dd = type(d)()
            C.memmove(C.byref(dd), C.byref(d), C.sizeof(d))
            seen[key] = [0, dd, tuple(ts)]
        seen[key][0] += 1
        return ts

    unet = UNetModel(**SD15_UNET_CONFIG)
    rng.load_synth_weights(unet, seed=0)
    unet = unet.to(dev).eval()
    b = args.batch
    x = rng.synth_input("bench.x", (b, 4, 64, 64), seed=1).to(dev)
    ctx = rng.synth_input("bench.ctx", (b, 77, 768), seed=1).to(dev)
    keep = []
    with torch.no_grad():
        unet(x, torch.full((b,), 500, device=dev), ctx, extra_info=None)      # packs
        ops._tune_recorder = recorder
        # keep every intermediate alive so the recorded operand pointers stay valid: disable the caching allocator's reuse
        os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
        unet(x, torch.full((b,), 500, device=dev), ctx, extra_info=None)
        ops._tune_recorder = None
    torch.cuda.synchronize()
    big = torch.empty(1 << 28, dtype=torch.float16, device=dev)               # 512 MB scratch: any operand pointer we re-aim lands here
    st = torch.cuda.current_stream().cuda_stream
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if args.cold else None
    rows = []
    for key, (cnt, d, ts) in seen.items():
        # re-aim activation operands / outputs at the scratch (timing only; weights stay real)
        for f in ("a1", "a2", "residual", "out", "out2", "rowbias"):
            if getattr(d, f):
                setattr(d, f, big.data_ptr())
        d.tile, d.splits = ts[0], ts[1] if len(ts) > 1 else 0
        d.zeros = ops._zero_page(dev).data_ptr()
        if d.splits > 1:
            ws = ops._splitk_workspace(dev)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        for _ in range(3):
            rc = L.af_gemm(C.byref(d), st)
        if rc < 0:
            print("skip", key, L.af_last_error())
            continue
        if args.cold:
            us = 0.0
            for _ in range(5):
                flush.fill_(1.0)                                      # 1 GB of writes: nothing of the operands stays in L2 / MALL
                if args.warm_weights:
                    L.af_prefetch(d.wt, int(d.kpad) * ((int(d.N) + 127) // 128 * 128) * 2, st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.af_gemm(C.byref(d), st)
                e1.record()
                e1.synchronize()
                us += e0.elapsed_time(e1) / 5 * 1e3
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                L.af_gemm(C.byref(d), st)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
        fl = 2.0 * d.M * d.N * d.K
        rows.append((cnt * us, key, cnt, us, fl / us / 1e6, ts))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    flops = sum(2.0 * seen[r[1]][1].M * seen[r[1]][1].N * seen[r[1]][1].K * r[2] for r in rows)
    print(f"GEMM family: {len(rows)} shapes, {sum(r[2] for r in rows)} launches, {tot / 1e3:.2f} ms back-to-back, {flops / tot / 1e6:.0f} TFLOP/s")
    print("key = taps,M,N,K,act,out_mode,stride,upsample")
    acc = 0.0
    for t, key, cnt, us, tf, ts in rows:
        acc += t
        print(f"{key:38s} x{cnt:3d} {us:8.1f} us {tf:7.1f} TF/s  tile,split={ts}  {100 * t / tot:5.1f}%  cum {100 * acc / tot:5.1f}%")


if __name__ == "__main__":
    main()
