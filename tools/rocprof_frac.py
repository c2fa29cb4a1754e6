#!/usr/bin/env python3
"""GEMM-family time per denoise step and its MFMA-roofline fraction from a rocprofv3 kernel-stats summary (tools/rocpd_summary.py output of the
graph-replayed denoise leg: profiles/<tag>_bench_kernel_stats.txt, last 10 steps) -> a small JSON bench.py reports beside its own instrumented figure
(`roofline.rocprof`): the kernels that produced ms_per_step, not an eager instrumented pass.

    python tools/rocprof_frac.py profiles/r06z_bench_kernel_stats.txt [--steps 10] [--json profiles/r06z_rocprof_frac.json]
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEMM_FAMILY = ("af_conv3h", "af_gemm3w_kernel", "af_gemm3_kernel", "af_gemm_kernel", "af_gemm_direct", "af_ff320", "af_gn_proj320", "af_splitk_reduce")
GEMM_TFLOP_PER_STEP = 677.23e-3 * 8          # conv3x3 + Linear + conv1x1 per U-Net sample x U-Net batch 8
PEAK = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    fam, total, rows = 0.0, 0.0, []
    for line in open(a.stats):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)$", line)
        if not m:
            continue
        calls, ms, name = int(m.group(1)), float(m.group(2)), m.group(5)
        total += ms
        if any(k in name for k in GEMM_FAMILY):
            fam += ms
            rows.append((name.strip(), calls / a.steps, ms / a.steps))
    fam_step = fam / a.steps
    from adaface_dev_amd import _lib
    out = {"file": os.path.basename(a.stats), "steps": a.steps, "gemm_family_ms_per_step": round(fam_step, 4), "kernel_ms_per_step": round(total / a.steps, 4),
           "achieved_tflops": round(GEMM_TFLOP_PER_STEP / (fam_step * 1e-3), 1), "frac": round(GEMM_TFLOP_PER_STEP / (fam_step * 1e-3) / PEAK, 4),
           "what": "GEMM / implicit-conv family under rocprofv3 --kernel-trace of the graph-replayed denoise leg: 5.418 TFLOP per step / summed kernel time",
           "sources_sha": _lib.sources_sha()}
    print(json.dumps(out, indent=1))
    for r in sorted(rows, key=lambda r: -r[2]):
        print(f"  {r[2]:8.4f} ms/step  {r[1]:6.1f} launches/step  {r[0][:90]}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
