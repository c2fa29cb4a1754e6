#!/usr/bin/env python
"""Same-process A/B of the fused C = 320 feed-forward (af_ff_fused) against [GEGLU GEMM with the folded LayerNorm + output GEMM] at the
64 x 64 level of a U-Net batch-8 step.    python tools/bench_ff.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench_kernel import timeit  # noqa: E402


def main():
    from adaface_dev_amd import ops
    from adaface_dev_amd.ldm.modules.attention import FeedForward
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    C = 320
    ff, ln = FeedForward(C, glu=True).to(dev), LayerNorm(C).to(dev)
    for M in (32768, 16384, 4096):
        x = torch.randn(M, C, device=dev).half()
        pw1, pw2 = ff.net[0].packed_ln(ln), ff.net[2].packed()
        t_a = timeit(lambda: ops.gemm(x, pw1, act=ops.AF_ACT_GEGLU), reps) * 1e3
        h = ops.gemm(x, pw1, act=ops.AF_ACT_GEGLU)
        t_b = timeit(lambda: ops.gemm(h, pw2, residual=x), reps) * 1e3
        t_pair = timeit(lambda: ops.gemm(ops.gemm(x, pw1, act=ops.AF_ACT_GEGLU), pw2, residual=x), reps) * 1e3
        t_f = timeit(lambda: ops.ff_fused(x, pw1, pw2, residual=x), reps) * 1e3
        fl = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
        print(f"M{M:6d}: GEGLU GEMM {t_a:6.1f} + output GEMM {t_b:6.1f} = pair {t_pair:6.1f} us ({fl / t_pair / 1e6:6.1f} TFLOP/s) | fused {t_f:6.1f} us ({fl / t_f / 1e6:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
