#!/usr/bin/env python
"""In-process A/B of main-loop variants of the whole-line GEMM kernel (AF_GEMM3_ABLATE switches) on the shapes that carry the
denoise step: interleaved rounds, graph-replayed timing, median and min per (shape, variant).
    AF_GEMM3_ABLATE_DYNAMIC=1 python tools/exp_gemm_variants.py [rounds]"""
import os
import sys

os.environ["AF_GEMM3_ABLATE_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from bench_kernel import timeit  # noqa: E402


def main():
    from adaface_dev_amd import ops
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = [int(v) for v in os.environ.get("AF_EXP_VARIANTS", "0,128,256,384").split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).half().to(dev)
    cases = []
    # (name, callable factory): convs B H W Cin Cout tile splits ; gemms M N K tile splits act
    for B, H, W, ci, co, tile, sp in ((8, 64, 64, 320, 320, 7, 1), (8, 32, 32, 640, 640, 7, 2), (8, 16, 16, 1280, 1280, 7, 4), (8, 64, 64, 640, 320, 7, 1)):
        x = rnd(B, H, W, ci)
        pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5, torch.randn(co, generator=g), dev)
        cases.append((f"conv {B}x{H}x{W} {ci}->{co} t{tile} s{sp}", 2.0 * B * H * W * co * ci * 9,
                      lambda x=x, pw=pw, tile=tile, sp=sp: ops.conv3x3(x, pw, tile=tile, splits=sp)))
    for M, N, K, tile, sp, act in ((32768, 320, 320, 7, 1, 0), (8192, 640, 640, 8, 1, 0), (2048, 1280, 1280, 8, 1, 0), (2048, 1280, 1280, 8, 2, 0),
                                   (32768, 2560, 320, 10, 1, 2), (8192, 5120, 640, 7, 1, 2), (32768, 320, 1280, 7, 1, 0), (8192, 640, 2560, 8, 1, 0)):
        a = rnd(M, K)
        pw = ops.pack_matrix(torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g), dev)
        cases.append((f"gemm {M} {N} {K} t{tile} s{sp} act{act}", 2.0 * M * N * K,
                      lambda a=a, pw=pw, tile=tile, sp=sp, act=act: ops.gemm(a, pw, tile=tile, splits=sp, act=act)))
    if os.environ.get("AF_EXP_SPLITK"):
        # fused (in-kernel) vs two-launch split-K on the small / mid shapes where a split can pay: variant = SPLITK_FUSED_MAX (0 = never fuse)
        cases, variants = [], [0, 4]
        for M, N, K, tile, sp in ((2048, 1280, 1280, 8, 1), (2048, 1280, 1280, 8, 2), (2048, 1280, 1280, 2, 2), (8192, 640, 640, 8, 2), (512, 1280, 1280, 2, 1), (512, 1280, 1280, 2, 2),
                                  (512, 1280, 1280, 2, 4), (388, 768, 768, 2, 1), (388, 768, 768, 2, 2), (388, 768, 768, 2, 4), (388, 3072, 768, 2, 2), (388, 768, 3072, 2, 4), (2048, 1280, 5120, 8, 3)):
            a = rnd(M, K)
            pw = ops.pack_matrix(torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g), dev)
            cases.append((f"gemm {M} {N} {K} t{tile} s{sp}", 2.0 * M * N * K, lambda a=a, pw=pw, tile=tile, sp=sp: ops.gemm(a, pw, tile=tile, splits=sp)))
        for B, H, W, ci, co, tile, sp in ((8, 16, 16, 1280, 1280, 7, 4), (8, 32, 32, 640, 640, 7, 2), (8, 8, 8, 1280, 1280, 7, 4)):
            x = rnd(B, H, W, ci)
            pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5, torch.randn(co, generator=g), dev)
            cases.append((f"conv {B}x{H}x{W} {ci}->{co} t{tile} s{sp}", 2.0 * B * H * W * co * ci * 9, lambda x=x, pw=pw, tile=tile, sp=sp: ops.conv3x3(x, pw, tile=tile, splits=sp)))
    if os.environ.get("AF_EXP_TILES"):
        # tile shapes against each other on the same operands: variant = tile id (7 = 128x320 / 8 waves, 8 = 128x128, 11 = 128x160 two workgroups per CU)
        cases, variants = [], [int(v) for v in os.environ["AF_EXP_TILES"].split(",")]
        tile_of = [7]
        for B, H, W, ci, co, sp in ((8, 64, 64, 320, 320, 1), (8, 32, 32, 640, 640, 2), (8, 16, 16, 1280, 1280, 4), (8, 64, 64, 640, 320, 1), (8, 32, 32, 1280, 640, 2),
                                    (8, 64, 64, 960, 320, 1), (8, 32, 32, 640, 640, 1), (2, 64, 64, 320, 320, 2), (4, 64, 64, 320, 320, 1)):
            x = rnd(B, H, W, ci)
            pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5, torch.randn(co, generator=g), dev)
            cases.append((f"conv {B}x{H}x{W} {ci}->{co} s{sp}", 2.0 * B * H * W * co * ci * 9, lambda x=x, pw=pw, sp=sp: ops.conv3x3(x, pw, tile=tile_of[0], splits=sp)))
        for M, N, K, sp in ((32768, 320, 320, 1), (32768, 320, 1280, 1), (8192, 640, 640, 1), (8192, 640, 2560, 1), (2048, 1280, 1280, 1), (2048, 1280, 5120, 3), (32768, 960, 320, 1)):
            a = rnd(M, K)
            pw = ops.pack_matrix(torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g), dev)
            r = rnd(M, N)
            cases.append((f"gemm {M} {N} {K} s{sp}", 2.0 * M * N * K, lambda a=a, pw=pw, sp=sp, r=r: ops.gemm(a, pw, tile=tile_of[0], splits=sp, residual=r)))
        for name, fl, fn in cases:                      # same results from every tile
            outs = []
            for v in variants:
                tile_of[0] = v
                outs.append(fn().float())
            for o in outs[1:]:
                err = float((o - outs[0]).norm() / outs[0].norm())
                assert err < 2e-3, (name, err)
    res = {(n, v): [] for n, _, _ in cases for v in variants}
    for r in range(rounds):
        for name, fl, fn in cases:
            for v in variants:
                if os.environ.get("AF_EXP_TILES"):
                    tile_of[0] = v
                elif os.environ.get("AF_EXP_SPLITK"):
                    ops.SPLITK_FUSED_MAX = v
                else:
                    os.environ["AF_GEMM3_ABLATE"] = str(v)
                res[(name, v)].append(timeit(fn, 20) * 1e3)
    print(f"{'shape':44s} " + " ".join(f"{'v' + str(v):>16s}" for v in variants) + "   (us median/min, TFLOP/s at median)")
    for name, fl, _ in cases:
        cells = []
        for v in variants:
            t = sorted(res[(name, v)])
            med = t[len(t) // 2]
            cells.append(f"{med:6.1f}/{t[0]:6.1f} {fl / med / 1e6:4.0f}")
        print(f"{name:44s} " + " ".join(f"{c:>16s}" for c in cells))


if __name__ == "__main__":
    main()
