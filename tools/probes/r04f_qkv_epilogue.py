"""q | k | v projection (transposed-V split) at the three U-Net levels: direct epilogue (AF_GEMM3_ABLATE=32) against the staged forms, one process.
Measured (r04f): 43.7 -> 40.9, 36.1 -> 34.8, 31.2 -> 30.7 us; the 128 x 160 two-workgroups-per-CU tile with the same epilogues: 41.8 / 33.9 / 32.0 (no gain, not wired)."""
import os, sys
os.environ["AF_GEMM3_ABLATE_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from adaface_dev_amd import ops
from bench_kernel import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (M, N, K, tok, tile) in ((32768, 960, 320, 4096, 7), (8192, 1920, 640, 1024, 8), (2048, 3840, 1280, 256, 8), (32768, 960, 320, 4096, 8)):
    a = torch.randn(M, K, generator=g).half().to(dev)
    pw = ops.pack_matrix(torch.randn(N, K, generator=g) * K ** -0.5, None, dev)
    res = {"32": [], "0": []}
    for r in range(3):
        for abl in ("32", "0"):
            os.environ["AF_GEMM3_ABLATE"] = abl
            res[abl].append(timeit(lambda: ops.gemm(a, pw, rows_per_batch=tok, split_col=N // 3 * 2, tile=tile), 30) * 1e3)
    print(f"qkv M{M} N{N} K{K} tile {tile}: direct epilogue {min(res['32']):.1f} us   staged {min(res['0']):.1f} us")
