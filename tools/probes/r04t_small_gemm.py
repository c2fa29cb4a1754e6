"""The register-staged 64 x 64 GEMM (tile 2) on the small shapes that fill the training legs (profiles/r04s_gemm_census_stage2.txt): one register stage
(AF_GEMM_PF=1, the kernel of rounds 1-3) against the ring of four, per split-K count; alternating in one process.  Times include the split-K reduce launch."""
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
shapes = ((4096, 320, 320), (256, 1280, 1280), (1024, 640, 640), (4096, 640, 640), (1024, 1280, 1280), (97, 320, 768), (64, 1280, 1280), (256, 1280, 5120),
          (512, 1280, 1280), (4096, 320, 1280), (388, 768, 768), (388, 3072, 768), (388, 768, 3072), (8192, 320, 320), (2048, 640, 640))
for (M, N, K) in shapes:
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    pw = ops.pack_matrix(w, None, dev)
    ref = a.float() @ w.to(dev).t()
    line = f"M{M:5d} N{N:5d} K{K:5d}:"
    for sp in (1, 2, 3, 4, 8):
        if sp > K // 64:
            continue
        t = {}
        for r in range(3):
            for pf in ("1", "4"):
                os.environ["AF_GEMM_PF"] = pf
                t.setdefault(pf, []).append(timeit(lambda: ops.gemm(a, pw, tile=2, splits=sp), 40) * 1e3)
        os.environ["AF_GEMM_PF"] = "4"
        err = float((ops.gemm(a, pw, tile=2, splits=sp).float() - ref).norm() / ref.norm())
        assert err < 2e-3, (M, N, K, sp, err)
        line += f"  x{sp}: {min(t['1']):5.1f} -> {min(t['4']):5.1f}"
    print(line, flush=True)
