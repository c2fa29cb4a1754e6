#!/bin/bash
# r05aa: ping-pong main loop in the whole-line 128 x 320 tile (tile 7; AF_GEMM3W_PP=0 keeps the lock-step loop): bit identity, tests, timing
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "conv3x3 or groupnorm_statistics or gemm" 2>&1 | tail -3 > gpurun_out/r05aa_tests.txt
cat gpurun_out/r05aa_tests.txt
timeout 600 python - > gpurun_out/r05aa_gemm3w_pp.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
convs = [(8, 64, 64, 320, 0, 320, 0, 1), (8, 64, 64, 320, 0, 320, 640, 1), (8, 64, 64, 320, 0, 320, 960, 1), (8, 64, 64, 640, 320, 320, 0, 1), (8, 32, 32, 640, 0, 640, 320, 2), (8, 32, 32, 640, 0, 640, 1280, 2),
         (8, 16, 16, 1280, 0, 1280, 2560, 4), (8, 16, 16, 1280, 0, 1280, 0, 4), (8, 8, 8, 1280, 0, 1280, 0, 16), (4, 64, 64, 320, 0, 320, 0, 1), (1, 64, 64, 320, 0, 320, 640, 1)]
for (B, H, W, c1, c2, co, cs, sp) in convs:
    x1 = rnd(B, H, W, c1)
    x2 = rnd(B, H, W, c2) if c2 else None
    w = rnd(co, c1 + c2, 3, 3) * 0.05
    skip = None
    if cs:
        pw = ops.pack_conv3x3_skip(w, None, rnd(co, cs, 1, 1) * 0.05, None, dev)
        skip = (rnd(B, H, W, cs), None)
    else:
        pw = ops.pack_conv3x3(w, None, dev)
    run = lambda: ops.conv3x3(x1, pw, x2=x2, skip=skip, tile=7, splits=sp)
    os.environ['AF_GEMM3W_PP'] = '0'
    ref = run().clone()
    os.environ['AF_GEMM3W_PP'] = '1'
    bad = sum(0 if torch.equal(run(), ref) else 1 for _ in range(20))
    fl = 2.0 * B * H * W * co * (9 * (c1 + c2) + cs)
    line = f"conv B{B} {H}x{W} {c1}+{c2}->{co} tail{cs} splits{sp}: ping-pong == lock-step {20 - bad}/20"
    for rep in range(2):
        for pp in (0, 1):
            os.environ['AF_GEMM3W_PP'] = str(pp)
            ms = timeit(run, 20)
            line += f" | pp{pp} {ms * 1e3:.1f} us {fl / ms / 1e9:.0f}"
    print(line, flush=True)
for (M, N, K, sp) in [(32768, 320, 320, 1), (32768, 320, 1280, 1), (32768, 320, 640, 1), (8192, 640, 2560, 1), (8192, 640, 640, 1), (2048, 1280, 5120, 2), (32768, 320, 960, 1)]:
    a, w = rnd(M, K), rnd(N, K) * 0.05
    pw = ops.pack_matrix(w, None, dev)
    run = lambda: ops.gemm(a, pw, tile=7, splits=sp)
    os.environ['AF_GEMM3W_PP'] = '0'
    ref = run().clone()
    os.environ['AF_GEMM3W_PP'] = '1'
    bad = sum(0 if torch.equal(run(), ref) else 1 for _ in range(20))
    line = f"gemm {M} {N} {K} splits{sp}: ping-pong == lock-step {20 - bad}/20"
    for rep in range(2):
        for pp in (0, 1):
            os.environ['AF_GEMM3W_PP'] = str(pp)
            ms = timeit(run, 20)
            line += f" | pp{pp} {ms * 1e3:.1f} us {2.0 * M * N * K / ms / 1e9:.0f}"
    print(line, flush=True)
PY
cat gpurun_out/r05aa_gemm3w_pp.txt
