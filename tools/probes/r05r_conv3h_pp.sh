#!/bin/bash
# r05r: the halo-resident 3x3 kernel with the ping-pong main loop (tile 14) against the tap-by-tap 128 x 320 tile (7): parity, repeatability, timing
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "conv3x3 or groupnorm_statistics or gemm" 2>&1 | tail -5 > gpurun_out/r05r_tests.txt
cat gpurun_out/r05r_tests.txt
timeout 600 python - > gpurun_out/r05r_conv3h_pp.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
# (B, H, W, c1, c2, cout, skip channels, splits)
shapes = [(8, 64, 64, 320, 0, 320, 0, 1), (8, 64, 64, 320, 320, 320, 0, 1), (8, 64, 64, 640, 320, 320, 0, 1), 
          (8, 32, 32, 640, 0, 640, 0, 2), (8, 32, 32, 640, 640, 640, 0, 1), (8, 32, 32, 1280, 640, 640, 0, 1), 
          (8, 16, 16, 1280, 0, 1280, 0, 4), (8, 16, 16, 1280, 1280, 1280, 0, 4)]
for (B, H, W, c1, c2, co, cs, sp) in shapes:
    x1 = rnd(B, H, W, c1)
    x2 = rnd(B, H, W, c2) if c2 else None
    w = rnd(co, c1 + c2, 3, 3) * 0.05
    skip = None
    if cs:
        pw = ops.pack_conv3x3_skip(w, None, rnd(co, cs, 1, 1) * 0.05, None, dev)
        skip = (rnd(B, H, W, cs), None)
    else:
        pw = ops.pack_conv3x3(w, None, dev)
    run = lambda tile: ops.conv3x3(x1, pw, x2=x2, skip=skip, tile=tile, splits=sp)
    ref7 = run(7).clone()
    ref = run(14).clone()
    bad = sum(0 if torch.equal(run(14), ref) else 1 for _ in range(20))
    d7 = (ref.float() - ref7.float()).abs().max().item()
    fl = 2.0 * B * H * W * co * (9 * (c1 + c2) + cs)
    line = f"conv B{B} {H}x{W} {c1}+{c2}->{co} tail{cs} splits{sp}: repeatable {20 - bad}/20, max |t14 - t7| {d7:.1e}"
    for rep in range(2):
        for tile in (7, 14):
            ms = timeit(lambda: run(tile), 20)
            line += f" | t{tile} {ms * 1e3:.1f} us {fl / ms / 1e9:.0f}"
    print(line, flush=True)
PY
cat gpurun_out/r05r_conv3h_pp.txt
