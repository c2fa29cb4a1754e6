#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "conv3x3 or groupnorm_statistics" 2>&1 | tail -3 > gpurun_out/r05ab_tests.txt
cat gpurun_out/r05ab_tests.txt
timeout 600 python - > gpurun_out/r05ab_ups.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
for (B, H, W, ci, co, cfgs) in [(8, 32, 32, 640, 640, ((7, 1), (14, 1))), (8, 16, 16, 1280, 1280, ((7, 1), (14, 1), (14, 2))), (8, 8, 8, 1280, 1280, ((7, 4), (14, 4), (14, 2), (14, 8)))]:
    x, w = rnd(B, H, W, ci), rnd(co, ci, 3, 3) * 0.05
    pw = ops.pack_conv3x3(w, None, dev)
    fl = 2.0 * B * 4 * H * W * co * 9 * ci
    line = f"upsample conv B{B} {H}x{W}->{2*H}x{2*W} {ci}->{co}:"
    for rep in range(2):
        for (tile, sp) in cfgs:
            ms = timeit(lambda: ops.conv3x3(x, pw, upsample=True, tile=tile, splits=sp), 20)
            line += f" | t{tile}x{sp} {ms * 1e3:.1f} us {fl / ms / 1e9:.0f}"
    print(line, flush=True)
PY
cat gpurun_out/r05ab_ups.txt
