#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "conv3x3 or groupnorm_statistics" 2>&1 | tail -3 > gpurun_out/r05ah_tests.txt
cat gpurun_out/r05ah_tests.txt
timeout 600 python - > gpurun_out/r05ah_8x8.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
for (B, H, W, c1, c2, co, cfgs) in [(8, 8, 8, 1280, 0, 1280, ((1, 12), (7, 16), (14, 4), (14, 8), (14, 12), (14, 16), (14, 20))), (8, 8, 8, 1280, 1280, 1280, ((1, 12), (7, 16), (14, 8), (14, 16), (14, 20), (14, 40)))]:
    x1 = rnd(B, H, W, c1)
    x2 = rnd(B, H, W, c2) if c2 else None
    w = rnd(co, c1 + c2, 3, 3) * 0.05
    pw = ops.pack_conv3x3(w, None, dev)
    fl = 2.0 * B * H * W * co * 9 * (c1 + c2)
    ref = ops.conv3x3(x1, pw, x2=x2, tile=7, splits=16).float()
    line = f"conv B{B} {H}x{W} {c1}+{c2}->{co}:"
    for rep in range(2):
        for (tile, sp) in cfgs:
            y = ops.conv3x3(x1, pw, x2=x2, tile=tile, splits=sp)
            err = (y.float() - ref).abs().max().item()
            ms = timeit(lambda: ops.conv3x3(x1, pw, x2=x2, tile=tile, splits=sp), 20)
            line += f" | t{tile}x{sp} {ms * 1e3:.1f} us (max diff {err:.1e})" if rep == 0 else f" | t{tile}x{sp} {ms * 1e3:.1f}"
    print(line, flush=True)
PY
cat gpurun_out/r05ah_8x8.txt
