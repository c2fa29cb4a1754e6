#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export AF_GEMM3_ABLATE_DYNAMIC=1 AF_CONV3H_PP=1
timeout 600 python - > gpurun_out/r05t_conv3hp_prio.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
for (B, H, W, ci, co, sp) in [(8, 64, 64, 320, 320, 1), (8, 32, 32, 640, 640, 2), (8, 16, 16, 1280, 1280, 4)]:
    x, w = rnd(B, H, W, ci), rnd(co, ci, 3, 3) * 0.05
    pw = ops.pack_conv3x3(w, None, dev)
    line = f"conv B{B} {H}x{W} {ci}->{co} splits{sp}:"
    for rep in range(2):
        for name, bits in (("prio-on-mfma", 0), ("no-prio", 32), ("prio-on-loaders", 64)):
            os.environ['AF_GEMM3_ABLATE'] = str(bits)
            ms = timeit(lambda: ops.conv3x3(x, pw, tile=14, splits=sp), 20)
            line += f" | {name} {ms * 1e3:.1f}"
    print(line, flush=True)
PY
cat gpurun_out/r05t_conv3hp_prio.txt
