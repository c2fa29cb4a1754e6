#!/bin/bash
# r05s: where a stage of the ping-pong halo kernel goes: in-kernel ablation (AF_GEMM3_ABLATE bits, results are wrong by construction)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export AF_GEMM3_ABLATE_DYNAMIC=1 AF_CONV3H_PP=1
timeout 600 python - > gpurun_out/r05s_conv3hp_ablate.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
for (B, H, W, ci, co, sp) in [(8, 64, 64, 320, 320, 1), (1, 64, 64, 320, 320, 1), (8, 16, 16, 1280, 1280, 4)]:
    x, w = rnd(B, H, W, ci), rnd(co, ci, 3, 3) * 0.05
    pw = ops.pack_conv3x3(w, None, dev)
    line = f"conv B{B} {H}x{W} {ci}->{co} splits{sp}:"
    for name, bits in (("full", 0), ("no-dma", 1), ("frags-once", 2), ("no-barrier", 4), ("no-mfma", 16), ("no-dma+frags-once", 3), ("no-dma,frags,barrier", 7), ("only-mfma+barrier", 3), ("no-loop", 8), ("dma+barrier only", 18), ("reads+barrier only", 17), ("barriers only", 19)):
        os.environ['AF_GEMM3_ABLATE'] = str(bits)
        ms = min(timeit(lambda: ops.conv3x3(x, pw, tile=14, splits=sp), 20) for _ in range(2))
        line += f" | {name} {ms * 1e3:.1f}"
    print(line, flush=True)
PY
cat gpurun_out/r05s_conv3hp_ablate.txt
