#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python - > gpurun_out/r05v_debug.txt 2>&1 <<'PY'
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from adaface_dev_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half()
B, H, W, c1, co = 1, 16, 16, 128, 160
x1 = rnd(B, H, W, c1)
wfull = rnd(co, c1, 3, 3) * 0.1
for chunk in (0, 1):
    for tap in range(9):
        w = torch.zeros_like(wfull)
        w[:, chunk * 64:(chunk + 1) * 64, tap // 3, tap % 3] = wfull[:, chunk * 64:(chunk + 1) * 64, tap // 3, tap % 3]
        ref = F.conv2d(x1.float().permute(0, 3, 1, 2), w.float(), None, padding=1)
        pw = ops.pack_conv3x3(w, None, dev)
        errs = []
        for sp in (1, 2):
            o = ops.conv3x3(x1.to(dev), pw, tile=14, splits=sp).float().cpu().permute(0, 3, 1, 2)
            errs.append((o - ref).abs().max().item())
        print(f"chunk {chunk} tap {tap}: err splits1 {errs[0]:.2e} splits2 {errs[1]:.2e} |ref| {ref.abs().max().item():.2f}", flush=True)
PY
cat gpurun_out/r05v_debug.txt
