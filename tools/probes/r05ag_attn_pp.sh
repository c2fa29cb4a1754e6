#!/bin/bash
# r05ag: ping-pong two-chain self-attention (af_attn2x_kernel, AF_ATTN_PP=1) against the shipped kernel: bit identity, tests, timing
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python - > gpurun_out/r05ag_attn_pp.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from adaface_dev_amd import ops
sys.path.insert(0, 'tools')
from bench_kernel import timeit
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).half().to(dev)
for (B, N, L, H, d) in [(8, 4096, 4096, 8, 40), (4, 4096, 4096, 8, 40), (2, 4096, 4096, 8, 40), (1, 4096, 4096, 8, 40), (8, 1024, 1024, 8, 40), (3, 1000, 1024, 5, 40), (2, 640, 192, 3, 40), (1, 512, 64, 2, 40)]:
    C = H * d
    q, k, v = rnd(B * N, C), rnd(B * L, C), rnd(B * L, C)
    vt = ops.transpose_tokens(v, B, L, C, C)
    outs = {}
    for pp in (0, 1):
        os.environ['AF_ATTN_PP'] = str(pp)
        outs[pp] = ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=H, d=d, ldq=C, ldk=C).clone()
    same = torch.equal(outs[0], outs[1])
    mx = (outs[0].float() - outs[1].float()).abs().max().item()
    rep_ok = all(torch.equal(ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=H, d=d, ldq=C, ldk=C), outs[1]) for _ in range(10))
    line = f"B{B} N{N} L{L} H{H} d{d}: bit-identical {same} (max abs diff {mx:.3e}), repeatable {rep_ok}"
    fl = 4.0 * B * H * N * L * d
    for rep in range(2):
        for pp in (0, 1):
            os.environ['AF_ATTN_PP'] = str(pp)
            ms = timeit(lambda: ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=H, d=d, ldq=C, ldk=C), 20)
            line += f" | pp{pp} {ms * 1e3:.1f} us {fl / ms / 1e9:.0f} TF/s"
    print(line, flush=True)
PY
AF_ATTN_PP=1 timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "attention" -x 2>&1 | tail -3 >> gpurun_out/r05ag_attn_pp.txt
cat gpurun_out/r05ag_attn_pp.txt
