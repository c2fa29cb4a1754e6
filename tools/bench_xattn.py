"""The C = 320 cross-attention block (norm2 folded q projection + 77-key core + to_out + residual) per layer: one launch (af_xattn_fused)
against the three-launch path, hipGraphs of 10 calls replayed interleaved.   python tools/bench_xattn.py [U-Net batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaface_dev_amd import ops
from adaface_dev_amd.ldm.modules import attention as A
from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm

dev = torch.device("cuda:0")
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [8, 4, 2]):
    N, L, C = 4096, 77, 320
    m = A.CrossAttention(C, 768, heads=8, dim_head=40).to(dev)
    ln = LayerNorm(C).to(dev)
    x = torch.randn(B * N, C, device=dev).half()
    ctx = torch.randn(B, L, 768, device=dev).half()
    k, vt = ops.gemm(ctx.reshape(B * L, 768), m._packed_kv(), rows_per_batch=L, split_col=C)
    m._kv_pre = (k, vt, C)

    def graph(fused):
        A.FUSE_XATTN = fused
        A.XATTN_FUSE_MIN_TOKENS = 0          # the comparison itself decides from which batch on the one-launch form pays
        for _ in range(2):
            m.hip(x, B, N, context=ctx, residual=x, ln=ln)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                y = m.hip(x, B, N, context=ctx, residual=x, ln=ln)
        return g
    gs = {"three launches": graph(False), "one launch": graph(True)}
    ts = {n: [] for n in gs}
    for _ in range(9):
        for n, g in gs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts[n].append(e0.elapsed_time(e1) * 100.0)
    flop = B * N * (2 * 2 * C * C + 2 * 2 * L * C)
    print(f"  U-Net batch {B}: " + " | ".join(f"{n} {sorted(v)[4]:6.1f} us ({flop / sorted(v)[4] * 1e-6:4.0f} TFLOP/s)" for n, v in ts.items()), flush=True)
