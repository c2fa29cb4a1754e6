"""Round 6: the one-launch cross-attention block at C = 640 (af_xattn640t_kernel) against the three-launch form, per layer, in isolation
(hipGraph of 20 calls each, alternating), at the denoise step's shape (U-Net batch 8, 32 x 32 tokens, 77 keys)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaface_dev_amd import ops
from adaface_dev_amd.ldm.modules import attention as A
from adaface_dev_amd.ldm.modules.diffusionmodules.util import LayerNorm

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for C, N in ((640, 1024), (320, 4096)):
    B, L, Cc = 8, 77, 768
    m = A.CrossAttention(C, Cc, heads=8, dim_head=C // 8).to(dev)
    ln = LayerNorm(C).to(dev)
    x = torch.randn(B * N, C, device=dev).half()
    ctx = torch.randn(B, L, Cc, device=dev).half()
    k, vt = ops.gemm(ctx.reshape(B * L, Cc), m._packed_kv(), rows_per_batch=L, split_col=C)
    m._kv_pre = (k, vt, C)
    A.XATTN640_FUSE_MIN_TOKENS = 64
    res = {}
    for rep in range(3):
        for fused in (False, True):
            A.FUSE_XATTN640 = fused
            A.FUSE_XATTN = fused
            t = timed(lambda: m.hip(x, B, N, context=ctx, residual=x, ln=ln))
            res.setdefault(fused, []).append(t)
    A.FUSE_XATTN640 = True; A.FUSE_XATTN = True
    o1 = m.hip(x, B, N, context=ctx, residual=x, ln=ln)
    A.FUSE_XATTN640 = False; A.FUSE_XATTN = False
    o3 = m.hip(x, B, N, context=ctx, residual=x, ln=ln)
    A.FUSE_XATTN = True
    err = ((o1.float() - o3.float()).norm() / o3.float().norm()).item()
    flop = 2 * B * N * C * C * 2 + 4 * B * N * L * C
    print(f"C={C} N={N}: three launches {['%.1f' % t for t in res[False]]} us | one launch {['%.1f' % t for t in res[True]]} us | rel diff {err:.2e} | "
          f"one launch = {flop / min(res[True]) / 1e6:.0f} TFLOP/s = {flop / min(res[True]) / 1e6 / 2500:.3f} of the matrix roof")
