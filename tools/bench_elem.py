"""Element-wise training kernels per call (hipGraph of 20 calls): GEGLU forward / backward, quick-GELU, add -- at the Stage-1 micro-batch's shapes.
   python tools/bench_elem.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaface_dev_amd import ops

dev = torch.device("cuda:0")


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(ts)[3]


for (M, inner) in [(16384, 1280), (4096, 2560), (1024, 5120), (49152, 1280)]:
    hp = torch.randn(M, 2 * inner, device=dev).half()
    do = torch.randn(M, inner, device=dev).half()
    print(f"  geglu [{M}, 2x{inner}]  fwd {timed(lambda: ops.geglu_fwd(hp)):6.1f} us   bwd {timed(lambda: ops.geglu_bwd(hp, do)):6.1f} us", flush=True)
x = torch.randn(388, 3072, device=dev).half()
print(f"  quick-GELU [388, 3072]  fwd {timed(lambda: ops.quickgelu_fwd(x)):6.1f} us   bwd {timed(lambda: ops.quickgelu_bwd(x, x)):6.1f} us")
