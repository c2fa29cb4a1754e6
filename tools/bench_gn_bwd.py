"""GroupNorm(+SiLU) backward (af_groupnorm_bwd: partial sums + apply) per call at the training batch's shapes, hipGraph of 20 calls.
   python tools/bench_gn_bwd.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaface_dev_amd import ops

dev = torch.device("cuda:0")
for (B, HW, c1, c2) in [(4, 4096, 320, 0), (4, 4096, 640, 320), (4, 1024, 640, 0), (4, 1024, 1280, 0), (4, 256, 1280, 0), (12, 4096, 320, 0), (8, 4096, 320, 0)]:
    C = c1 + c2
    x1 = torch.randn(B, HW, c1, device=dev).half()
    x2 = torch.randn(B, HW, c2, device=dev).half() if c2 else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.randn(B, HW, C, device=dev).half()
    y, stats = ops.groupnorm_train(x1, g, b, 1e-5, True, x2=x2)
    for _ in range(3):
        ops.groupnorm_bwd(x1, g, b, stats, dy, True, x2=x2)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            out = ops.groupnorm_bwd(x1, g, b, stats, dy, True, x2=x2)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts.sort()
    print(f"  [{B}, {HW}, {c1}+{c2}]  {ts[3]:7.1f} us per backward call", flush=True)
