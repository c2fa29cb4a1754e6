"""Timing half of tools/probes/r06ab_gn_in_conv_probe.sh: halo-resident 3x3 convolutions (tile 14) of the denoise step's 64 x 64 / 32 x 32 levels and the GroupNorm + SiLU
launches in front of them, warm, hipGraph of 20 launches each.  Run once per library build (AF_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaface_dev_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "?"


def timed(fn, n=20, reps=4):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for (B, H, W, ci, co, sp) in ((8, 64, 64, 320, 320, 1), (8, 64, 64, 640, 320, 1), (8, 64, 64, 960, 320, 1), (8, 32, 32, 640, 640, 2), (8, 32, 32, 1280, 640, 2), (8, 32, 32, 320, 640, 1)):
    x = (torch.randn(B, H, W, ci, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) * (9 * ci) ** -0.5).half()
    pw = ops.pack_conv3x3(w, torch.randn(co, generator=g).abs() + 0.5, dev)
    tc = timed(lambda: ops.conv3x3(x, pw, tile=14, splits=sp))
    gm, bt = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
    xf = x.reshape(B, H * W, ci)
    tg = timed(lambda: ops.groupnorm(xf, gm, bt, 1e-5, True))
    print(f"[{tag}] conv B{B} {H}x{W} {ci}->{co} splits{sp}: {tc:7.1f} us | GroupNorm + SiLU of its input [{B}, {H * W}, {ci}]: {tg:6.1f} us", flush=True)
