"""Round 6: the guided denoise step's U-Net pass (cond + uncond = batch 8) as TWO CONCURRENT batch-4 passes on two streams of one hipGraph, against the
batch-8 pass and against the two batch-4 passes back to back on one stream (what per-launch efficiency costs at half the rows, and what concurrency
gives back).  python tools/probes/r06w_two_streams.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaface_dev_amd import SD15_UNET_CONFIG, ops, rng
from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with rng.skip_default_init():
    unet = UNetModel(**SD15_UNET_CONFIG)
unet = unet.to(dev).eval()
rng.load_synth_weights(unet, seed=0, on_device=True)
unet.prepare()
for p in unet.parameters():
    p.requires_grad_(False)
x = rng.synth_input("bench.x", (8, 4, 64, 64), seed=1).to(dev)
ctx = rng.synth_input("bench.ctx", (8, 77, 768), seed=1).to(dev).half()
t = torch.full((8,), 500, device=dev, dtype=torch.int64)
xa, xb, ca, cb, ta, tb = x[:4].contiguous(), x[4:].contiguous(), ctx[:4].contiguous(), ctx[4:].contiguous(), t[:4].contiguous(), t[4:].contiguous()
s2 = torch.cuda.Stream()


def whole():
    return unet(x, t, ctx, extra_info=None)


def halves_sequential():
    return unet(xa, ta, ca, extra_info=None), unet(xb, tb, cb, extra_info=None)


def halves_concurrent():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    a = unet(xa, ta, ca, extra_info=None)
    with torch.cuda.stream(s2):
        prev = ops.set_workspace_lane(1)
        try:
            b = unet(xb, tb, cb, extra_info=None)
        finally:
            ops.set_workspace_lane(prev)
    cur.wait_stream(s2)
    return a, b


def capture(fn):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
            with torch.cuda.graph(g, stream=s):
                out = fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
    return g, out


def timed(g, n=30):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gw, ow = capture(whole)
gs, os_ = capture(halves_sequential)
gc, oc = capture(halves_concurrent)
gw.replay(); gs.replay(); gc.replay(); torch.cuda.synchronize()
ref = ow.float()
for name, o in (("sequential halves", os_), ("concurrent halves", oc)):
    got = torch.cat([o[0], o[1]]).float()
    print(f"{name}: rel diff to the batch-8 pass {((got - ref).norm() / ref.norm()).item():.2e}")
for r in range(reps):
    print(f"rep {r}: batch 8 {timed(gw):.3f} ms | two batch-4 passes back to back {timed(gs):.3f} ms | two batch-4 passes on two streams {timed(gc):.3f} ms")
