"""Round 6: what the VAE decoder's 3x3 convolutions run at (table tile, per image and 4 images), hipGraph of 5 calls each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaface_dev_amd import ops

dev = torch.device("cuda:0")


def timed(fn, n=5, reps=3):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
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


tot = {1: 0.0, 4: 0.0}
tot14 = {1: 0.0, 4: 0.0}
for (hw, cin, cout, up, count) in ((64, 512, 512, False, 11), (64, 512, 512, True, 1), (128, 512, 512, False, 6), (128, 512, 512, True, 1), (256, 512, 256, False, 1),
                                   (256, 256, 256, False, 5), (256, 256, 256, True, 1), (512, 256, 128, False, 1), (512, 128, 128, False, 5)):
    w = torch.randn(cout, cin, 3, 3) * (cin * 9) ** -0.5
    pw = ops.pack_conv3x3(w.half(), torch.zeros(cout), dev)
    for B in (1, 4):
        x = torch.randn(B, hw, hw, cin, device=dev).half()
        t = timed(lambda: ops.conv3x3(x, pw, upsample=up))
        t14 = timed(lambda: ops.conv3x3(x, pw, upsample=up, tile=14, splits=1))
        ho = hw * (2 if up else 1)
        fl = 2.0 * B * ho * ho * cout * cin * 9
        tot[B] += t * count
        tot14[B] += min(t, t14) * count
        print(f"{hw}x{hw} {cin}->{cout}{' up' if up else ''} B={B}: table {t:8.1f} us {fl / t / 1e6:6.0f} TFLOP/s | halo-resident (tile 14) {t14:8.1f} us {fl / t14 / 1e6:6.0f} TFLOP/s"
              f"  ({fl/1e9:.0f} GFLOP) x{count} per decode")
        del x
print("3x3 convolutions of one decode (sum), table:", {b: f"{v / 1e3 / b:.2f} ms per image" for b, v in tot.items()})
print("... with the faster of the two per shape:", {b: f"{v / 1e3 / b:.2f} ms per image" for b, v in tot14.items()})
