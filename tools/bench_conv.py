"""3x3 convolution launch forms at the U-Net's shapes: the tuned tap-by-tap tile vs the halo-resident kernel (tile 14), hipGraph of 10
calls over 10 different weight tensors (so the weights are not L2-hot), median of 7 replays.   python tools/bench_conv.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AF_GEMM3_ABLATE_DYNAMIC"] = "1"        # AF_GEMM3_ABLATE is re-read per launch: in-process A/B of kernel variants
import torch

from adaface_dev_amd import ops


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    shapes = [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 320, 640), (32, 640, 640), (32, 1280, 640), (32, 960, 640),
              (16, 640, 1280), (16, 1280, 1280), (16, 2560, 1280), (16, 1920, 1280)]
    print(f"# batch {B}: us per launch (TFLOP/s), interleaved replays")
    for (HW, cin, cout) in shapes:
        x = torch.randn(B, HW, HW, cin, device=dev).half()
        packs = [ops.pack_conv3x3(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.zeros(cout), dev) for _ in range(10)]
        flops = 2.0 * B * HW * HW * cout * 9 * cin

        def graph_of(env_bits=0, **kw):
            if env_bits:
                os.environ["AF_GEMM3_ABLATE"] = str(env_bits)
            for pw in packs[:2]:
                ops.conv3x3(x, pw, **kw)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for pw in packs:
                    ops.conv3x3(x, pw, **kw)
            os.environ.pop("AF_GEMM3_ABLATE", None)
            return g
        key = f"9,{B * HW * HW},{cout},{9 * cin},0,0,1,0"
        tuned = tuple(ops.tune_table().get(key, (0, 1)))
        variants = [(f"tuned{tuned}", graph_of())]
        for sp in ({64: 1, 32: 2, 16: 4}[HW],):
            if sp <= cin // 64 and B * HW * HW // 256 * (cout // 160) * sp <= 1024:
                variants.append((f"t14 s{sp}", graph_of(tile=14, splits=sp)))
                variants.append((f"t14 s{sp} early-dma", graph_of(1024, tile=14, splits=sp)))
        times = {n: [] for n, _ in variants}
        for _ in range(9):                      # interleaved rounds: every variant sees the same clocks / cache state drift
            for n, g in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) * 100.0)
        line = f"  {HW}x{HW} {cin:4d}->{cout:4d} "
        for n, _ in variants:
            t = sorted(times[n])[4]
            line += f" | {n} {t:6.1f} ({flops / t * 1e-6:4.0f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
