"""GroupNorm(+SiLU) at the denoise step's shapes (hipGraph of 20 calls each, cold-ish: a 256 MB flush between shapes is
not done -- the producer's output normally still sits in the Infinity Cache, as here).   python tools/bench_gn.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from adaface_dev_amd import ops
    dev = torch.device("cuda:0")
    shapes = [(8, 4096, 320, 0), (8, 4096, 320, 320), (8, 4096, 640, 320), (8, 1024, 640, 0), (8, 1024, 640, 320), (8, 1024, 1280, 0),
              (8, 1024, 640, 640), (4, 4096, 320, 0), (4, 4096, 640, 320), (8, 256, 1280, 0)]
    for (B, HW, c1, c2) in shapes:
        C = c1 + c2
        x1 = torch.randn(B, HW, c1, device=dev).half()
        x2 = torch.randn(B, HW, c2, device=dev).half() if c2 else None
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        for _ in range(3):
            ops.groupnorm(x1, g, b, 1e-5, True, x2=x2)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                y = ops.groupnorm(x1, g, b, 1e-5, True, x2=x2)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        mb = 2 * B * HW * C * 2 / 1e6
        print(f"  [{B}, {HW}, {c1}+{c2}]  {ts[3]:7.1f} us   {mb:6.1f} MB  {mb / ts[3] * 1e-3:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run()
    else:
        for tag, env in (("default", {}), ("no single-launch small / pair forms (AF_GN_NO_SMALL=1)", {"AF_GN_NO_SMALL": "1"})):
            print(tag, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), check=True)
