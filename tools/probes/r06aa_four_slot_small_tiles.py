"""Round 6: tiles 20 / 21 (64 x 128 / 128 x 64 on a four-slot ring) against tiles 16 / 17 / 2 and the table's choice on the small plain GEMMs of the training legs and of the
16 x 16 / 8 x 8 levels: warm operands, hipGraph of 20 launches.
NOT RUNNABLE ON THE TREE AS IT IS: tiles 20 / 21 were two more instantiations of the whole-line kernel (launch3w<1, 1, 4, 2, E3_STD, 4> / <1, 2, 2, 2, E3_STD, 4> behind
wide ids 17 / 18 in af_gemm3_try_launch, tile range 0 .. 21 in af_gemm); they measured slower than tiles 16 / 17 on every shape (profiles/r06aa_four_slot_small_tiles.txt) and
were taken out again."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaface_dev_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn, n=20, reps=3):
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


for (M, N, K) in ((256, 1280, 1280), (512, 1280, 1280), (1024, 640, 640), (1024, 1920, 640), (1024, 640, 2560), (2048, 640, 640), (2048, 1280, 1280), (2048, 1280, 5120), (4096, 320, 320),
                  (4096, 960, 320), (4096, 320, 1280), (8192, 640, 640), (8192, 640, 2560), (64, 1280, 1280), (388, 768, 768)):
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    pw = ops.pack_matrix((torch.randn(N, K, generator=g) * K ** -0.5).half(), torch.zeros(N), dev)
    row = []
    for tile, sp in ((0, 0), (2, 1), (16, 1), (17, 1), (20, 1), (21, 1), (20, 2), (21, 2), (16, 2)):
        if tile in (16, 20) and N % 128:
            continue
        try:
            t = timed(lambda: ops.gemm(a, pw, tile=tile, splits=sp))
        except Exception as e:
            t = float("nan")
        row.append(f"{'table' if tile == 0 else f'{tile}x{sp}'} {t:6.1f}")
    print(f"M{M} N{N} K{K}: " + " | ".join(row), flush=True)
