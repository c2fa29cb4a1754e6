#!/usr/bin/env python
"""Micro-benchmark of one kernel shape on the GPU box (also the target of rocprofv3 --pmc runs).
    python tools/bench_kernel.py attn  B N L heads d [reps]
    python tools/bench_kernel.py gemm  M N K [tile] [splits] [reps] [act]      (act 2 = GEGLU, 3 = quick-GELU, 1 = SiLU)
    python tools/bench_kernel.py conv  B H W Cin Cout [tile] [splits] [reps] [upsample]
    python tools/bench_kernel.py gn    B HW C [reps]
    python tools/bench_kernel.py blas  M N K [reps]      (hipBLASLt via torch.matmul: reference point only)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, reps):
    """Average device time of fn(): `reps` calls captured into ONE hipGraph and replayed (so the host's launch rate -- ~10 us per
    Python/ctypes call -- does not floor the measurement of short kernels)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("AF_BENCH_COLD"):          # single launches after evicting L2 / Infinity Cache (median of 7): cold operands
        flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        ts = []
        for _ in range(7):
            flush.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[3]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    from adaface_dev_amd import ops
    kind = sys.argv[1]
    a = [int(v) for v in sys.argv[2:]]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).half().to(dev)
    if kind == "attn":
        B, N, L, H, d = a[:5]
        reps = a[5] if len(a) > 5 else 20
        C = H * d
        q, k, v = rnd(B * N, C), rnd(B * L, C), rnd(B * L, C)
        vt = ops.transpose_tokens(v, B, L, C, C)
        ms = timeit(lambda: ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=H, d=d, ldq=C, ldk=C), reps)
        fl = 4.0 * B * H * N * L * d
        print(f"attn B{B} N{N} L{L} H{H} d{d}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s")
    elif kind == "add":           # calibration: a trivial element-wise kernel on n halves (per-node floor of the timing harness)
        n = a[0]
        x, y = rnd(n), rnd(n)
        ms = timeit(lambda: ops.add(x, y), a[1] if len(a) > 1 else 50)
        print(f"add n{n}: {ms * 1e3:.2f} us")
    elif kind == "gemm":
        M, N, K = a[:3]
        tile, splits = (a[3] if len(a) > 3 else 0), (a[4] if len(a) > 4 else 0)
        reps = a[5] if len(a) > 5 else 20
        act = a[6] if len(a) > 6 else 0
        x, w = rnd(M, K), rnd(N, K)
        pw = ops.pack_matrix(w, torch.zeros(N), dev)
        ms = timeit(lambda: ops.gemm(x, pw, tile=tile, splits=splits, act=act), reps)
        print(f"gemm M{M} N{N} K{K} tile{tile} splits{splits} act{act}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
    elif kind == "blas":          # the vendor library (hipBLASLt through torch.matmul) on the same shape: a reference point, not a product path
        M, N, K = a[:3]
        reps = a[3] if len(a) > 3 else 20
        x, w = rnd(M, K), rnd(N, K)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ms = timeit(lambda: torch.matmul(x, w.t(), out=out), reps)
        print(f"blas M{M} N{N} K{K}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
    elif kind == "conv":
        B, H, W, ci, co = a[:5]
        tile, splits = (a[5] if len(a) > 5 else 0), (a[6] if len(a) > 6 else 0)
        reps = a[7] if len(a) > 7 else 20
        ups = bool(a[8]) if len(a) > 8 else False          # nearest x2 folded into the gather (H, W are the INPUT size)
        x, w = rnd(B, H, W, ci), rnd(co, ci, 3, 3)
        pw = ops.pack_conv3x3(w, None, dev)
        ms = timeit(lambda: ops.conv3x3(x, pw, tile=tile, splits=splits, upsample=ups), reps)
        f = 4 if ups else 1
        print(f"conv B{B} {H}x{W} {ci}->{co} tile{tile} splits{splits} ups{int(ups)}: {ms * 1e3:.1f} us  {2.0 * f * B * H * W * co * 9 * ci / ms / 1e9:.1f} TFLOP/s")
    elif kind == "gn":
        B, HW, C = a[:3]
        reps = a[3] if len(a) > 3 else 20
        x = rnd(B, HW, C)
        gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        ms = timeit(lambda: ops.groupnorm(x, gm, bt, 1e-5, True), reps)
        print(f"gn B{B} HW{HW} C{C}: {ms * 1e3:.1f} us  {3.0 * B * HW * C * 2 / ms / 1e6:.1f} GB/s (2R+1W)")


if __name__ == "__main__":
    main()
