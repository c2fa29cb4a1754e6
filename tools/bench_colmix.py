#!/usr/bin/env python
"""af_xattn_colmix (dv = P^T dO, dk = dS^T q of the captured cross-attention layers) per call at the training legs' shapes: hipGraph of 10 calls.
    python tools/bench_colmix.py [B heads N L d]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tools.bench_kernel import timeit  # noqa: E402


def main():
    from adaface_dev_amd import ops
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a)] if len(a) == 5 else [(4, 8, 4096, 97, 40), (2, 8, 4096, 97, 40), (1, 8, 4096, 97, 40), (4, 8, 4096, 77, 40)]
    dev = torch.device("cuda:0")
    for B, H, N, L, d in shapes:
        w = torch.rand(B, H, N, L, device=dev).softmax(-1).contiguous()
        x = torch.randn(B * N, H * d, device=dev).half()
        ms = timeit(lambda: ops.xattn_colmix(w, x, 1.0, B=B, Nq=N, L=L, heads=H, d=d), 10)
        ref = torch.einsum("bhnl,bnhc->blhc", w.double(), x.double().reshape(B, N, H, d)).reshape(B * L, H * d)
        got = ops.xattn_colmix(w, x, 1.0, B=B, Nq=N, L=L, heads=H, d=d).double()
        err = float((got - ref).norm() / ref.norm())
        fl = 2.0 * B * H * N * L * d
        print(f"colmix B{B} H{H} N{N} L{L} d{d}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.2f} TFLOP/s  ({w.numel() * 4 / ms / 1e6:.0f} GB/s of w)  rel-L2 vs fp64 {err:.2e}")
        # the other explicit-attention kernels of the captured layers (round 5: MFMA forms; AF_XATTN_EXPLICIT_MFMA=0 = the wave-per-row kernels)
        kw = dict(B=B, Nq=N, L=L, heads=H, d=d)
        q, k = x, torch.randn(B * L, H * d, device=dev).half()
        score = ops.xattn_scores(q, k, scale=d ** -0.5, **kw)
        prob, o = ops.xattn_softmax_pv(score, k, **kw)
        t = {"scores": timeit(lambda: ops.xattn_scores(q, k, scale=d ** -0.5, **kw), 10), "softmax_pv": timeit(lambda: ops.xattn_softmax_pv(score, k, **kw), 10),
             "softmax_pv_bwd": timeit(lambda: ops.xattn_softmax_pv_bwd(prob, k, q, None, **kw), 10), "rowmix": timeit(lambda: ops.xattn_rowmix(prob, k, 1.0, **kw), 10)}
        print("   " + "  ".join(f"{n} {v * 1e3:.1f} us" for n, v in t.items()))


if __name__ == "__main__":
    main()
