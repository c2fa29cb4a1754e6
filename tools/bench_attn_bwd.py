#!/usr/bin/env python
"""Attention backward (af_attention_bwd: three transposes + delta + dQ kernel + dK/dV kernel) timed per kernel with rocprofv3-free
HIP events on the shapes of the U-Net: self-attention of the 64x64 / 32x32 / 16x16 levels and the 97-key cross-attention.
    python tools/bench_attn_bwd.py [batch] [substring of the shape's name: only that one]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from adaface_dev_amd import ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, N, L, heads, d in (("self 64x64", 4096, 4096, 8, 40), ("self 32x32", 1024, 1024, 8, 80), ("self 16x16", 256, 256, 8, 160),
                                 ("cross 64x64", 4096, 97, 8, 40), ("cross 32x32", 1024, 97, 8, 80)):
        if only not in name:
            continue
        C = heads * d
        q = torch.randn(B * N, C, generator=g).half().to(dev)
        k = torch.randn(B * L, C, generator=g).half().to(dev)
        v = torch.randn(B * L, C, generator=g).half().to(dev)
        do = torch.randn(B * N, C, generator=g).half().to(dev)
        vt = ops.transpose_tokens(v, B, L, C, C)
        o, lse = ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=heads, d=d, ldq=C, ldk=C, scale=d ** -0.5, want_lse=True)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        kw = dict(B=B, Nq=N, L=L, heads=heads, d=d, ldq=C, ldk=C, ldv=C, dq=dq, dk=dk, dv=dv, lddq=C, lddk=C, lddv=C)
        for _ in range(3):
            ops.attention_bwd(q, k, v, o, do, lse, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.attention_bwd(q, k, v, o, do, lse, **kw)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        fl = 2.0 * B * heads * N * L * d * 7                     # QK^T twice, dP twice, dV, dK, dQ (unpadded head dim)
        print(f"{name:12s} B{B} N{N} L{L} d{d}: {us:8.1f} us per backward (6 launches)   {fl / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
