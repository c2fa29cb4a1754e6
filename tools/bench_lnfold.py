#!/usr/bin/env python
"""Same-process A/B of the folded LayerNorm (ops.pack_matrix_ln) per transformer-block shape of one denoise step at U-Net batch 8:
[af_layernorm + the consuming GEMM at its tuned tile] against [the one GEMM with the LayerNorm folded in], for norm1 -> q|k|v,
norm2 -> to_q, norm3 -> GEGLU projection at the four levels.    python tools/bench_lnfold.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench_kernel import timeit  # noqa: E402


def main():
    from adaface_dev_amd import ops
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    tot_a = tot_b = 0.0
    for M, C, n_blocks in ((32768, 320, 5), (8192, 640, 5), (2048, 1280, 5), (512, 1280, 1)):
        x = rnd(M, C).half().to(dev)
        gam, bet = (rnd(C) * 0.2 + 1).to(dev), (rnd(C) * 0.2).to(dev)
        N_tok = M // 8
        for what, N in (("qkv", 3 * C), ("to_q", C), ("geglu", 8 * C)):
            w, b = rnd(N, C) * C ** -0.5, (rnd(N) * 0.1 if what == "geglu" else None)
            if what == "geglu":
                wi, bi = ops.interleave_geglu(w, b)
                pw = ops.pack_matrix(wi, bi, dev)
                wl = w * gam.cpu()[None, :]
                wli, bli = ops.interleave_geglu(wl, b + w @ bet.cpu())
                pl = ops.pack_matrix(wli, bli, dev)
                pl.ln_cs, pl.ln_eps = pl.wt.float().sum(dim=1).contiguous(), 1e-5
                kw = dict(act=ops.AF_ACT_GEGLU)
            else:
                pw = ops.pack_matrix(w, b, dev)
                pl = ops.pack_matrix_ln(w, b, gam, bet, 1e-5, dev)
                kw = dict(rows_per_batch=N_tok, split_col=2 * C) if what == "qkv" else {}
            t_ln = timeit(lambda: ops.layernorm(x, gam, bet, 1e-5), reps) * 1e3
            y = ops.layernorm(x, gam, bet, 1e-5)
            t_g = timeit(lambda: ops.gemm(y, pw, **kw), reps) * 1e3
            t_pair = timeit(lambda: ops.gemm(ops.layernorm(x, gam, bet, 1e-5), pw, **kw), reps) * 1e3
            t_f = timeit(lambda: ops.gemm(x, pl, **kw), reps) * 1e3
            tot_a += t_pair * n_blocks
            tot_b += t_f * n_blocks
            print(f"M{M:6d} C{C:5d} {what:6s}: layernorm {t_ln:6.1f} + gemm {t_g:6.1f} = pair {t_pair:6.1f} us | folded {t_f:6.1f} us  ({t_f - t_pair:+6.1f})")
    print(f"per denoise step: pairs {tot_a / 1e3:.3f} ms, folded {tot_b / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
