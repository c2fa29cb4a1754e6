"""torch.autograd nodes of the TRAINABLE part of the hot path: the CLIP text transformer inside SubjBasisGenerator
(``prompt2token_proj``, the only weights the Arc2Face-distillation stage updates besides the 3 layer-mix weights;
reference ddpm.py:4120-4170 collects them, subj_basis_generator.py:841-853 freezes the embeddings).

Every node's forward and backward is one or a few launches of this package's gfx950 kernels through the C ABI:

  LinearFn     y = x W^T + b (+ residual)     fwd af_gemm; dx af_gemm on the transposed pack; dW = dy^T x as af_gemm
                                              over token-transposed operands (af_transpose_tokens); db af_colsum
  LayerNormFn  af_layernorm / af_layernorm_bwd; dgamma = colsum(dy * x_hat), dbeta = colsum(dy)
  QuickGeluFn  af_quickgelu_fwd / af_quickgelu_bwd
  AttentionFn  causal (multi-key-per-token) flash attention forward with log-sum-exp / af_attention_bwd

Activations and activation gradients are fp16, so callers scale the loss (ldm/trainer.py LossScaler) exactly like
fp16 AMP does in the reference (`precision: 16` Lightning runs); parameter gradients are returned in the parameter's
dtype (fp32 master weights) still multiplied by that scale and are unscaled in the flat gradient arena."""
import torch

from . import _lib, ops
from .ops import F16, round_up


def wgrad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW [N, K] (fp32) = dy^T [N, M] @ x [M, K] through af_gemm: both operands are token-transposed so the reduction (M) is the
    K-contiguous axis the GEMM kernels want; pads are zero-filled.  The result is the fp32 accumulator (AF_OUT_F32): a sum over
    thousands of tokens of |dy| ~ 40, |x| ~ 20 passes 65504 -- seen on the 1x1 shortcut adapters -- and an fp16 cast there would
    make every optimizer step of the run an overflow-skip, because the U-Net backward normalises d(eps) itself (openaimodel.py)
    and the loss scaler cannot shrink it."""
    M, N = dy.shape
    K = x.shape[1]
    m64 = round_up(M, 64)
    dev = dy.device
    dyt = torch.empty((N, m64), dtype=F16, device=dev)                         # the transpose zero-fills columns M..m64
    k128 = round_up(K, 128)
    xt = torch.empty((k128, m64), dtype=F16, device=dev)
    if k128 != K:
        xt[K:].zero_()                                                         # only the pad rows (the packed-weight layout wants 128-row multiples)
    L = _lib.lib()
    _lib.check(L.af_transpose_tokens_pair(ops._p(dy), ops._p(dyt), N, N, ops._p(x), ops._p(xt), K, K, 1, M, m64, ops._stream()),
               "af_transpose_tokens_pair")                                     # both operands in one launch
    pw = ops.PackedWeight(xt, None, K, m64, m64, 1, m64)
    return ops.gemm(dyt, pw, out_f32=True)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, mod):
        ctx.save_for_backward(x)
        ctx.mod = mod
        ctx.has_res = residual is not None
        return ops.gemm(x, mod.packed(), residual=residual)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        mod = ctx.mod
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dx = ops.gemm(dy, mod.packed_bwd()) if need[0] else None
        dw = wgrad(dy, x).to(mod.weight.dtype) if need[1] else None
        db = ops.colsum(dy).to(mod.bias.dtype) if need[2] and mod.bias is not None else None
        return dx, dw, db, (dy if ctx.has_res and need[3] else None), None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        g32, b32 = gamma.detach().float(), beta.detach().float()
        ctx.save_for_backward(x, g32)
        ctx.eps = eps
        ctx.pdtype = gamma.dtype
        return ops.layernorm(x, g32, b32, eps)

    @staticmethod
    def backward(ctx, dy):
        x, g32 = ctx.saved_tensors
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dx = ops.layernorm_bwd(x, g32, dy, ctx.eps) if need[0] else None
        dg = db = None
        if (need[1] or need[2]) and x.numel() // x.shape[-1] <= 2048 and x.shape[-1] <= 1536:
            dg, db = ops.layernorm_param_grads(x, dy, ctx.eps)               # row statistics + one column pass (af_layernorm_param_grads)
            dg = dg.to(ctx.pdtype) if need[1] else None
            db = db.to(ctx.pdtype) if need[2] else None
        else:
            if need[1]:
                xhat = ops.layernorm(x, torch.ones_like(g32), torch.zeros_like(g32), ctx.eps)
                dg = ops.colsum(dy, xhat).to(ctx.pdtype)
            if need[2]:
                db = ops.colsum(dy).to(ctx.pdtype)
        return dx, dg, db, None


class QuickGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.quickgelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return ops.quickgelu_bwd(x, dy.contiguous())


class AttentionFn(torch.autograd.Function):
    """q [B*T, E], k / v [B*T*m, E] (token-major, m keys per token) -> [B*T, E]."""

    @staticmethod
    def forward(ctx, q, k, v, B, T, m, heads, scale, causal):
        E = q.shape[1]
        d = E // heads
        vt = ops.transpose_tokens(v, B, T * m, E, E)
        o, lse = ops.attention(q, k, vt, B=B, Nq=T, L=T * m, heads=heads, d=d, ldq=E, ldk=E, scale=scale, want_lse=True,
                               causal_m=m if causal else 0)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cfg = (B, T, m, heads, d, scale, causal)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, T, m, heads, d, scale, causal = ctx.cfg
        E = heads * d
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attention_bwd(q, k, v, o, do.contiguous(), lse, B=B, Nq=T, L=T * m, heads=heads, d=d, ldq=E, ldk=E, ldv=E,
                          dq=dq, dk=dk, dv=dv, lddq=E, lddk=E, lddv=E, scale=scale, causal_m=m if causal else 0)
        return dq, dk, dv, None, None, None, None, None, None


class QKVLinearFn(torch.autograd.Function):
    """q | k | v projections of one attention layer as ONE GEMM on the concatenated weights ([3E, E]; multiplier-1 layers): forward one
    launch instead of three; backward one input-gradient GEMM (instead of three and two adds), one weight-gradient GEMM over ONE pair
    of transposed operands, one column sum.  Returns qkv [M, 3E]; the gradients are handed back per projection as views."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, bq, bk, bv, attn):
        ctx.save_for_backward(x)
        ctx.attn = attn
        return ops.gemm(x, attn.qkv_packed())

    @staticmethod
    def backward(ctx, dqkv):
        x, = ctx.saved_tensors
        attn = ctx.attn
        dqkv = dqkv.contiguous()
        need = ctx.needs_input_grad
        E = attn.embed_dim
        dx = ops.gemm(dqkv, attn.qkv_packed_bwd()) if need[0] else None
        dws = [None, None, None]
        if any(need[1:4]):
            dw = wgrad(dqkv, x)                                   # [3E, E] fp32
            dws = [dw[i * E:(i + 1) * E] if need[1 + i] else None for i in range(3)]
        dbs = [None, None, None]
        if any(need[4:7]):
            db = ops.colsum(dqkv)
            dbs = [db[i * E:(i + 1) * E] if need[4 + i] else None for i in range(3)]
        return (dx, *dws, *dbs, None)


class AttentionQKVFn(torch.autograd.Function):
    """Causal flash attention on a packed qkv [B*T, 3E] (q | k | v column blocks, row stride 3E) -> [B*T, E]; the backward writes
    dq | dk | dv straight into the column blocks of one [B*T, 3E] gradient."""

    @staticmethod
    def forward(ctx, qkv, B, T, heads, scale, causal):
        E = qkv.shape[1] // 3
        d = E // heads
        q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
        vt = ops.transpose_tokens(v, B, T, E, 3 * E)
        o, lse = ops.attention(q, k, vt, B=B, Nq=T, L=T, heads=heads, d=d, ldq=3 * E, ldk=3 * E, scale=scale, want_lse=True,
                               causal_m=1 if causal else 0)
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (B, T, heads, d, scale, causal)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, T, heads, d, scale, causal = ctx.cfg
        E = heads * d
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], o, do.contiguous(), lse, B=B, Nq=T, L=T, heads=heads, d=d,
                          ldq=3 * E, ldk=3 * E, ldv=3 * E, dq=dqkv[:, :E], dk=dqkv[:, E:2 * E], dv=dqkv[:, 2 * E:], lddq=3 * E, lddk=3 * E,
                          lddv=3 * E, scale=scale, causal_m=1 if causal else 0)
        return dqkv, None, None, None, None, None


def linear(mod, x, residual=None):
    return LinearFn.apply(x, mod.weight, mod.bias, residual, mod)


def layer_norm(mod, x):
    return LayerNormFn.apply(x, mod.weight, mod.bias, mod.eps)


class MatmulNTFn(torch.autograd.Function):
    """C [M, N] (fp32) = A [M, K] @ Bt [N, K]^T on the MFMA GEMM kernel (fp16 operands, fp32 accumulation and output, AF_OUT_F32), with
    both input gradients: dA = dC @ Bt (af_gemm on the transposed pack), dBt = dC^T @ A (``wgrad``).  The incoming dC is a loss
    gradient of arbitrary size, so it is normalised by a power of two on the device before its fp16 cast (largest entry ~ 256) and the
    results are unscaled in fp32, as the U-Net's backward node does.  The Stage-2 feature-matching losses' [4096 x C] x [C x 4096]
    products (reference ldm/util.py:2336-2344, 2269-2281) run through this on the GPU."""

    @staticmethod
    def forward(ctx, a, bt):
        a16, bt16 = a.detach().to(F16).contiguous(), bt.detach().to(F16).contiguous()
        ctx.save_for_backward(a16, bt16)
        ctx.dtypes = (a.dtype, bt.dtype)
        return ops.gemm(a16, ops.pack_matrix(bt16, None, a.device), out_f32=True)

    @staticmethod
    def backward(ctx, dc):
        a16, bt16 = ctx.saved_tensors
        need = ctx.needs_input_grad
        amax = dc.abs().amax().float().clamp_min(1e-30)
        scale = torch.exp2(torch.floor(torch.log2(256.0 / amax)))
        d16 = (dc.float() * scale).to(F16).contiguous()
        da = dbt = None
        if need[0]:
            da = (ops.gemm(d16, ops.pack_matrix(bt16.t(), None, dc.device), out_f32=True) / scale).to(ctx.dtypes[0])
        if need[1]:
            dbt = (wgrad(d16, a16) / scale).to(ctx.dtypes[1])
        return da, dbt


def matmul_nt(a: torch.Tensor, bt: torch.Tensor) -> torch.Tensor:
    """a [M, K] @ bt[N, K]^T -> fp32 [M, N]: the MFMA kernel for device tensors whose sizes it takes (K % 8 == 0, N % 4 == 0), plain torch
    otherwise (host tensors: the CPU test-suite of the loss functions)."""
    if a.is_cuda and a.shape[1] % 8 == 0 and bt.shape[0] % 4 == 0:
        return MatmulNTFn.apply(a, bt)
    return a.float() @ bt.float().t()


def low_rank_product(b: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """B [Cout, r] @ A [r, Cin] -> fp32: the LoRA / DoRA weight delta that is re-evaluated every step (DoRA's weight norm ||W + s B A||, the
    merged weights of the no-grad passes).  On the device it is the MFMA GEMM of this package (fp16 operands, fp32 accumulation: the
    product is a small correction to W, which is itself packed to fp16 afterwards) instead of a vendor-BLAS sgemm under torch's ``@``;
    host tensors (CPU tests of the adapter algebra, offline merging) use torch."""
    if b.is_cuda and a.shape[0] % 8 == 0 and a.shape[1] % 4 == 0:
        return matmul_nt(b.detach(), a.detach().t().contiguous())
    return b.float() @ a.float()
