"""CPU oracle for the attention-score rewrites and the gradient scaler of the live path.
TEST INFRASTRUCTURE ONLY: imported by tests/ alone.

PINNED by tests/golden/sdpa.npz, written by the REFERENCE's own ``scaled_dot_product_attention`` / ``ScaleGrad`` /
``gen_gradient_scaler`` (adaface/diffusers_attn_lora_capture.py:23-77, 79-139; imported with empty stand-ins for diffusers / peft,
tests/golden/ref_import.py): outputs, scores, probabilities and the gradients w.r.t. q, k, v and the learnable scale factor."""
import math

import torch


class ScaleGrad(torch.autograd.Function):
    """Identity forward, gradient x alpha (diffusers_attn_lora_capture.py:23-42)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = float(alpha)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.alpha, None


def gradient_scaler(alpha):
    """gen_gradient_scaler (:62-70): 1 -> identity, 0 -> detach, else ScaleGrad."""
    if alpha == 1:
        return lambda x: x
    if alpha == 0:
        return torch.detach
    return lambda x: ScaleGrad.apply(x, alpha)


def scaled_dot_product_attention(query, key, value, cross_attn_scale_factor, attn_mask=None, subj_indices=None,
                                 normalize_cross_attn=False, mix_attn_mats_in_batch=False, scale=None):
    """The explicit attention of the LoRA / capture processor (:79-139), dropout 0.
    query [B,H,L,d], key / value [B,H,S,d]; attn_mask [B,1,L,S] bool (False -> -inf) or additive.
    mix_attn_mats_in_batch (:108-118): the batch is [SC..., MC...]; scores <- (SC + MC.detach()) / 2 for both halves.
    normalize_cross_attn (:119-133): for the subject tokens (b, n) in ``subj_indices`` the score column [b, :, :, n] is centred over
    the L pixels (mean detached) and multiplied by the learnable ``cross_attn_scale_factor`` seen through a x10 gradient scaler.
    Returns (output, scores, probabilities)."""
    B, L, S = query.size(0), query.size(-2), key.size(-2)
    sf = 1 / math.sqrt(query.size(-1)) if scale is None else scale
    bias = torch.zeros(B, 1, L, S, dtype=query.dtype)
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            bias = bias.masked_fill(~attn_mask, float("-inf"))
        else:
            bias = bias + attn_mask
    score = query @ key.transpose(-2, -1) * sf + bias
    if mix_attn_mats_in_batch:
        sc, mc = score.chunk(2, dim=0)
        score = ((sc + mc.detach()) / 2).repeat(2, 1, 1, 1)
    elif normalize_cross_attn:
        b, n = subj_indices
        sub = score[b, :, :, n]                                        # [n_subj, H, L]
        sub = sub - sub.mean(dim=2, keepdim=True).detach()
        sub = sub * gradient_scaler(10)(cross_attn_scale_factor)
        score = score.clone()
        score[b, :, :, n] = sub
    prob = torch.softmax(score, dim=-1)
    return prob @ value, score, prob
