"""Losses of the Stage-2 (compositional distillation) iterations that are taken on the captured cross-attention activations
(SURVEY.md 8f rank 4).  Reference: ``ldm/util.py`` -- ``calc_sc_rep_attn_distill_loss`` :2047-2121,
``calc_subj_attn_cross_t_diff_loss`` :2123-2147, ``calc_attn_norm_loss`` :1756-1818, ``calc_subj_masked_bg_suppress_loss`` :1822-1918,
``calc_dyn_loss_scale`` :1485-1518 and their helpers (``masked_mean`` :1194, ``masked_l2_loss`` :1215, ``sel_emb_attns_by_indices``
:1398, ``resize_mask_to_target_size`` :1333, ``extend_indices_B_by_n_times`` :1060); assembled by
``LatentDiffusion.calc_comp_feat_distill_loss`` (ddpm.py:3190-3602).

Inputs are what the capture pass (modules/diffusionmodules/capture_graph.py) returns: ``ca_layers_activations[key][layer]`` with
``attn`` [4B, heads, N, L] (the batch is four blocks: subject-single, subject-comp, subject-comp-rep, class-comp), ``k`` / ``v``
[4B, C, L], carrying gradients where the pass ran with them.  These are small reductions over a few MB of captured tensors (the
compute of a Stage-2 iteration is its U-Net passes); they are host-side tensor bookkeeping like the reference's, on device tensors.
Pinned on fixtures written by the reference functions themselves (tests/golden/comp_losses.npz, tests/test_comp_losses.py).

NOT built: the face-detection driven terms (RetinaFace crops, ArcFace alignment through the VAE decoder: external packages absent
from the reference tree) and the optical-flow elastic matching (``calc_elastic_matching_loss`` :2549-2759 needs the GMA flow model)."""
import numpy as np
import torch
import torch.nn.functional as F

DISTILL_LAYERS = {23: 1.0, 24: 1.0}            # the layers every function here weighs (normalised to sum 1)


def normalize_dict_values(d):
    tot = float(np.sum(list(d.values())))
    return d if tot == 0 else {k: v / tot for k, v in d.items()}


def calc_dyn_loss_scale(loss, base_loss_and_scale, ref_loss_and_scale, valid_scale_range=(0, 100)):
    """Linear map loss -> scale through (base_loss, base_scale) and (ref_loss, ref_scale), clipped."""
    (l0, s0), (l1, s1) = base_loss_and_scale, ref_loss_and_scale
    assert l1 != l0, "ref_loss and base_loss cannot be the same."
    return np.clip((loss - l0) / (l1 - l0) * (s1 - s0) + s0, valid_scale_range[0], valid_scale_range[1])


def masked_mean(ts, mask, instance_weights=None, dim=None, keepdim=False):
    w = 1 if instance_weights is None else instance_weights
    if torch.is_tensor(w):
        w = w.view(list(w.shape) + [1] * (ts.ndim - w.ndim))
    if mask is None:
        return (ts * w).mean()
    mask = mask.expand(ts.shape)
    denom = mask.sum(dim=dim, keepdim=keepdim)
    denom = torch.maximum(denom, torch.full_like(denom, 1e-6))
    return (ts * w * mask).sum(dim=dim, keepdim=keepdim) / denom


def masked_l2_loss(predictions, targets, mask):
    """Mean over the batch of (sum of masked squared differences / number of unmasked ELEMENTS); the mask broadcasts over channels."""
    if mask.ndim != predictions.ndim:
        raise ValueError("mask must have the predictions' number of dimensions")
    dims = tuple(range(1, mask.ndim))
    per_inst = (((predictions - targets) ** 2) * mask).sum(dim=dims)
    count = mask.sum(dim=dims) * predictions.shape[1:].numel() / mask.shape[1:].numel()
    return (per_inst / (count + 1e-8)).mean()


def split_indices_by_instance(indices):
    b, n = indices
    return [(b[b == u], n[b == u]) for u in torch.unique(b)]


def extend_indices_B_by_n_times(indices, n, block_offset):
    """(b, n) token indices of block 0 -> the same tokens in blocks 0 .. n-1 (block i = batch indices + i * block_offset)."""
    if indices is None:
        return None
    b, t = indices
    return torch.cat([b + block_offset * i for i in range(n)]), torch.cat([t] * n)


def sel_emb_attns_by_indices(attn_mat, indices, do_sum=True, do_mean=False):
    """attn_mat [B, L, heads, N]: per instance the rows of its indexed tokens, summed / averaged over those tokens -> [n_inst, heads, N]."""
    rows = [attn_mat[ii].unsqueeze(0) for ii in split_indices_by_instance(indices)]
    if do_sum:
        rows = [r.sum(dim=1) for r in rows]
    elif do_mean:
        rows = [r.mean(dim=1) for r in rows]
    return torch.cat(rows, dim=0)


def resize_mask_to_target_size(mask, target_spatial_area, mode="nearest|bilinear"):
    side = int(np.sqrt(target_spatial_area)) if isinstance(target_spatial_area, int) else target_spatial_area[0]
    near = F.interpolate(mask.float(), size=(side, side), mode="nearest")
    if mode != "nearest|bilinear":
        return near
    return torch.maximum(near, F.interpolate(mask.float(), size=(side, side), mode="bilinear", align_corners=False))


# ----------------------------------------------------------------------------- the losses
def calc_sc_rep_attn_distill_loss(ca_layers_activations, subj_indices_1b, prompt_emb_mask_4b, prompt_pad_mask_4b, sc_fg_mask_percent,
                                  FG_THRES=0.1):
    """Pull the subject-comp instance towards its repeated-prompt twin and its neighbours (only when the detected face covers at least
    FG_THRES of the image): attention probabilities SC -> SC-rep, subject-token k / v SC -> SS, other (and padding) tokens' k / v
    SC -> MC.  Returns (subj_attn, subj_k, nonsubj_k, subj_v, nonsubj_v); plain 0 when skipped."""
    zero = (0, 0, 0, 0, 0)
    if sc_fg_mask_percent < FG_THRES:
        return zero
    weights = normalize_dict_values(dict(DISTILL_LAYERS))
    sc_emb_mask = prompt_emb_mask_4b.squeeze(2).chunk(4)[1]
    sc_pad_mask = prompt_pad_mask_4b.squeeze(2).chunk(4)[1]
    nonsubj = sc_emb_mask.clone()
    nonsubj[subj_indices_1b] = 0
    nonsubj = torch.logical_or(nonsubj, sc_pad_mask).unsqueeze(1)                 # [B, 1, L] over k / v [B, C, L]
    l_attn = l_sk = l_nk = l_sv = l_nv = 0
    for li, attn in ca_layers_activations["attn"].items():
        if li not in weights:
            continue
        w = weights[li]
        a = attn.permute(0, 3, 1, 2)                                              # [4B, L, heads, N]
        _, sc_a, sr_a, _ = a.chunk(4)
        l_attn = l_attn + F.mse_loss(sc_a, sr_a.detach()) * (attn.shape[3] * 10) * w
        ss_k, sc_k, _, mc_k = ca_layers_activations["k"][li].chunk(4)
        ss_v, sc_v, _, mc_v = ca_layers_activations["v"][li].chunk(4)
        tok = lambda t: t.permute(0, 2, 1)[subj_indices_1b]
        l_sk = l_sk + F.mse_loss(tok(sc_k), tok(ss_k).detach()) * w
        l_sv = l_sv + F.mse_loss(tok(sc_v), tok(ss_v).detach()) * w
        l_nk = l_nk + masked_l2_loss(sc_k, mc_k.detach(), nonsubj) * w
        l_nv = l_nv + masked_l2_loss(sc_v, mc_v.detach(), nonsubj) * w
    return l_attn, l_sk, l_nk, l_sv, l_nv


def calc_subj_attn_cross_t_diff_loss(ca_layers_activations, future_ca_layers_activations, subj_indices_1b):
    """Subject-token attention of the SC instance at this denoising step vs the (detached) next step, x10 per layer."""
    weights = normalize_dict_values(dict(DISTILL_LAYERS))
    tot = 0
    for li, attn in ca_layers_activations["attn"].items():
        if li not in weights:
            continue
        sc = attn.permute(0, 3, 1, 2).chunk(4)[1][subj_indices_1b]
        sc_next = future_ca_layers_activations["attn"][li].permute(0, 3, 1, 2).chunk(4)[1][subj_indices_1b]
        tot = tot + F.mse_loss(sc, sc_next.detach()) * 10 * weights[li]
    return tot


def calc_attn_norm_loss(ca_outfeats, ca_attns, subj_indices_2b, BLOCK_SIZE):
    """L1 between the mean |attention| the subject tokens receive in the subject instances and in the (detached) class instances,
    for the single and the compositional prompt, per head."""
    weights = normalize_dict_values(dict(DISTILL_LAYERS))
    k_subj = len(subj_indices_2b[0]) // len(torch.unique(subj_indices_2b[0]))
    idx4 = extend_indices_B_by_n_times(subj_indices_2b, 2, BLOCK_SIZE * 2)
    terms = []
    for li in ca_outfeats:
        if li not in weights:
            continue
        a = ca_attns[li].permute(0, 3, 1, 2)
        subj = a[idx4].reshape(BLOCK_SIZE * 4, k_subj, *a.shape[2:]).sum(dim=1)
        ss, sc, cs, cc = subj.chunk(4)
        lvl = lambda t: t.abs().mean(dim=-1)
        terms.append((F.l1_loss(lvl(sc), lvl(cc.detach())) + F.l1_loss(lvl(ss), lvl(cs.detach()))) * weights[li])
    return sum(terms)


def calc_subj_masked_bg_suppress_loss(ca_attn, subj_indices, BLOCK_SIZE, fg_mask):
    """Attention of the subject tokens that lands on BACKGROUND pixels of the first block, above a tolerance of 0.02, averaged over the
    offending (head, pixel) entries."""
    device = next(iter(ca_attn.values())).device
    if subj_indices is None or len(subj_indices) == 0 or fg_mask is None or fg_mask.chunk(4)[0].float().mean() >= 0.998:
        return torch.tensor(0.0, device=device)
    weights = normalize_dict_values(dict(DISTILL_LAYERS))
    k_subj = len(subj_indices[0]) // len(torch.unique(subj_indices[0]))
    idx = (subj_indices[0][:BLOCK_SIZE * k_subj], subj_indices[1][:BLOCK_SIZE * k_subj])
    tol = 0.02
    terms = []
    for li, attn in ca_attn.items():
        if li not in weights:
            continue
        subj = sel_emb_attns_by_indices(attn.permute(0, 3, 1, 2), idx, do_sum=True)                 # [BLOCK_SIZE, heads, N]
        m = resize_mask_to_target_size(fg_mask, subj.shape[-1]).reshape(BLOCK_SIZE, 1, -1).repeat(1, subj.shape[1], 1)
        fg = (m > 1e-6).to(m.dtype)
        bg = 1 - fg
        if (fg.sum(dim=(1, 2)) == 0).any() or (bg.sum(dim=(1, 2)) == 0).any():
            continue
        excess = subj * bg - tol
        terms.append(masked_mean(excess, excess > 0) * weights[li])
    return sum(terms) if terms else torch.tensor(0.0, device=device)


def comp_rep_distill_total(losses, sc_fg_mask_percent, rep_dist_fg_bounds=(0.1, 0.20, 0.25)):
    """The weighting of the five rep-distillation terms inside calc_comp_feat_distill_loss (ddpm.py:3557-3589): subject terms x2,
    non-subject k x5, non-subject v x2, all scaled by a factor that grows with the detected face's share of the image."""
    l_attn, l_sk, l_nk, l_sv, l_nv = losses
    if sc_fg_mask_percent > 0:
        scale = calc_dyn_loss_scale(sc_fg_mask_percent, (rep_dist_fg_bounds[1], 0.5), (rep_dist_fg_bounds[2], 2), valid_scale_range=(0.05, 2))
    else:
        scale = 0
    return ((l_attn + l_sk + l_sv) * 2 + l_nk * 5 + l_nv * 2) * scale
