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

Round 3 added the feature-matching ("elastic matching") losses on the captured ``q2`` / ``attn_out`` / ``outfeat`` tensors --
``calc_comp_subj_bg_preserve_loss`` :1920-2045, ``calc_elastic_matching_loss`` :2549-2759, ``calc_sc_recon_ssfg_mc_losses`` :2314-2547 --
in the form the reference runs by default: ``flow_model is None`` (``ddpm.py:652-662``: the GMA flow network is only instantiated under
``use_face_flow_for_sc_matching_loss``, default False), where the "flow" candidate is the same-location matching (:2382-2391); a flow
model is refused.  Also ``calc_recon_and_suppress_losses`` :1715-1754 of the ``do_normal_recon`` iteration and small helpers
(``add_dict_to_dict`` :1097, ``map_bboxes_coords`` :1588, ``clamp`` :81, ``torch_uniform`` :1269, ``var_of_laplacian``).
The face boxes these functions take come from the caller's face detector (the RetinaFace network is an external package)."""
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
    """(b, n) -> [(b of instance u, n of instance u) for the instances u present].  ``torch.unique`` and the boolean selections each make
    the host wait for the device; the same index pair comes back many times per iteration (every layer of every step), so the split is
    kept on the tensor OBJECT it was made from (index tensors are never written in place)."""
    b, n = indices
    memo = getattr(b, "af_split", None)
    if memo is not None and memo[0] is n and memo[2] == (b._version, n._version):      # an in-place write to either tensor voids the memo
        return memo[1]
    out = [(b[b == u], n[b == u]) for u in torch.unique(b)]
    try:
        b.af_split = (n, out, (b._version, n._version))
    except AttributeError:
        pass
    return out


def count_instances(b):
    """len(torch.unique(b)), remembered on the tensor object like ``split_indices_by_instance``."""
    memo = getattr(b, "af_n_unique", None)
    if memo is not None and memo[1] == b._version:
        return memo[0]
    k = len(torch.unique(b))
    try:
        b.af_n_unique = (k, b._version)
    except AttributeError:
        pass
    return k


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
    k_subj = len(subj_indices_2b[0]) // count_instances(subj_indices_2b[0])
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
    # a mask made from host boxes (ddpm_losses.box_mask) carries its host copy: the three decisions on the mask below are then taken on
    # that copy (same arithmetic on 0 / 1 values) instead of reading the device
    memo = getattr(fg_mask, "af_host", None) if fg_mask is not None else None
    fg_host = memo[0] if memo is not None and memo[1] == fg_mask._version else None            # (host copy, version at the copy): ddpm_losses.box_mask
    decide_on = fg_host if fg_host is not None else fg_mask
    if subj_indices is None or len(subj_indices) == 0 or fg_mask is None or decide_on.chunk(4)[0].float().mean() >= 0.998:
        return torch.zeros((), device=device)
    weights = normalize_dict_values(dict(DISTILL_LAYERS))
    k_subj = len(subj_indices[0]) // count_instances(subj_indices[0])
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
        if fg_host is not None:
            fg_d = (resize_mask_to_target_size(fg_host, subj.shape[-1]).reshape(BLOCK_SIZE, 1, -1) > 1e-6).float()
            empty = bool((fg_d.sum(dim=(1, 2)) == 0).any() or ((1 - fg_d).sum(dim=(1, 2)) == 0).any())
        else:
            empty = bool((fg.sum(dim=(1, 2)) == 0).any() or (bg.sum(dim=(1, 2)) == 0).any())
        if empty:
            continue
        excess = subj * bg - tol
        terms.append(masked_mean(excess, excess > 0) * weights[li])
    return sum(terms) if terms else torch.zeros((), device=device)


def comp_rep_distill_total(losses, sc_fg_mask_percent, rep_dist_fg_bounds=(0.1, 0.20, 0.25)):
    """The weighting of the five rep-distillation terms inside calc_comp_feat_distill_loss (ddpm.py:3557-3589): subject terms x2,
    non-subject k x5, non-subject v x2, all scaled by a factor that grows with the detected face's share of the image."""
    l_attn, l_sk, l_nk, l_sv, l_nv = losses
    if sc_fg_mask_percent > 0:
        scale = calc_dyn_loss_scale(sc_fg_mask_percent, (rep_dist_fg_bounds[1], 0.5), (rep_dist_fg_bounds[2], 2), valid_scale_range=(0.05, 2))
    else:
        scale = 0
    return ((l_attn + l_sk + l_sv) * 2 + l_nk * 5 + l_nv * 2) * scale


# ----------------------------------------------------------------------------- small helpers of the loss assembly
def clamp(x, min_val, max_val):
    assert min_val <= max_val, "min_val should be less than or equal to max_val"
    return max(min_val, min(x, max_val))


def torch_uniform(low, high, size=1, device=None):
    return torch.rand(size, device=device) * (high - low) + low


def add_dict_to_dict(d1, d2, weight=1, session_prefix=None):
    """d1[k] += d2[k] * weight (missing keys start at 0); keys get ``session_prefix/`` in front when given."""
    for k, v in d2.items():
        k2 = k if session_prefix is None else f"{session_prefix}/{k}"
        d1[k2] = d1.get(k2, 0) + v * weight
    return d1


def map_bboxes_coords(bboxes, W1, W2):
    """Integer boxes on a W1-wide image -> the same boxes on a W2-wide one (pixel 512 -> latent 64), floor division as the reference."""
    return None if bboxes is None else bboxes * W2 // W1


def var_of_laplacian(images):
    """Sharpness score per image (reference ldm/util.py ``var_of_laplacian``): variance of the 3x3 Laplacian of the grey image."""
    grey = images.mean(dim=1, keepdim=True) if images.shape[1] > 1 else images
    # the reference's F.conv2d with the [[0,1,0],[1,-4,1],[0,1,0]] kernel and padding 1, written as the five-point stencil on the zero-padded
    # image: element-wise ops only (a library convolution of a 1-channel image lowers to a vendor-BLAS GEMM on the device)
    p = F.pad(grey, (1, 1, 1, 1))
    lap = p[:, :, :-2, 1:-1] + p[:, :, 2:, 1:-1] + p[:, :, 1:-1, :-2] + p[:, :, 1:-1, 2:] - 4.0 * grey
    return lap.var(dim=(1, 2, 3))


# ----------------------------------------------------------------------------- feature matching between the four blocks (flow_model = None)
MATCHING_TYPES = ("attn", "flow", "sameloc")


def _bmm_nt(a, bt):
    """Batched a [B, M, K] @ bt [B, N, K]^T -> fp32 [B, M, N] through autograd_ops.matmul_nt (the MFMA GEMM for device tensors)."""
    from ..autograd_ops import matmul_nt
    return torch.stack([matmul_nt(a[i], bt[i]) for i in range(a.shape[0])], dim=0)


def reconstruct_feat_with_attn_aggregation(sc_feat, sc_to_ss_prob):
    """[B, C, N_sc] x [B, N_sc, N_t] -> [B, N_t, C]: every target token as the probability-weighted sum of the subject-comp tokens."""
    return _bmm_nt(sc_to_ss_prob.transpose(1, 2), sc_feat).to(sc_feat.dtype)


def calc_sc_recon_ssfg_mc_losses(layer_idx, flow_model, target_feats, scfg_feat, scbg_feat, ssfg_q, scfg_q, scbg_q, mc_q, ss2sc_flow, mc2sc_flow,
                                 H, W, small_motion_ignore_thres, num_flow_est_iters, objective_name, verbose=False):
    """How well the subject-comp (SC) features reconstruct (a) the subject-single face features (``ssfg``) and (b) the class-comp
    features (``mc``), per target token, under three matchings: attention aggregation (softmax over the SC tokens of q_sc^T q_target),
    "flow" and same location.  Without a flow model (the reference's default, :2382-2391) the flow candidate IS the same-location one.
    Per token the smallest of {10 x attn, m x flow, sameloc} (m = 1.02 ssfg / 1.1 mc) is optimised (:2453-2462); the sparse
    (identity) matching is also distilled into the attention matching with detached per-token weights (:2464-2516).
    Returns (losses {name: [attn, flow, sameloc, min]}, sparse-distillation losses, win-rate stats, None, None)."""
    if flow_model is not None:
        raise NotImplementedError("calc_sc_recon_ssfg_mc_losses: the GMA optical-flow network (use_face_flow_for_sc_matching_loss) is an "
                                  "external model; the reference's default flow_model=None path is what is built")
    device, B, N = scbg_feat.device, scbg_feat.shape[0], H * W
    # The reference holds the matching probabilities as [B, N_sc, N_t] and normalises over dim 1 (:2336-2344).  Here they are held
    # TRANSPOSED, [B, N_t, N_sc], computed that way round (target queries x SC queries): the softmax then runs over the contiguous axis
    # (torch's strided-softmax kernel took 4.2 + 2.6 ms forward + backward per 4096 x 4096 matrix, 15 % of a Stage-2 micro-batch,
    # profiles/r03w_train2_kernel_stats.txt) and the aggregation GEMM reads them without a transposed copy.  Everything below is
    # element-wise or a sum, so it is the same arithmetic with the two token axes swapped.
    probs_t = {"ssfg": F.softmax(_bmm_nt(ssfg_q.transpose(1, 2), scfg_q.transpose(1, 2)).to(scfg_q.dtype), dim=2),
               "mc": F.softmax(_bmm_nt(mc_q.transpose(1, 2), scbg_q.transpose(1, 2)).to(scbg_q.dtype), dim=2)}
    sources = {"ssfg": scfg_feat, "mc": scbg_feat}
    eye = torch.eye(N, device=device, dtype=scbg_feat.dtype).repeat(B, 1, 1)
    losses, sparse_distill, stats = {}, {}, {}
    for name in ("ssfg", "mc"):
        target = target_feats[name].permute(0, 2, 1)
        sameloc = sources[name].permute(0, 2, 1)
        candidates = (_bmm_nt(probs_t[name], sources[name]).to(sources[name].dtype), sameloc, sameloc)   # reconstruct_feat_with_attn_aggregation
        tok = [F.mse_loss(c, target, reduction="none").mean(dim=2) for c in candidates]           # each [B, N_t]
        losses[name] = [t.mean() for t in tok]
        scaled = torch.stack([tok[0] * 10, tok[1] * (1.1 if name == "mc" else 1.02), tok[2]], dim=0)
        losses[name].append(scaled.min(dim=0).values.mean())
        # distil the sparse matching into the attention matching where it reconstructs better
        adv = scaled[:1] - scaled[1:]                                                             # [2, B, N_t]
        best_adv, best_type = adv.max(dim=0)
        best_adv = best_adv.unsqueeze(1)
        tok_w = (5 * F.layer_norm(best_adv, (best_adv.shape[2],), weight=None, bias=None, eps=1e-5)).sigmoid()
        # the sparse matching of the winning candidate, as [B, N_t, N_sc] like probs_t (both candidates are the identity without a flow model)
        sparse_t = torch.cat([eye, eye], dim=0).gather(0, best_type.view(B, -1, 1).expand(-1, -1, N))
        sc_w = ((sparse_t + probs_t[name]).detach() * tok_w.detach().transpose(1, 2)).sum(dim=1, keepdim=True)   # tok_w [B,N_t,1] x ensemble [B,N_t,N_sc] -> [B,1,N_sc]
        sparse_distill[name] = ((sparse_t - probs_t[name]).abs() * sc_w).mean()
        for i in range(adv.shape[0]):
            stats[f"{name}_{MATCHING_TYPES[i + 1]}_win_rate"] = torch.logical_and(adv[i] > 0, best_type == i).float().mean(dim=1)
        stats[f"{name}_avg_sparse_distill_weight"] = sc_w.mean()
    return losses, sparse_distill, stats, None, None


def _face_crops_resized(x4d, bboxes, H, W):
    """Per instance the box [x1, y1, x2, y2] of the feature map, bilinearly resized back to (H, W) -- the face brought to a common frame."""
    crops = []
    for bi in range(len(bboxes)):
        x1, y1, x2, y2 = bboxes[bi]
        crops.append(F.interpolate(x4d[:, :, y1:y2, x1:x2], (H, W), mode="bilinear", align_corners=False))
    return torch.cat(crops, dim=0)


def calc_elastic_matching_loss(layer_idx, flow_model, ca_q, ca_attn_out, ca_outfeat, H, W, ss_face_bboxes, sc_face_bboxes,
                               sc_face_shrink_ratio_for_bg_matching_mask=1, recon_scaled_loss_threses={"mc": 0.4, "ssfg": 0.4},
                               recon_max_scale_of_threses=5, small_motion_ignore_thres=0.3, num_flow_est_iters=12):
    """One captured layer: queries and features of the four blocks [SS, SC, SC-rep, MC] -> matching losses of ``outfeat`` and
    ``attn_out`` (averaged).  Faces: the SS and SC boxes are cropped and resized to the full frame and de-meaned together; background:
    SC with its (possibly shrunk) face box zeroed against MC, de-meaned by the average of the two means (the SC mean re-weighted by
    its unmasked share).  A loss whose min-term reaches thres x max_scale is dropped, one above thres is scaled down to thres
    (:2715-2731)."""
    BLOCK_SIZE = ca_q.shape[0] // 4
    ss_q, sc_q, _, mc_q = ca_q.chunk(4)
    as4d = lambda t: t.reshape(*t.shape[:2], H, W)
    sc_bg_mask = torch.ones((BLOCK_SIZE, 1, H, W), device=ca_q.device, dtype=ca_q.dtype)
    for bi in range(len(sc_q)):
        x1, y1, x2, y2 = [int(v * sc_face_shrink_ratio_for_bg_matching_mask) for v in sc_face_bboxes[bi]]
        sc_bg_mask[bi, :, y1:y2, x1:x2] = 0
    bg3d = sc_bg_mask.reshape(*sc_bg_mask.shape[:-2], -1)
    bg_share = bg3d.numel() / (bg3d.sum() + 1e-5)

    def split_fg_bg(ss, sc, mc):
        """-> (ssfg, scfg, scbg, mc) de-meaned as described above."""
        ssfg = _face_crops_resized(as4d(ss), ss_face_bboxes, H, W).reshape(*ss.shape)
        scfg = _face_crops_resized(as4d(sc), sc_face_bboxes, H, W).reshape(*sc.shape)
        fg_mean = torch.cat([ssfg, scfg], dim=0).mean(dim=(0, 2), keepdim=True).detach()
        scbg = sc * bg3d
        bg_mean = ((mc.mean(dim=(0, 2), keepdim=True) + scbg.mean(dim=(0, 2), keepdim=True) * bg_share) / 2).detach()
        return ssfg - fg_mean, scfg - fg_mean, (scbg - bg_mean) * bg3d, mc - bg_mean

    ssfg_q, scfg_q, scbg_q, mc_q = split_fg_bg(ss_q, sc_q, mc_q)
    kept = {"ssfg": [], "mc": []}
    sparse = {"ssfg": [], "mc": []}
    all_stats = {}
    total, discarded = 0, 0
    ss2sc_flow = mc2sc_flow = None
    for feat_type, feat_obj in (("outfeat", ca_outfeat), ("attn_out", ca_attn_out)):
        ss_f, sc_f, _, mc_f = feat_obj.chunk(4)
        ssfg_f, scfg_f, scbg_f, mc_f = split_fg_bg(ss_f, sc_f, mc_f)
        losses, sparse_obj, stats, ss2sc_flow, mc2sc_flow = calc_sc_recon_ssfg_mc_losses(
            layer_idx, flow_model, {"ssfg": ssfg_f, "mc": mc_f}, scfg_f, scbg_f, ssfg_q, scfg_q, scbg_q, mc_q, ss2sc_flow, mc2sc_flow,
            H, W, small_motion_ignore_thres, num_flow_est_iters, objective_name=feat_type.ljust(8))
        total, discarded = 0, 0              # as the reference (:2706-2707): the reported ratio is that of the LAST feature type
        for name, ls in losses.items():
            ls = torch.stack(ls, dim=0)
            total += 1
            thres = recon_scaled_loss_threses[name]
            if ls[-1] >= thres * recon_max_scale_of_threses:
                discarded += 1
            else:
                kept[name].append(ls * min(thres / (ls[-1].detach().item() + 1e-6), 1))
            sparse[name].append(sparse_obj[name])
        for k, v in stats.items():
            all_stats.setdefault(k, []).append(v)
    losses_sc_recons = {n: (torch.stack(v, dim=0).mean(dim=0) if v else torch.zeros(4, device=ca_q.device)) for n, v in kept.items()}
    sparse_distill = {n: torch.stack(v).mean() for n, v in sparse.items()}
    stats = {k: torch.stack(v).mean() for k, v in all_stats.items()}
    return losses_sc_recons, sparse_distill, stats, torch.tensor(discarded / (total + 1e-6))


PRESERVE_MONITORS = ("loss_sc_recon_ssfg_attn_agg", "loss_sc_recon_ssfg_flow", "loss_sc_recon_ssfg_sameloc", "loss_sc_recon_ssfg_min",
                     "loss_sc_recon_mc_attn_agg", "loss_sc_recon_mc_flow", "loss_sc_recon_mc_sameloc", "loss_sc_recon_mc_min")


def calc_comp_subj_bg_preserve_loss(mon_loss_dict, session_prefix, device, flow_model, ca_layers_activations, ss_face_bboxes, sc_face_bboxes,
                                    sc_face_shrink_ratio_for_bg_matching_mask=1, recon_scaled_loss_threses={"mc": 0.4, "ssfg": 0.4},
                                    recon_max_scale_of_threses=5, do_sc_fg_faces_suppress=False):
    """Preserve the subject's face (SC vs SS) and the composition's background (SC vs MC) in the captured features of layers 22-24
    (equal weights): 0.1 x the ssfg min-loss (0 when the SC face is being suppressed) + 0.2 x the mc min-loss; the sparse-attention
    distillation terms are monitored at weight 0 (:2032-2043).  Adds its monitors to ``mon_loss_dict`` under ``session_prefix/``."""
    outfeats, attn_outs, qs = ca_layers_activations["outfeat"], ca_layers_activations["attn_out"], ca_layers_activations["q2"]
    layer_w = normalize_dict_values({22: 1, 23: 1, 24: 1})
    opt = {k: torch.zeros((), device=device) for k in ("loss_sc_recon_ssfg_min", "loss_sc_recon_mc_min",
                                                         "loss_sc_to_ssfg_sparse_attns_distill", "loss_sc_to_mc_sparse_attns_distill")}
    for li, outfeat in outfeats.items():
        if li not in layer_w:
            continue
        q = qs[li]
        qh = int(np.sqrt(q.shape[2] * outfeat.shape[2] // outfeat.shape[3]))
        qw = q.shape[2] // qh
        if outfeat.shape[2:] != (qh, qw):
            outfeat = F.interpolate(outfeat, size=(qh, qw), mode="bilinear", align_corners=False)
        recons, sparse, stats, discarded = calc_elastic_matching_loss(
            li, flow_model, q, attn_outs[li], outfeat.reshape(*outfeat.shape[:2], -1), qh, qw, ss_face_bboxes, sc_face_bboxes,
            sc_face_shrink_ratio_for_bg_matching_mask=sc_face_shrink_ratio_for_bg_matching_mask,
            recon_scaled_loss_threses=recon_scaled_loss_threses, recon_max_scale_of_threses=recon_max_scale_of_threses,
            small_motion_ignore_thres=0.3, num_flow_est_iters=12)
        if recons is None:
            continue
        named = dict(zip(PRESERVE_MONITORS, list(recons["ssfg"]) + list(recons["mc"])))
        named["loss_sc_to_ssfg_sparse_attns_distill"], named["loss_sc_to_mc_sparse_attns_distill"] = sparse["ssfg"], sparse["mc"]
        add_dict_to_dict(opt, {k: named[k] for k in opt}, layer_w[li], None)
        add_dict_to_dict(mon_loss_dict, dict(named, discarded_loss_ratio=discarded), layer_w[li], session_prefix)
        add_dict_to_dict(mon_loss_dict, stats, layer_w[li], session_prefix)
    return (opt["loss_sc_recon_ssfg_min"] * (0 if do_sc_fg_faces_suppress else 0.1) + opt["loss_sc_recon_mc_min"] * 0.2
            + opt["loss_sc_to_ssfg_sparse_attns_distill"] * 0 + opt["loss_sc_to_mc_sparse_attns_distill"] * 0)


# ----------------------------------------------------------------------------- do_normal_recon iteration (ldm/util.py:1715-1754)
def calc_recon_and_suppress_losses(noise_gt, noise_pred, noise_pred_cls, face_detected_inst_weights, ca_layers_activations, all_subj_indices,
                                   img_mask, fg_mask, bg_pixel_weight, BLOCK_SIZE, recon_on_pure_noise):
    """(masked eps MSE against the true noise, masked eps MSE of the BACKGROUND against the class-prompt prediction, subject-token
    attention landing on the background) of one recon denoising step."""
    from .util import calc_recon_loss
    dev = noise_pred.device
    if not recon_on_pure_noise:
        loss_recon, _ = calc_recon_loss(F.mse_loss, noise_pred, noise_gt, img_mask, fg_mask, face_detected_inst_weights,
                                        fg_pixel_weight=1, bg_pixel_weight=bg_pixel_weight)
    else:
        loss_recon = torch.zeros((), device=dev)
    if noise_pred_cls is not None:
        bg_mask = 1 - fg_mask
        if bg_mask.sum() == 0:
            bg_mask = torch.ones_like(noise_pred_cls)
        if img_mask is not None:
            bg_mask = bg_mask * img_mask
        loss_recon_cls, _ = calc_recon_loss(F.mse_loss, noise_pred, noise_pred_cls, img_mask, bg_mask, face_detected_inst_weights,
                                            fg_pixel_weight=1, bg_pixel_weight=bg_pixel_weight)
    else:
        loss_recon_cls = torch.zeros((), device=dev)
    return loss_recon, loss_recon_cls, calc_subj_masked_bg_suppress_loss(ca_layers_activations["attn"], all_subj_indices, BLOCK_SIZE, fg_mask)
