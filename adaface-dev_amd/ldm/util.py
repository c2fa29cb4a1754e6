"""Hot-path slice of the reference's ``ldm/util.py``: the masked reconstruction loss used by the
U-Net distillation objective and the per-rank / per-batch seeding.  (The remaining ~2.7 kLoC of
that file is the Stage-2 loss zoo: SURVEY.md section 8f rank 4.)

The loss runs on a [B,4,64,64] fp32 tensor (64 KB per sample): it is host-level glue written with
torch tensor ops on the device (differentiable through torch autograd up to the U-Net node)."""
import random

import numpy as np
import torch


def calc_recon_loss(loss_func, noise_pred, noise_gt, img_mask, fg_mask, instance_weights=None, fg_pixel_weight=1,
                    bg_pixel_weight=1):
    """Pixel-wise loss weighted separately on foreground / background (reference ldm/util.py:1678-1713).
    img_mask, fg_mask: [BS,1,H,W] or None; returns (loss, per-pixel loss)."""
    if img_mask is None:
        img_mask = torch.ones_like(noise_pred)
    if fg_mask is None:
        fg_mask = torch.ones_like(noise_pred)
    if instance_weights is None:
        instance_weights = torch.ones_like(noise_pred)
    else:
        if instance_weights.sum() == 0:
            z = torch.zeros((), device=noise_pred.device)
            return z, z
        instance_weights = instance_weights.float().reshape(-1, 1, 1, 1)
    fg_mask = fg_mask * instance_weights
    img_mask = img_mask * instance_weights
    noise_pred = noise_pred * img_mask
    noise_gt = noise_gt * img_mask
    loss_recon_pixels = loss_func(noise_pred, noise_gt, reduction="none")
    weighted_fg_mask = (fg_mask * img_mask * fg_pixel_weight).expand_as(loss_recon_pixels)
    weighted_bg_mask = ((1 - fg_mask) * img_mask * bg_pixel_weight).expand_as(loss_recon_pixels)
    loss_recon = ((loss_recon_pixels * weighted_fg_mask).sum() + (loss_recon_pixels * weighted_bg_mask).sum()) / (
        weighted_fg_mask.sum() + weighted_bg_mask.sum() + 1e-6)
    return loss_recon, loss_recon_pixels


def set_seed_per_rank_and_batch(rank, epoch, iteration, base_seed=42):
    """Reference ldm/util.py:524-530: distinct, reproducible streams per rank and batch."""
    seed = base_seed + epoch * 10 ** 6 + iteration + rank * 10 ** 8
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed + 1)
    random.seed(seed + 2)
    return seed


# ----------------------------------------------------------------------------- prompt-token bookkeeping (embedding manager)
# Host-side index arithmetic on [B, 77] token-id tensors: restated from the reference helpers with the same results; the
# reference's ``breakpoint()`` guards are ValueErrors here.
def split_indices_by_instance(indices, as_dict=False):
    """(B idx, N idx) -> per-instance groups, instances in ascending order (reference ldm/util.py:1051-1058)."""
    b, n = indices
    groups = [(int(u), b == u) for u in torch.unique(b)]
    if as_dict:
        return {u: n[sel] for u, sel in groups}
    return [(b[sel], n[sel]) for _, sel in groups]


def extract_first_index_in_each_instance(token_indices):
    """Keep the first occurrence of a token in every instance that has it (reference ldm/util.py:1384-1394)."""
    b, n = token_indices
    order = torch.argsort(b, stable=True)
    bs, ns = b[order], n[order]
    first = torch.ones_like(bs, dtype=torch.bool)
    first[1:] = bs[1:] != bs[:-1]
    return bs[first], ns[first]


def scan_cls_delta_strings(tokenized_text, placeholder_indices_1st, subj_name_to_cls_delta_tokens, MAX_SEARCH_SPAN=5):
    """In a batch whose first half holds the subject token and whose second half holds class prompts, find where the class
    string (e.g. "young woman": 2 tokens) starts in every class prompt, searching MAX_SEARCH_SPAN tokens from the slot of the
    paired subject token.  -> [(batch_i, start_N, num_tokens, subj_name)]  (reference ldm/util.py:616-678)."""
    if not subj_name_to_cls_delta_tokens:
        return []
    ph_b, ph_n = placeholder_indices_1st
    if len(torch.unique(ph_b)) != len(ph_b):
        raise ValueError("scan_cls_delta_strings: more than one first-occurrence index per instance")
    BS = tokenized_text.shape[0]
    if len(ph_b) == BS:
        return []
    half = BS // 2
    if len(ph_b) != half or (ph_b != torch.arange(half, device=tokenized_text.device)).any():
        raise ValueError("scan_cls_delta_strings: the subject token must occur in exactly the first half of the batch")
    rows = tokenized_text.tolist()
    found = []
    for i in range(half, BS):
        start0 = int(ph_n[i - half])
        hit = None
        for j in range(MAX_SEARCH_SPAN + 1):
            s = start0 + j
            for name, toks in subj_name_to_cls_delta_tokens.items():
                toks = toks.tolist() if torch.is_tensor(toks) else list(toks)
                if rows[i][s:s + len(toks)] == toks:
                    hit = (i, s, len(toks), name)
                    break
            if hit:
                break
        if hit:
            found.append(hit)
    return found


def merge_cls_token_embeddings(prompt_embedding, cls_delta_string_indices):
    """Sum the M embeddings of a class string into its first slot and shift the rest of the prompt left by M - 1, so the class
    token lines up with the subject token of the paired instance (reference ldm/util.py:683-741).  The vacated tail keeps its old
    values, as in the reference."""
    if not cls_delta_string_indices:
        return prompt_embedding
    out = prompt_embedding.clone()
    shift = {}
    for bi, start, M, _name in sorted(cls_delta_string_indices, key=lambda x: (x[0], x[1])):
        off = shift.get(bi, 0)
        out[bi, start - off] = prompt_embedding[bi, start:start + M].sum(dim=0)
        red = off + M - 1
        if red > 0:
            out[bi, start - off + 1: -red] = prompt_embedding[bi, start + M:]
        shift[bi] = red
    return out


def select_and_repeat_instances(sel_indices, REPEAT, *args):
    """args[sel_indices] repeated REPEAT times along dim 0 (tensors / arrays) or as a list (reference ldm/util.py:1366-1380)."""
    out = []
    for a in args:
        if a is None:
            out.append(None)
        elif isinstance(a, (torch.Tensor, np.ndarray)):
            sel = a[sel_indices]
            out.append(sel.repeat([REPEAT] + [1] * (a.ndim - 1)) if isinstance(a, torch.Tensor) else np.tile(sel, [REPEAT] + [1] * (a.ndim - 1)))
        elif isinstance(a, (list, tuple)):
            out.append(a[sel_indices] * REPEAT)
        else:
            raise TypeError(f"select_and_repeat_instances: unsupported argument type {type(a)}")
    return out


def get_clip_tokens_for_string(clip_tokenizer, string, force_single_token=False):
    """Token ids of `string` without BOS / EOS padding (reference ldm/util.py:867-887)."""
    enc = clip_tokenizer(string, truncation=True, max_length=77, padding="max_length", return_tensors="pt")
    tokens = enc["input_ids"] if isinstance(enc, dict) or hasattr(enc, "__getitem__") else enc.input_ids
    count = int(torch.count_nonzero(tokens - 49407)) - 1
    if count < 1:
        raise ValueError(f"No token found in string '{string}'")
    if force_single_token and count != 1:
        raise ValueError(f"String '{string}' maps to more than a single token. Please use another string")
    return tokens[0, 1:1 + count]


def get_embeddings_for_clip_tokens(embedder, tokens):
    """embedder: the text model's embedding module; -> [N, 768] (reference ldm/util.py:898-902)."""
    return embedder(tokens)[0]


def anneal_value(training_percent, final_percent, value_range):
    """Reference ldm/util.py:1242-1251."""
    if not -1e-6 <= training_percent <= 1 + 1e-6:
        raise ValueError("training_percent must lie in [0, 1]")
    v0, v1 = value_range
    return v0 + (v1 - v0) * training_percent if training_percent < final_percent else v1


def anneal_perturb_embedding(embeddings, training_percent, begin_noise_std_range, end_noise_std_range, perturb_prob,
                             perturb_std_is_relative=True, keep_norm=False, std_dim=-1, norm_dim=-1, verbose=False):
    """With probability perturb_prob add Gaussian noise whose (relative) std is drawn uniformly from the annealed range
    (reference ldm/util.py:1569-1585).  Consumes the global torch RNG in the reference's order: rand(1), rand(1), randn_like."""
    from ..adaface.util import perturb_tensor
    if torch.rand(1) > perturb_prob:
        return embeddings
    if end_noise_std_range is not None:
        lb = anneal_value(training_percent, 1, (begin_noise_std_range[0], end_noise_std_range[0]))
        ub = anneal_value(training_percent, 1, (begin_noise_std_range[1], end_noise_std_range[1]))
    else:
        lb, ub = begin_noise_std_range
    std = torch.rand(1).item() * (ub - lb) + lb
    return perturb_tensor(embeddings, std, perturb_std_is_relative, keep_norm, std_dim=std_dim, norm_dim=norm_dim)


# ----------------------------------------------------------------------------- prompt-delta regularisation (host glue on [4, 77, 768])
def ortho_subtract(a, b, b_discount=1, on_last_n_dims=1, return_align_coeffs=False):
    """a minus its projection onto b over the last ``on_last_n_dims`` dims: a - b <a,b> / (<b,b> + 1e-6) (reference ldm/util.py:296-332)."""
    if a.ndim != b.ndim:
        raise ValueError("ortho_subtract: a and b must have the same number of dimensions")
    shape = None
    if on_last_n_dims > 1:
        if a.numel() < b.numel():
            a = a.expand(b.shape)
        elif b.numel() < a.numel():
            b = b.expand(a.shape)
        shape = a.shape
        a = a.reshape(*shape[:-on_last_n_dims], -1)
        b = b.reshape(*shape[:-on_last_n_dims], -1)
    w = (a * b).sum(dim=-1) / ((b * b).sum(dim=-1) + 1e-6)
    out = a - b * w.unsqueeze(-1) * b_discount
    if shape is not None:
        out = out.reshape(shape)
        w = w.reshape(list(shape[:-on_last_n_dims]) + [1] * on_last_n_dims)
    return (out, w) if return_align_coeffs else out


def demean(x, demean_dims=(-1,)):
    """Reference ldm/util.py:334-340."""
    return x if demean_dims is None else x - x.mean(dim=list(demean_dims), keepdim=True)


def calc_ref_cosine_loss(delta, ref_delta, emb_mask=None, exponent=2, do_demeans=(False, False), first_n_dims_into_instances=2,
                         ref_grad_scale=0, aim_to_align=True, reduction="mean"):
    """Mean (1 - cos) between every embedding of ``delta`` and the sign-preserving power of the matching embedding of ``ref_delta``,
    per batch item over the tokens whose mask is > 0, weighted by the mask; the reference side gets ``ref_grad_scale`` of the gradient
    (reference ldm/util.py:365-470)."""
    import torch.nn.functional as F
    from ..adaface.subj_basis_generator import ScaleGrad
    per_item = []
    for i in range(delta.shape[0]):
        d, r = delta[i:i + 1], ref_delta[i:i + 1]
        m = None if emb_mask is None else emb_mask[i:i + 1]
        if m is not None:
            lead = d.shape[:first_n_dims_into_instances]
            keep = (m > 0).squeeze(-1).expand(lead)
            d, r = d[keep], r[keep]
            m = m.squeeze(-1).expand(lead)[keep]
        else:
            d = d.reshape(d.shape[:first_n_dims_into_instances].numel(), -1)
            r = r.reshape(d.shape)
        if do_demeans[0]:
            d = demean(d)
        if do_demeans[1]:
            r = demean(r)
        r = ScaleGrad.apply(r, ref_grad_scale)
        r_pow = r * r.abs().pow(exponent - 1)
        label = torch.full_like(d[:, 0], 1.0 if aim_to_align else -1.0)
        li = F.cosine_embedding_loss(d, r_pow, label, reduction="none")
        if m is not None:
            li = li * m
        if reduction == "mean":
            per_item.append(li.sum() / (m.sum() + 1e-8) if m is not None else li.mean())
        elif reduction == "none":
            per_item.append(li)
        else:
            raise ValueError(f"reduction {reduction!r}")
    return sum(per_item) / delta.shape[0] if reduction == "mean" else torch.stack(per_item, dim=0)


def calc_prompt_emb_delta_loss(prompt_embeddings, prompt_emb_mask, cls_delta_grad_scale=0.05):
    """Prompt-delta regularisation (reference ldm/util.py:1426-1480; weight 1e-4 in p_losses, ddpm.py:2285-2293): the batch holds the
    prompt embeddings of (subject-single, subject-comp, class-single, class-comp); the change a composition makes to the SUBJECT
    prompt (orthogonal to the single prompt) should align with the change it makes to the CLASS prompt.  Tokens present in both
    prompts weigh 1, composition-only tokens 0.25, the BOS token 0."""
    ss, sc, cs, cc = prompt_embeddings.chunk(4)
    w = None
    if prompt_emb_mask is not None:
        m = prompt_emb_mask.float().clone()           # (the reference zeroes the BOS row of the caller's float mask in place)
        m[:, 0] = 0
        ms, mc, _, _ = m.chunk(4)
        w = (ms + mc).pow(2) / 4
    subj_delta = ortho_subtract(sc, ss)
    cls_delta = ortho_subtract(cc, cs)
    return calc_ref_cosine_loss(subj_delta, cls_delta, emb_mask=w, do_demeans=(False, True), first_n_dims_into_instances=2,
                                ref_grad_scale=cls_delta_grad_scale, aim_to_align=True)


def collate_dicts(dicts):
    """Concatenate a list of equally structured (nested) dicts of tensors / lists along the batch (reference ldm/util.py:1112-1126)."""
    out = {}
    for k, v in dicts[0].items():
        col = [d[k] for d in dicts]
        if isinstance(v, list):
            out[k] = sum(col, [])
        elif torch.is_tensor(v):
            out[k] = torch.cat(col, dim=0)
        elif isinstance(v, dict):
            out[k] = collate_dicts(col)
        else:
            raise TypeError(f"collate_dicts: {k!r} holds {type(v).__name__}")
    return out


def split_dict(d_all, num_splits):
    """Reverse of collate_dicts (reference ldm/util.py:1129-1165)."""
    result = [{} for _ in range(num_splits)]
    for k, v in d_all.items():
        if isinstance(v, list):
            per = len(v) // num_splits
            parts = [v[i * per:(i + 1) * per] for i in range(num_splits)]
        elif torch.is_tensor(v):
            parts = torch.split(v, v.size(0) // num_splits, dim=0)
        elif isinstance(v, dict):
            parts = split_dict(v, num_splits)
        else:
            raise TypeError(f"split_dict: {k!r} holds {type(v).__name__}")
        for i in range(num_splits):
            result[i][k] = parts[i]
    return result


def recursive_detach(obj):
    """detach every tensor of a nested dict / list / tuple structure."""
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: recursive_detach(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(recursive_detach(v) for v in obj)
    return obj
