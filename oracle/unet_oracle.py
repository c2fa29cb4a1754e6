"""CPU oracle for the SD-1.5 U-Net epsilon-prediction path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 restatement (torch-CPU functional ops over a state dict) of the
reference's in-tree LDM U-Net.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``adaface_dev_amd``) never does and fails loudly if its HIP library is missing.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imports the real reference modules
from /root/reference in the build container and commits their outputs under
``tests/golden/``; ``tests/test_oracle_vs_golden.py`` checks this restatement against
them (tiny-config full tensors incl. captures and masks; full SD-1.5-size epsilon).

Every function cites the reference lines (relative to /root/reference) it follows.
Nothing here reads /root/reference at run time.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- leaf ops
def timestep_embedding(timesteps, dim, max_period=10000):
    """ldm/modules/diffusionmodules/util.py:154-174 (repeat_only=False branch)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm(x, w, b, eps, silu=False, groups=32):
    """GroupNorm32 (util.py:195-212, eps 1e-5) / Normalize (attention.py:70-71, eps 1e-6);
    optionally followed by nn.SiLU (openaimodel.py:202-233)."""
    y = F.group_norm(x.float(), groups, w, b, eps)
    return F.silu(y) if silu else y


def conv2d(x, w, b, stride=1, padding=1):
    """conv_nd(2, ...) = nn.Conv2d (util.py:214-224)."""
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def attention_core(q, k, v, heads, mask=None, want_probs=False):
    """CrossAttention.forward core, attention.py:180-204.
    q [b,n,C], k/v [b,l,C]; mask [b,l] bool (True = keep) or None."""
    b, n, C = q.shape
    l = k.shape[1]
    d = C // heads
    scale = d ** -0.5
    qh = q.reshape(b, n, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(b, l, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(b, l, heads, d).permute(0, 2, 1, 3)
    score = torch.einsum("bhid,bhjd->bhij", qh, kh) * scale
    if mask is not None:
        neg = -torch.finfo(score.dtype).max
        score = score.masked_fill(~mask.bool()[:, None, None, :], neg)
    attn = score.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, vh)
    out = out.permute(0, 2, 1, 3).reshape(b, n, C)
    if want_probs:
        return out, attn, score
    return out


def cross_attention(sd, p, x, context=None, mask=None, heads=8, capture=None, rewrite=None):
    """CrossAttention.forward, attention.py:168-222.  `capture`: dict to fill or None.  `rewrite`: the live path's explicit attention
    of the captured layers (adaface/diffusers_attn_lora_capture.py:79-139, 309-315, 344-362; pinned through oracle/capture_oracle.py):
    dict with normalize_cross_attn / mix_attn_mats_in_batch / subj_indices / cross_attn_scale_factor -- the scores are rewritten
    between the product and the softmax, and q2 / k / v are captured as well."""
    q = F.linear(x, sd[p + "to_q.weight"])
    ctx = x if context is None else context
    loras = (rewrite or {}).get("attn_lora") or {}                # {'q' | 'k' | 'v' | 'out': (A, B, magnitude, scaling)}, no dropout

    def proj(name, inp, wkey, bkey=None):
        from . import lora_oracle as LO
        bias = None if bkey is None else sd[p + bkey]
        if name in loras:
            A, Bm, mag, scaling = loras[name]
            return LO.dora_linear_train(inp, sd[p + wkey], bias, A, Bm, mag, scaling)
        return F.linear(inp, sd[p + wkey], bias)
    q2 = proj("q", x, "to_q.weight") if "q" in loras else q      # the q adapter feeds query2 only (:239-249)
    if (rewrite or {}).get("q_lora_updates_query"):
        q = q2
    k = proj("k", ctx, "to_k.weight")
    v = proj("v", ctx, "to_v.weight")
    m = None if mask is None else mask.reshape(mask.shape[0], -1)
    d = q.shape[-1] // heads
    if rewrite is not None and capture is not None:
        from . import capture_oracle as C
        b, n, Cq = q.shape
        split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
        o, score, attn = C.scaled_dot_product_attention(split(q), split(k), split(v), rewrite.get("cross_attn_scale_factor", torch.tensor(0.8)),
                                                        subj_indices=rewrite.get("subj_indices"),
                                                        normalize_cross_attn=bool(rewrite.get("normalize_cross_attn", False)),
                                                        mix_attn_mats_in_batch=bool(rewrite.get("mix_attn_mats_in_batch", False)))
        core = o.permute(0, 2, 1, 3).reshape(b, n, Cq)
    else:
        core, attn, score = attention_core(q, k, v, heads, m, want_probs=True)
    out = proj("out", core, "to_out.0.weight", "to_out.0.bias")
    if capture is not None:
        s = math.sqrt(d ** -0.5)
        capture["q"] = q.permute(0, 2, 1).contiguous() * s          # attention.py:212
        capture["q2"] = q2.permute(0, 2, 1).contiguous() * s         # query2 = query without a q LoRA (diffusers_attn_lora_capture.py:250)
        capture["k"] = k.permute(0, 2, 1).contiguous() * s
        capture["v"] = v.permute(0, 2, 1).contiguous() * s
        capture["attn"] = attn.contiguous()                          # :217
        capture["attnscore"] = score.contiguous()                    # :218
        capture["attn_out"] = out.permute(0, 2, 1).contiguous()      # :220
    return out


def geglu_ff(sd, p, x):
    """FeedForward(glu=True), attention.py:31-58: Linear C->8C, a*gelu(g), Linear 4C->C."""
    h = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
    a, g = h.chunk(2, dim=-1)
    h = a * F.gelu(g)
    return F.linear(h, sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def basic_transformer_block(sd, p, x, context, mask, heads, capture=None, rewrite=None):
    """BasicTransformerBlock._forward, attention.py:242-252 (LayerNorm eps 1e-5)."""
    C = x.shape[-1]
    ln = lambda t, n: F.layer_norm(t, (C,), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5)
    x1 = cross_attention(sd, p + "attn1.", ln(x, "norm1"), None, mask, heads) + x
    x2 = x1 + cross_attention(sd, p + "attn2.", ln(x1, "norm2"), context, None, heads, capture, rewrite)
    return geglu_ff(sd, p + "ff.", ln(x2, "norm3")) + x2


def spatial_transformer(sd, p, x, context, mask, heads, capture=None, rewrite=None):
    """SpatialTransformer.forward, attention.py:287-304."""
    b, c, h, w = x.shape
    x_in = x
    y = group_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    y = conv2d(y, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"], padding=0)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    mask2 = None
    if mask is not None:
        mask2 = F.interpolate(mask, size=(h, w), mode="nearest")         # attention.py:298
    y = basic_transformer_block(sd, p + "transformer_blocks.0.", y, context, mask2, heads, capture, rewrite)
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    y = conv2d(y, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"], padding=0)
    return y + x_in


def res_block(sd, p, x, emb, lora=None):
    """ResBlock._forward, openaimodel.py:256-276 (no up/down, no scale-shift).  lora: optional {"conv1" | "conv2" |
    "conv_shortcut": (lora_A, lora_B, magnitude, scaling, mask | None)} -- DoRA adapters on the block's convolutions as the live
    path attaches them (oracle/lora_oracle.py::dora_conv2d_train; diffusers_attn_lora_capture.py:541-591)."""
    from . import lora_oracle as LO
    lora = lora or {}

    def conv(h, wname, key, padding=1):
        w, b = sd[p + wname + ".weight"], sd[p + wname + ".bias"]
        if key in lora:
            A, Bm, m, scaling, mask = lora[key]
            return LO.dora_conv2d_train(h, w, b, A, Bm, m, scaling, mask, 1, padding)
        return conv2d(h, w, b, padding=padding)

    h = group_norm(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"], 1e-5, silu=True)
    h = conv(h, "in_layers.2", "conv1")
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = group_norm(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"], 1e-5, silu=True)
    h = conv(h, "out_layers.3", "conv2")
    if (p + "skip_connection.weight") in sd:
        x = conv(x, "skip_connection", "conv_shortcut", padding=0)
    return x + h


# --------------------------------------------------------------------------- topology
def unet_topology(cfg):
    """Layer kinds per block, following UNetModel.__init__, openaimodel.py:520-690.
    Returns (input_blocks, middle, output_blocks); each block is a list of
    (kind, key_prefix) with kind in {conv_in, res, attn, down, up}."""
    mc = cfg["model_channels"]
    mult = cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]
    ares = cfg["attention_resolutions"]
    inputs = [[("conv_in", "input_blocks.0.0.")]]
    ds = 1
    idx = 1
    for level, _m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", f"input_blocks.{idx}.0.")]
            if ds in ares:
                layers.append(("attn", f"input_blocks.{idx}.1."))
            inputs.append(layers)
            idx += 1
        if level != len(mult) - 1:
            inputs.append([("down", f"input_blocks.{idx}.0.")])
            idx += 1
            ds *= 2
    middle = [("res", "middle_block.0."), ("attn", "middle_block.1."), ("res", "middle_block.2.")]
    outputs = []
    idx = 0
    for level, _m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            layers = [("res", f"output_blocks.{idx}.0.")]
            j = 1
            if ds in ares:
                layers.append(("attn", f"output_blocks.{idx}.{j}."))
                j += 1
            if level and i == nrb:
                layers.append(("up", f"output_blocks.{idx}.{j}."))
                ds //= 2
            outputs.append(layers)
            idx += 1
    return inputs, middle, outputs


CAPTURED_LAYERS = (22, 23, 24)      # openaimodel.py:852


def unet_forward(sd, cfg, x, timesteps, context, extra_info=None):
    """UNetModel.forward, openaimodel.py:820-952.  x [b,4,H,W] fp32, context [b,T,ctx]."""
    heads = cfg["num_heads"]
    extra_info = {} if extra_info is None else extra_info
    capture_on = bool(extra_info.get("capture_ca_activations", False))
    img_mask = extra_info.get("img_mask", None)
    inputs, middle, outputs = unet_topology(cfg)

    t_emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])

    acts = {}

    def run_block(layers, h, layer_idx):
        cap = None
        for kind, p in layers:
            if kind == "conv_in":
                h = conv2d(h, sd[p + "weight"], sd[p + "bias"])
            elif kind == "res":
                h = res_block(sd, p, h, emb, (extra_info.get("ffn_lora") or {}).get(p))
            elif kind == "attn":
                cap = {} if (capture_on and layer_idx in CAPTURED_LAYERS) else None
                rw = None
                al = (extra_info.get("attn_lora") or {}).get(layer_idx)
                if layer_idx in CAPTURED_LAYERS and (extra_info.get("normalize_cross_attn") or extra_info.get("mix_attn_mats_in_batch") or al):
                    cap = {} if cap is None else cap
                    factors = extra_info.get("cross_attn_scale_factors")
                    rw = dict(normalize_cross_attn=extra_info.get("normalize_cross_attn", False), subj_indices=extra_info.get("subj_indices"),
                              mix_attn_mats_in_batch=extra_info.get("mix_attn_mats_in_batch", False),
                              cross_attn_scale_factor=factors[CAPTURED_LAYERS.index(layer_idx)] if factors is not None else torch.tensor(0.8),
                              attn_lora=al, q_lora_updates_query=extra_info.get("q_lora_updates_query", False))
                h = spatial_transformer(sd, p, h, context, img_mask, heads, cap, rw)
            elif kind == "down":
                h = conv2d(h, sd[p + "op.weight"], sd[p + "op.bias"], stride=2)     # :135-161
            elif kind == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")                # :117
                h = conv2d(h, sd[p + "conv.weight"], sd[p + "conv.bias"])
        if cap is not None:
            cap["outfeat"] = h
            acts[layer_idx] = cap
        return h

    h = x.float()
    hs = []
    layer_idx = 0
    for layers in inputs:
        h = run_block(layers, h, layer_idx)
        hs.append(h)
        layer_idx += 1
    h = run_block(middle, h, layer_idx)
    layer_idx += 1
    gs = float(extra_info.get("res_hidden_states_gradscale", 1) or 1)
    for oi, layers in enumerate(outputs):
        skip = hs.pop()
        if gs != 1 and oi >= 3:
            # live path only (diffusers_attn_lora_capture.py:366-446, set_lora_and_capture_flags :606-609): the GRADIENT of
            # the skip tensors entering up_blocks[1:] (= LDM output_blocks 3..11) is scaled; forward values are unchanged
            skip = skip * gs + (skip * (1 - gs)).detach()
        h = torch.cat([h, skip], dim=1)                                             # :916-918
        h = run_block(layers, h, layer_idx)
        layer_idx += 1

    extra_info["ca_layers_activations"] = {                                        # :931-935
        key: {li: acts[li][key] for li in acts}
        for key in ("outfeat", "attn", "attnscore", "q", "q2", "k", "v", "attn_out")
    }
    h = group_norm(h, sd["out.0.weight"], sd["out.0.bias"], 1e-5, silu=True)
    return conv2d(h, sd["out.2.weight"], sd["out.2.bias"])
