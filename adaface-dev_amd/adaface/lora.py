"""Inference-time LoRA / DoRA adapters of the U-Net as a WEIGHT MERGE.

The reference attaches peft DoRA adapters (rank 192, ``lora_alpha`` 16 for the FFN adapters / ``rank // 8`` for attention,
``use_dora=True``; ``adaface/diffusers_attn_lora_capture.py:171-181, 541-556``) to

* the three conv layers of ``up_blocks.3.resnets.{1,2}`` (``conv1``, ``conv2``, ``conv_shortcut``) -- in this package's LDM naming
  ``output_blocks.{10,11}.0.{in_layers.2, out_layers.3, skip_connection}`` -- under three adapter names (``recon_loss``,
  ``unet_distill``, ``comp_distill``), and
* ``to_q / to_k / to_v / to_out.0`` of the cross-attention layers of ``up_blocks.3`` (LDM ``output_blocks.{9,10,11}.1...attn2``).

In eval mode (no LoRA dropout) peft's DoRA layer computes, per output channel c (``DoraConv2dLayer.forward`` / ``DoraLinearLayer.forward``)

    y = base(x) + (s_c - 1) * conv(x, W) + s_c * scaling * B(A(x)),      s_c = m_c / || W + scaling * B A ||_c

which is exactly a convolution with the merged weight  W'_c = s_c * (W + scaling * B A)_c  and the base bias.  On this path the
adapters therefore cost NOTHING at run time: ``merge_*`` rewrites the layer's weight in place (the packed fp16 copy is rebuilt on
the next call because the parameter version changes) and returns what is needed to restore it.  Training-time DoRA (dropout on the
adapter branch, gradients to A / B / m) is the next row.  peft itself is not available offline: the formula
above is restated from peft's published DoRA layer and is "parity unpinned" (oracle: ``oracle/lora_oracle.py``)."""

import torch

FFN_LORA_TARGETS = {                     # diffusers name (as in the reference's checkpoints) -> LDM module path
    "up_blocks.3.resnets.1.conv1": "output_blocks.10.0.in_layers.2",
    "up_blocks.3.resnets.1.conv2": "output_blocks.10.0.out_layers.3",
    "up_blocks.3.resnets.1.conv_shortcut": "output_blocks.10.0.skip_connection",
    "up_blocks.3.resnets.2.conv1": "output_blocks.11.0.in_layers.2",
    "up_blocks.3.resnets.2.conv2": "output_blocks.11.0.out_layers.3",
    "up_blocks.3.resnets.2.conv_shortcut": "output_blocks.11.0.skip_connection",
}
ATTN_LORA_TARGETS = {
    f"up_blocks.3.attentions.{i}.transformer_blocks.0.attn2.{d}": f"output_blocks.{9 + i}.1.transformer_blocks.0.attn2.{l}"
    for i in range(3) for d, l in (("to_q", "to_q"), ("to_k", "to_k"), ("to_v", "to_v"), ("to_out.0", "to_out.0"))
}


def dora_merged_weight(weight, lora_A, lora_B, magnitude=None, scaling=16 / 192):
    """weight [Cout, Cin, kh, kw] or [Cout, Cin]; lora_A [r, Cin, kh, kw] / [r, Cin]; lora_B [Cout, r, 1, 1] / [Cout, r];
    magnitude [Cout] (any shape with Cout elements) or None (plain LoRA).  Returns W' (fp32, weight's shape)."""
    w = weight.detach().float()
    from ..autograd_ops import low_rank_product
    delta = low_rank_product(lora_B.detach().flatten(1), lora_A.detach().flatten(1)).reshape(w.shape)
    merged = w + scaling * delta
    if magnitude is None:
        return merged
    norm = merged.flatten(1).norm(p=2, dim=1)                      # get_weight_norm: over (Cin, kh, kw) per output channel
    s = magnitude.detach().float().reshape(-1) / norm
    return merged * s.reshape(-1, *([1] * (w.dim() - 1)))


def _get(module, path):
    for p in path.split("."):
        module = module[int(p)] if p.isdigit() else getattr(module, p)
    return module


def extract_adapter(state_dict, target, adapter_name):
    """(lora_A, lora_B, magnitude | None) of one target layer from a peft-style state dict; keys
    ``<target>.lora_A.<adapter>.weight``, ``<target>.lora_B.<adapter>.weight``, ``<target>.lora_magnitude_vector.<adapter>[.weight]``."""
    a = state_dict.get(f"{target}.lora_A.{adapter_name}.weight")
    b = state_dict.get(f"{target}.lora_B.{adapter_name}.weight")
    if a is None or b is None:
        return None
    m = state_dict.get(f"{target}.lora_magnitude_vector.{adapter_name}.weight", state_dict.get(f"{target}.lora_magnitude_vector.{adapter_name}"))
    return a, b, m


@torch.no_grad()
def merge_unet_loras(unet, lora_state_dict, adapter_name="unet_distill", use_ffn_lora=True, use_attn_lora=False, lora_rank=None,
                     ffn_lora_alpha=16, attn_lora_scale_down=8, q_lora_updates_query=False):
    """Merge the named adapter into `unet` (this package's UNetModel) in place.  Returns {ldm_path: original weight} for
    `unmerge_unet_loras`.  Layers without an entry in the state dict are left untouched.  scaling = lora_alpha / rank with the
    reference's alphas (FFN: 16; attention: rank // 8, diffusers_attn_lora_capture.py:497-502, 541); the rank is read from
    lora_A unless given.  The q adapter of the attention layers is merged only with ``q_lora_updates_query``: by default
    (ddpm.py:134) it feeds the captured ``query2`` alone and leaves the attention output untouched (:239-249)."""
    saved = {}
    groups = []
    if use_ffn_lora:
        groups.append((FFN_LORA_TARGETS, lambda r: ffn_lora_alpha / r))
    if use_attn_lora:
        groups.append((ATTN_LORA_TARGETS, lambda r: (r // attn_lora_scale_down) / r))
    for targets, scale_of in groups:
        for dname, lpath in targets.items():
            if targets is ATTN_LORA_TARGETS and dname.endswith(".to_q") and not q_lora_updates_query:
                continue
            ad = extract_adapter(lora_state_dict, dname, adapter_name)
            if ad is None:
                continue
            layer = _get(unet, lpath)
            if not hasattr(layer, "weight"):          # nn.Identity skip connection has no conv_shortcut
                continue
            saved[lpath] = layer.weight.detach().clone()
            scaling = scale_of(lora_rank or ad[0].shape[0])
            w = dora_merged_weight(layer.weight, ad[0].to(layer.weight.device), ad[1].to(layer.weight.device),
                                   None if ad[2] is None else ad[2].to(layer.weight.device), scaling)
            layer.weight.copy_(w.to(layer.weight.dtype))
    return saved


@torch.no_grad()
def unmerge_unet_loras(unet, saved):
    for lpath, w in saved.items():
        _get(unet, lpath).weight.copy_(w)


def init_dora_adapter(weight, rank=192, generator=None):
    """peft's initial state for a layer: A ~ kaiming-uniform, B = 0, magnitude = ||W|| per output channel (so W' == W)."""
    w = weight.detach().float()
    a = torch.empty((rank,) + tuple(w.shape[1:]))
    bound = (6.0 / ((1 + 5) * a[0].numel())) ** 0.5              # kaiming_uniform_(a=sqrt(5))
    a.uniform_(-bound, bound, generator=generator)
    b = torch.zeros((w.shape[0], rank) + (1,) * (w.dim() - 2))
    return a, b, w.flatten(1).norm(dim=1)
