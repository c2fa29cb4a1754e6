"""Hot-path slice of the reference's ``adaface/util.py``: ``perturb_tensor`` (:30-53) and the weighted U-Net
ensemble (:174-248).  Small host-level tensor glue."""
import torch
import torch.nn as nn


def perturb_tensor(ts, perturb_std, perturb_std_is_relative=True, keep_norm=False, std_dim=-1, norm_dim=-1, verbose=False):
    if perturb_std_is_relative:
        perturb_std = perturb_std * ts.std(dim=std_dim).mean().detach()
    noise = torch.randn_like(ts) * perturb_std
    if keep_norm:
        orig_norm = ts.norm(dim=norm_dim, keepdim=True)
        ts = ts + noise
        ts = ts * orig_norm / (ts.norm(dim=norm_dim, keepdim=True).detach() + 1e-8)
    else:
        ts = ts + noise
    return ts


class UNetEnsemble(nn.Module):
    """Weighted sum of several U-Nets' outputs on one device (reference adaface/util.py:174-248)."""

    def __init__(self, unets, unet_weights=None):
        super().__init__()
        self.unets = nn.ModuleList(unets)
        w = torch.ones(len(unets)) if unet_weights is None else torch.as_tensor(unet_weights, dtype=torch.float32)
        self.register_buffer("unet_weights", w / w.sum())

    def forward(self, x, timesteps, contexts, extra_info=None):
        out = None
        for unet, w, ctx in zip(self.unets, self.unet_weights, contexts):
            e = unet(x, timesteps, ctx, extra_info=extra_info) * w
            out = e if out is None else out + e
        return out
