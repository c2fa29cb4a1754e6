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
            z = torch.tensor(0.0, device=noise_pred.device)
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
