"""CPU oracle (TEST INFRASTRUCTURE ONLY) for eval-mode DoRA layers as peft computes them -- the branch form, NOT the merged form the
product uses.  peft is a third-party dependency that is absent here and unpinned in the reference (requirements.txt:30): this restates
the published DoRA layer (``peft/tuners/lora/dora.py``: ``DoraLinearLayer.forward``, ``DoraConv2dLayer.forward``; weight norm over all
input dims per output channel, detached; result = base(x) + (s - 1) * op(x, W) + s * scaling * B(A(x))).  PARITY UNPINNED: no reference
test or runnable module holds these numbers (reference call sites: adaface/diffusers_attn_lora_capture.py:171-181, 547-556)."""
import torch
import torch.nn.functional as F


def dora_conv2d(x, weight, bias, lora_A, lora_B, magnitude, scaling, stride=1, padding=1):
    lora_weight = (lora_B.flatten(1) @ lora_A.flatten(1)).reshape(weight.shape)
    weight_norm = (weight + scaling * lora_weight).norm(p=2, dim=(1, 2, 3), keepdim=True).transpose(1, 0).detach()   # [1, Cout, 1, 1]
    s = magnitude.reshape(1, -1, 1, 1) / weight_norm
    base = F.conv2d(x, weight, bias, stride, padding)
    lora = F.conv2d(F.conv2d(x, lora_A, None, stride, padding), lora_B)
    return base + (s - 1) * F.conv2d(x, weight, None, stride, padding) + s * lora * scaling


def dora_linear(x, weight, bias, lora_A, lora_B, magnitude, scaling):
    weight_norm = torch.linalg.norm(weight + scaling * (lora_B @ lora_A), dim=1).detach()
    s = (magnitude / weight_norm).view(1, -1)
    base = F.linear(x, weight, bias)
    return base + (s - 1) * F.linear(x, weight) + s * F.linear(F.linear(x, lora_A), lora_B) * scaling


def dora_conv2d_train(x, weight, bias, lora_A, lora_B, magnitude, scaling, mask=None, stride=1, padding=1):
    """Training-mode branch form with an explicit dropout mask (values 0 or 1 / (1 - p)) on the adapter branch: peft's
    ``result = base_layer(x); x = dropout(x); result += dora(x)`` with dora(x) = (s - 1) * conv(x, W) + s * scaling * B(A(x))."""
    lora_weight = (lora_B.flatten(1) @ lora_A.flatten(1)).reshape(weight.shape)
    weight_norm = (weight + scaling * lora_weight.detach()).norm(p=2, dim=(1, 2, 3), keepdim=True).transpose(1, 0).detach()
    s = magnitude.reshape(1, -1, 1, 1) / weight_norm
    xd = x if mask is None else x * mask
    base = F.conv2d(x, weight, bias, stride, padding)
    return base + (s - 1) * F.conv2d(xd, weight, None, stride, padding) + s * F.conv2d(F.conv2d(xd, lora_A, None, stride, padding), lora_B) * scaling


def dora_linear_train(x, weight, bias, lora_A, lora_B, magnitude, scaling, mask=None):
    """Training-mode DoRA Linear (peft ``lora.Linear.forward`` with ``use_dora``: ``result = base(x); x = dropout(x);
    result += dora(x)``), explicit dropout mask; the weight norm is detached."""
    weight_norm = torch.linalg.norm(weight + scaling * (lora_B @ lora_A).detach(), dim=1).detach()
    s = (magnitude / weight_norm).view(1, -1)
    xd = x if mask is None else x * mask
    return F.linear(x, weight, bias) + (s - 1) * F.linear(xd, weight) + s * F.linear(F.linear(xd, lora_A), lora_B) * scaling
