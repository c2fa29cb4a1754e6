"""Cautious AdamW with the reference's interface (``ldm/c_adamw.py``: ``CAdamW(params, lr, betas, eps,
weight_decay, correct_bias)``), executed as ONE fused HIP launch pair per parameter group.

MI355X-first layout: every parameter group lives in a *flat fp32 arena* -- one contiguous buffer for the
parameters and one for the gradients, with the ``nn.Parameter``s (and their ``.grad``) re-pointed to views
of it.  The optimizer kernel streams the arena once (HBM-bound: 4 reads + 3 writes per element); the
data-parallel gradient all-reduce (``adaface_dev_amd.distributed.GradReducer``) reduces contiguous slices
of the same gradient arena over RCCL, so no per-tensor launches and no flatten/unflatten copies exist.
The caution mask ``exp_avg * grad > 0`` is renormalised per parameter TENSOR (segment), exactly as the
reference does per ``p`` (c_adamw.py:117-121)."""
from typing import Iterable

import torch
from torch.optim import Optimizer

from .. import ops


class FlatArena:
    """Flat fp32 parameter / gradient storage for a list of parameters (all on one CUDA device)."""

    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, "empty parameter list"
        dev = self.params[0].device
        offs = [0]
        for p in self.params:
            assert p.device == dev and p.dtype == torch.float32, "arena needs fp32 parameters on one device"
            offs.append(offs[-1] + p.numel())    # exact extents: the caution mask mean is per tensor
        self.numel = offs[-1]
        self.flat_p = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.offsets = offs
        # generation of the parameter values: bumped whenever the arena is written outside autograd (the fused optimizer
        # kernel, the data-parallel start-up broadcast); every derived-weight cache keys on it through ops.param_key
        self.gen = [0]
        with torch.no_grad():
            for p, o in zip(self.params, offs[:-1]):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
                p._af_gen = self.gen

    def bump_generation(self):
        self.gen[0] += 1

    def zero_grad(self):
        self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets[:-1]):            # re-attach in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)


class AdamW(Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0,
                 correct_bias: bool = True):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr} - should be >= 0.0")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameters: {betas} - should be in [0.0, 1.0)")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps} - should be >= 0.0")
        defaults = {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "correct_bias": correct_bias}
        super().__init__(params, defaults)
        self.init_lr = lr
        self._arenas = {}

    def arena(self, gi: int) -> FlatArena:
        """The flat arena of parameter group `gi` (built on first use; parameters must already be on the GPU)."""
        if gi not in self._arenas:
            group = self.param_groups[gi]
            a = FlatArena(group["params"])
            dev = a.flat_p.device
            a.seg_offsets = torch.tensor(a.offsets, dtype=torch.int64, device=dev)
            a.counts = torch.zeros(len(a.params), dtype=torch.int32, device=dev)
            a.exp_avg = torch.zeros_like(a.flat_p)
            a.exp_avg_sq = torch.zeros_like(a.flat_p)
            a.step = 0
            self._arenas[gi] = a
        return self._arenas[gi]

    def zero_grad(self, set_to_none: bool = False):
        for gi in range(len(self.param_groups)):
            self.arena(gi).zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            a = self.arena(gi)
            a.step += 1
            ops.cadamw_step(a.flat_p, a.flat_g, a.exp_avg, a.exp_avg_sq, a.seg_offsets, a.counts, lr=group["lr"],
                            betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"], step=a.step,
                            correct_bias=group["correct_bias"])
            a.bump_generation()          # raw write of flat_p: p._version cannot see it, the fp16 weight packs must be rebuilt
        return loss


CAdamW = AdamW   # the reference's class is ``ldm.c_adamw.AdamW`` (c_adamw.py:13); "CAdamW" is how ddpm.py refers to it
