"""Data-parallel gradient exchange for the distillation step: one process per GPU,
``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm), gradient all-reduce (mean) of the trainable
parameters once per optimizer step, bucketed and overlapped with the remaining backward.

Replaces what Lightning's ``strategy="ddp"`` (reference main.py:618) does implicitly, designed for this path:

* gradients already live in ONE flat fp32 arena per parameter group (``ldm.c_adamw.FlatArena``), so a bucket is
  a contiguous slice of that arena: no flatten/unflatten copies, the all-reduce runs in place;
* buckets are sized for xGMI's point-to-point links (7 x ~153 GB/s per GPU: a ring all-reduce is per-link bound):
  default 32 MB -- 0.48 GB of gradients = 15 collectives of ~0.2 ms each at ring bandwidth, large enough to be
  bandwidth- rather than latency-bound, small enough that the first one starts early in the backward;
* a bucket's all-reduce is launched (async, on RCCL's own stream) from the post-accumulate-grad hook of the LAST
  of its parameters to become ready.  What that overlaps with, precisely: the trainable set sits at the END of the backward --
  the U-Net's activation-gradient walk is one autograd node, its FFN-adapter gradients leave it only when it finishes, and the
  85 M SubjBasisGenerator weights receive theirs while the text encoder / generator layers behind it are being walked -- so the
  collectives hide under that encoder tail (a few ms) and the rest is waited for in ``finish()``.  At 0.36 GB per optimizer step
  (~2.5 ms of ring time at xGMI bandwidth, once per two micro-batches of ~60 ms) the exposed part is small either way;
* collectives are ISSUED IN BUCKET-INDEX ORDER on every rank (bucket b only after buckets 0..b-1; buckets are numbered
  from the end of the arena, the order gradients become ready in): ranks whose graphs differ for one iteration
  (per-rank RNG flags, unused parameters) still pair the same buffers, the rest is flushed in order by ``finish()``;
* ``no_sync()`` skips the exchange on the non-final micro-batches of gradient accumulation
  (``accumulate_grad_batches: 2`` in the reference yaml) -- the arena keeps accumulating.

Works on any torch.distributed backend (tests run it with gloo / world_size 2 on CPU).
"""
import contextlib
from typing import List

import torch.distributed as dist


class GradReducer:
    def __init__(self, arenas: List, bucket_bytes: int = 32 << 20, process_group=None, broadcast_params: bool = True,
                 reduce_single_rank: bool = False):
        """reduce_single_rank: run the collectives even when the group has ONE rank (they are identities there) -- lets a
        single-GPU box exercise the RCCL path end to end."""
        self.arenas = list(arenas)
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        self.active = self.world > 1 or (reduce_single_rank and dist.is_initialized())
        self.sync = True
        self.handles = []
        self.buckets = []          # (arena, lo, hi, n_params)
        self._pending = {}
        self._hooks = []
        self.launch_log = []       # bucket indices in the order their collectives were issued (last backward)
        if self.active and broadcast_params:
            for a in self.arenas:                      # same start point on every rank (DDP's initial broadcast)
                dist.broadcast(a.flat_p, src=0, group=self.pg)
                if hasattr(a, "bump_generation"):
                    a.bump_generation()                # the arena was written outside autograd: cached weight packs are stale
        # Buckets are numbered in GLOBAL readiness order, because collectives are issued strictly by bucket index: the backward reaches the
        # LAST arena first (the U-Net's adapters sit behind the encoder's parameters in the forward, so their gradients exist while the encoder
        # layers are still running), and inside an arena gradients become ready roughly in reverse parameter order.
        for a in reversed(self.arenas):
            per = max(1, bucket_bytes // 4)
            idx = len(a.params) - 1
            while idx >= 0:
                hi = a.offsets[idx + 1]
                j = idx
                while j > 0 and hi - a.offsets[j - 1] <= per:
                    j -= 1
                lo = a.offsets[j]
                b = len(self.buckets)
                self.buckets.append((a, lo, hi, idx - j + 1))
                for k in range(j, idx + 1):
                    self._hooks.append(a.params[k].register_post_accumulate_grad_hook(self._make_hook(b)))
                idx = j - 1
        self._reset()

    def _reset(self):
        self._pending = {b: n for b, (_, _, _, n) in enumerate(self.buckets)}
        self._next = 0             # next bucket index to issue

    def _launch_ready(self, flush: bool = False):
        """Issue, in index order, every bucket whose parameters all have their gradient (all remaining ones when flushing)."""
        while self._next < len(self.buckets) and (flush or self._pending[self._next] == 0):
            a, lo, hi, _ = self.buckets[self._next]
            self.handles.append(dist.all_reduce(a.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            self.launch_log.append(self._next)
            self._next += 1

    def _make_hook(self, b):
        def hook(param):
            if not self.sync or not self.active:
                return
            if self._next == 0 and not self.handles:
                self.launch_log = []
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch_ready()
        return hook

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: no exchange inside this context."""
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    def finish(self):
        """Wait for the in-flight bucket reductions of this backward and turn sums into means.  Buckets whose
        parameters received no gradient in this backward (unused parameters) are reduced here, so every rank
        issues the same collectives in the same order."""
        if not self.active:
            self._reset()
            return
        self._launch_ready(flush=True)
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if self.world > 1:
            for a in self.arenas:
                a.flat_g.div_(self.world)
        self._reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
