"""hipGraph capture of the fixed-shape segments of the training micro-batch.

A Stage-1 micro-batch issues ~4,100 kernel launches from Python (ctypes), ~10 us of host time each: the host, not the GPU, bounds the
step (rocprofv3: 55 ms of kernels in a 78 ms micro-batch, profiles/r02k_train_kernel_stats.txt).  The teacher's multi-step forward and
the student U-Net's forward and backward walks have fixed shapes per (batch, step count) and contain no host decision, so each is
captured ONCE per signature into a hipGraph (``torch.cuda.CUDAGraph``: capture on torch's stream, allocations from the graph's
private pool) and replayed afterwards: one host call instead of 400-1,300.

Protocol of a ``GraphedSegment``: call 1 with a new key runs eagerly (weights get packed, workspaces allocated); call 2 captures (the
inputs are copied into static buffers first, outputs and every tensor the segment saved live in the graph's pool); later calls copy
the inputs, replay, and return the SAME output tensors -- consumers must be done with them before the next replay of that key.  A consumer
that keeps them across other calls (the student U-Net's autograd node holds the saved activations until its backward) registers itself
with ``claim``; while it holds them, ``run`` of the same key executes eagerly instead of replaying (``busy``).  Anything that changes between replays must be visible through
fixed device addresses: parameters live in their (flat-arena) storage; derived weight packs are refreshed IN PLACE by the caller
before a replay (``refresh`` hook)."""
import atexit
import weakref

import torch

_LIVE = weakref.WeakSet()


def _release_all():
    """Drop every captured graph (and the tensors of its private pool) while the HIP runtime is still up: graph objects that survive
    until interpreter teardown are destroyed after it, which can fault."""
    segs = list(_LIVE)
    if not segs:
        return
    try:
        torch.cuda.synchronize()
    except Exception:                    # noqa: BLE001  (nothing to wait for if the runtime never came up)
        pass
    for s in segs:
        s.entries.clear()


atexit.register(_release_all)


class GraphedSegment:
    def __init__(self, name):
        self.name = name
        self.entries = {}          # key -> dict(state, graph, static_inputs, outputs, extra)
        self.enabled = True
        _LIVE.add(self)

    def busy(self, key) -> bool:
        """True while the tensors the last replay of ``key`` returned are still owned by a consumer that has not finished with them
        (``claim``).  A graph replay writes its outputs and saved activations into the SAME buffers every time, so a key must not be
        replayed while busy: ``run`` falls back to an eager call then (fresh tensors, same arithmetic)."""
        e = self.entries.get(key)
        if e is None or e.get("owner") is None:
            return False
        owner = e["owner"]()
        if owner is None or not getattr(owner, "af_holds_replay", False):
            e["owner"] = None
            return False
        return True

    def claim(self, key, owner):
        """``owner`` (any weak-referenceable object, e.g. the autograd ctx of the node that ran the replay) keeps the outputs of the
        last replay of ``key`` until it clears its ``af_holds_replay`` attribute or is garbage collected."""
        e = self.entries.get(key)
        if e is not None and e.get("state") == "graph":
            owner.af_holds_replay = True
            e["owner"] = weakref.ref(owner)

    def run(self, key, fn, inputs, refresh=None):
        """fn(*inputs) -> (outputs: tensor | tuple/list of tensors | nested, extra: any python object kept with the capture)."""
        if not self.enabled or self.busy(key):
            return fn(*inputs)
        e = self.entries.get(key)
        if e is None:
            self.entries[key] = {"state": "warm"}
            return fn(*inputs)                                   # eager: lazy initialisation happens here (and serves as the warm-up)
        if e["state"] == "warm":
            static = [None if t is None else t.detach().clone() for t in inputs]
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # no second warm-up run here: the eager call of this key already ran every lazy initialisation, and a segment that draws
            # noise / timesteps inside would consume extra generator draws (a use_graphs run must stay reproducible against eager)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = fn(*static)
            g.replay()                                           # capture only records: this is the launch that computes `out`
            e.update(state="graph", graph=g, static=static, out=out, owner=None)
            return out
        for dst, src in zip(e["static"], inputs):
            if dst is not None:
                dst.copy_(src)
        if refresh is not None:
            refresh()
        e["graph"].replay()
        return e["out"]

    def reset(self):
        self.entries.clear()
