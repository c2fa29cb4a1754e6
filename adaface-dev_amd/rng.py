"""Counter-based synthetic weights and inputs.

There is no network for SD-1.5 checkpoints, so benchmarks, fixtures and tests use
seeded random weights.  Every tensor is drawn from its own Philox stream keyed by
(seed, crc32(tensor name)), so the golden-fixture generator (which instantiates the
*reference* module tree in the build container) and the GPU box (which instantiates
this package's module tree) obtain bit-identical weights from names + shapes alone,
without shipping gigabytes.

The reference zero-initialises 39 tensors (``zero_module``: ResBlock ``out_layers.3``,
SpatialTransformer ``proj_out``, final ``out.2`` -- openaimodel.py:230-232,689,
attention.py:280); a random-weight network with those left at zero outputs exactly 0,
so they are drawn like every other tensor here (SURVEY.md 8c caveat 1).
"""
import contextlib
import zlib

import numpy as np
import torch


def _stream(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


# Optional memo of drawn parameters (tests only: tests/conftest.py switches it on).  The GPU suite builds the same reduced-width model
# stack dozens of times from the same (seed, name, shape) triples; on the GPU box's host a draw costs ~2 ms, 12 s per trainer set-up.
# Entries are handed out as they are (callers copy out of them: load_synth_weights) -- synth_state_dict clones.
_memo = None
_MEMO_MAX_ELEMS = 1 << 22          # per tensor; full-size weights are not kept
_MEMO_MAX_TOTAL = 1 << 29          # elements in total (2 GiB of fp32)
_memo_total = 0


def enable_memo(on: bool = True) -> None:
    global _memo, _memo_total
    _memo, _memo_total = ({} if on else None), 0


def synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """One synthetic fp32 parameter. Rule is decided by the name suffix and rank."""
    global _memo_total
    shape = tuple(int(s) for s in shape)
    if _memo is not None:
        hit = _memo.get((name, shape, int(seed)))
        if hit is not None:
            return hit
        t = _synth_tensor(name, shape, seed)
        if t.numel() <= _MEMO_MAX_ELEMS and _memo_total + t.numel() <= _MEMO_MAX_TOTAL:
            _memo[(name, shape, int(seed))] = t
            _memo_total += t.numel()
        return t
    return _synth_tensor(name, shape, seed)


def _synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    g = _stream(seed, name)
    z = g.standard_normal(size=shape, dtype=np.float32)
    if name.endswith(".bias"):
        z *= 0.05
    elif len(shape) == 1:  # GroupNorm / LayerNorm gamma
        z = 1.0 + 0.1 * z
    else:  # conv / linear weight: unit-gain fan-in scaling
        fan_in = int(np.prod(shape[1:]))
        z *= 1.0 / np.sqrt(fan_in)
    return torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32))


def synth_state_dict(named_shapes, seed: int = 0):
    """named_shapes: iterable of (name, shape). Returns {name: fp32 tensor}."""
    return {name: (synth_tensor(name, shape, seed).clone() if _memo is not None else synth_tensor(name, shape, seed)) for name, shape in named_shapes}


def load_synth_weights(module: torch.nn.Module, seed: int = 0, on_device: bool = False) -> None:
    """Overwrite every parameter of `module` in place with its synthetic value.  on_device: draw the values with torch's generator
    on the parameter's own (GPU) device instead -- the same rules and scales, NOT the Philox values the fixtures are made from:
    for benchmarks only, where 2.6 G numbers per process from a single-threaded host generator would be most of the run time."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not on_device:
                p.copy_(synth_tensor(name, p.shape, seed).to(p.dtype))
                continue
            g = torch.Generator(device=p.device)
            g.manual_seed(((int(seed) + 1) * 0x9E3779B1 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF)
            z = torch.randn(p.shape, device=p.device, dtype=torch.float32, generator=g)
            if name.endswith(".bias"):
                z *= 0.05
            elif p.dim() == 1:
                z = 1.0 + 0.1 * z
            else:
                z *= float(np.prod(p.shape[1:])) ** -0.5
            p.copy_(z.to(p.dtype))


@contextlib.contextmanager
def skip_default_init():
    """Construct modules without torch's default random initialisation (for callers that overwrite EVERY parameter right after:
    the default init of an 860 M-parameter U-Net is ~15 s of host time per process)."""
    names = ("kaiming_uniform_", "kaiming_normal_", "uniform_", "normal_", "trunc_normal_", "xavier_uniform_", "xavier_normal_")
    saved = {n: getattr(torch.nn.init, n) for n in names}
    try:
        for n in names:
            setattr(torch.nn.init, n, lambda t, *a, **k: t)
        yield
    finally:
        for n, f in saved.items():
            setattr(torch.nn.init, n, f)


def synth_input(name: str, shape, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    """Synthetic N(0, scale^2) activation-like input (latents, context, ID vectors)."""
    g = _stream(seed, "input:" + name)
    z = g.standard_normal(size=tuple(int(s) for s in shape), dtype=np.float32) * scale
    return torch.from_numpy(z.astype(np.float32))


def synth_face_state_dict(state_dict, seed: int = 0):
    """Synthetic values for every entry of a ResNetFace state dict (parameters AND BatchNorm buffers) from names + shapes:
    running_var > 0, PReLU slopes around 0.25, everything else by `synth_tensor`'s rules."""
    out = {}
    for name, t in state_dict.items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=t.dtype)
            continue
        z = synth_tensor(name, shape if shape else (1,), seed)
        if name.endswith("running_var"):
            z = 0.6 + 0.4 * z.abs()
        elif name.endswith("running_mean"):
            z = (z - 1.0)                                    # 1-D rule gives 1 + 0.1 z -> 0.1 z
        elif t.numel() == 1:                                  # nn.PReLU() slope
            z = 0.25 + 0.5 * (z - 1.0)
        out[name] = z.reshape(shape).to(torch.float32)
    return out
