"""Host-side mirror of the reference's ``ldm/modules/diffusionmodules/util.py`` for the hot path.

Same public names and argument meaning (``timestep_embedding``, ``normalization`` /
``GroupNorm32``, ``conv_nd``, ``linear``, ``zero_module``, ``checkpoint``,
``make_beta_schedule``, ``make_ddim_timesteps``, ``make_ddim_sampling_parameters``,
``extract_into_tensor``), but the layers execute on the MI355X through ``libadaface_hip.so``:

* every leaf layer keeps the reference's parameter names / shapes (so SD-1.5 ``.ckpt`` keys
  load with ``load_state_dict``) and adds a ``hip(...)`` method that works on fp16
  channels-last activations; ``forward(...)`` is the reference-compatible entry
  (NCHW in, same dtype out) built on top of it;
* weights are re-laid once per device into the K-contiguous fp16 matrices ``af_gemm`` consumes
  (``packed()``), and re-packed automatically when a parameter is modified or moved.
"""

import numpy as np
import torch
import torch.nn as nn

from .... import ops
from ....ops import F16


# ----------------------------------------------------------------------------- layout helpers
def to_nhwc_f16(x: torch.Tensor, cpad: int = 0) -> torch.Tensor:
    """Logical NCHW tensor of any float dtype -> contiguous fp16 [B, H, W, C(+pad)]."""
    if x.dtype == F16 and cpad <= x.shape[1] and x.is_contiguous(memory_format=torch.channels_last):
        return x.permute(0, 2, 3, 1)
    return ops.nchw_f32_to_nhwc_f16(x, cpad)


def from_nhwc_f16(y: torch.Tensor, dtype: torch.dtype, channels: int = 0) -> torch.Tensor:
    """fp16 [B, H, W, C] -> logical NCHW in `dtype` (fp16: zero-copy channels-last view)."""
    channels = channels or y.shape[-1]
    if dtype == F16 and channels == y.shape[-1]:
        return y.permute(0, 3, 1, 2)
    out = ops.nhwc_f16_to_nchw_f32(y, channels)
    return out if dtype == torch.float32 else out.to(dtype)


class _PackCache:
    """Caches a derived device tensor bundle until one of the source parameters changes."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, params, build):
        # ~4,300 calls per Stage-2 micro-batch: compare against the remembered keys one by one (a tuple built through a generator cost 4 us a call)
        key = self._key
        pk = ops.param_key
        if key is not None:
            i = 0
            for p in params:
                if p is not None:
                    if i >= len(key):
                        break
                    k = key[i]                                   # ops.param_key(p) == k, field by field (version first: what changes)
                    gen = p.__dict__.get("_af_gen")
                    if p._version != k[1] or p.data_ptr() != k[0] or (gen[0] if gen is not None else -1) != k[3] or p.device != k[2]:
                        break
                    i += 1
            else:
                if i == len(key):
                    return self._val
        self._val = build()
        self._key = tuple([pk(p) for p in params if p is not None])
        return self._val


def _require_cuda(p: torch.Tensor, what: str):
    if not p.is_cuda:
        raise RuntimeError(f"{what}: parameters are on {p.device}; this layer only runs on an MI355X "
                           "(HIP extension, no CPU fallback). Move the module with .cuda() first.")


# ----------------------------------------------------------------------------- leaf layers
class Conv2d(nn.Conv2d):
    """nn.Conv2d (3x3 pad 1, stride 1|2, or 1x1) executed by af_gemm as (implicit) GEMM."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        ks, st, pd = self.kernel_size, self.stride, self.padding
        if not ((ks == (3, 3) and pd == (1, 1) and st in ((1, 1), (2, 2))) or (ks == (1, 1) and pd == (0, 0) and st == (1, 1))):
            raise NotImplementedError(f"Conv2d kernel={ks} stride={st} padding={pd}: only 3x3/pad1/stride1|2 and 1x1 are on the hot path")
        self._cache = _PackCache()
        self.cin_pad = 0  # set >0 to accept channel-padded NHWC input (first conv: 4 -> 8)

    def packed(self) -> ops.PackedWeight:
        _require_cuda(self.weight, "Conv2d")

        def build():
            if self.kernel_size == (3, 3):
                return ops.pack_conv3x3(self.weight, self.bias, self.weight.device, self.cin_pad)
            return ops.pack_matrix(self.weight.detach().reshape(self.out_channels, self.in_channels), self.bias, self.weight.device)

        return self._cache.get((self.weight, self.bias), build)

    def hip(self, x, x2=None, upsample=False, rowbias=None, residual=None, gn_groups=0, defer_gn=False):
        """x [B,H,W,C1] (+x2 [B,H,W,C2]) fp16 -> [B,Ho,Wo,Cout] fp16.  gn_groups > 0: the output feeds a GroupNorm of that many groups -- where
        the launch can, it leaves the partial statistics with the tensor (ops.GnPartials) and the GroupNorm skips its statistics pass.
        defer_gn (3x3 only): the caller's NEXT launch is that GroupNorm on the returned tensor -- a split-K launch then leaves its reduce pass to
        it (ops.PendingReduce: the tensor is not written until then)."""
        pw = self.packed()
        cpg = self.out_channels // gn_groups if gn_groups and self.out_channels % gn_groups == 0 else 0
        if self.kernel_size == (3, 3):
            return ops.conv3x3(x, pw, x2=x2, stride=self.stride[0], upsample=upsample, rowbias=rowbias, residual=residual, gn_cpg=cpg,
                               defer_gn=defer_gn and cpg > 0)
        B, H, W, c1 = x.shape
        a2 = None if x2 is None else x2.reshape(B * H * W, x2.shape[-1])
        res = None if residual is None else residual.reshape(B * H * W, -1)
        out = ops.gemm(x.reshape(B * H * W, c1), pw, a2=a2, rowbias=rowbias, rows_per_batch=H * W, residual=res, gn_cpg=cpg)
        out4 = out.reshape(B, H, W, -1)
        if hasattr(out, "_gn_partials"):
            out4._gn_partials = out._gn_partials      # a view: same storage address and version counter (ops.partials_of)
        return out4

    def packed_bwd(self) -> ops.PackedWeight:
        """Weights of the input-gradient convolution: spatially flipped, in/out channels swapped
        (3x3) or the plain transpose (1x1).  No bias."""
        _require_cuda(self.weight, "Conv2d")

        def build():
            w = self.weight.detach()
            if self.kernel_size == (3, 3):
                wd = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()       # [Cin, Cout, 3, 3]
                return ops.pack_conv3x3(wd, None, w.device, ops.round_up(self.out_channels, 8))
            return ops.pack_matrix(w.reshape(self.out_channels, self.in_channels).t().contiguous(), None, w.device)

        if not hasattr(self, "_cache_bwd"):
            self._cache_bwd = _PackCache()
        return self._cache_bwd.get((self.weight,), build)

    # layer protocol used by TimestepEmbedSequential.hip_train / hip_bwd (bare conv = first input block)
    def hip_train(self, x):
        return self.hip(x), (x.shape[1], x.shape[2])

    def hip_bwd(self, saved, dy):
        return self.hip_dgrad(dy, in_hw=saved)

    def hip_dgrad(self, dy, in_hw=None, upsampled=False):
        """dy [B,Ho,Wo,Cout] -> dx [B,H,W,Cin].  in_hw: forward input size (needed for stride 2);
        upsampled: the forward ran on the nearest-x2 upsampled input (Upsample) -> 2x2 block sums."""
        pw = self.packed_bwd()
        if self.kernel_size == (3, 3):
            if self.stride[0] == 2:
                return ops.conv3x3(dy, pw, upsample=2, out_hw=in_hw)
            dx = ops.conv3x3(dy, pw)
            return ops.sumpool2x2(dx) if upsampled else dx
        B, H, W, c = dy.shape
        return ops.gemm(dy.reshape(B * H * W, c), pw).reshape(B, H, W, -1)

    def forward(self, x):
        y = self.hip(to_nhwc_f16(x, self.cin_pad))
        return from_nhwc_f16(y, x.dtype)


class Linear(nn.Linear):
    """nn.Linear executed by af_gemm.  Input [..., K] -> [..., N]."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._cache = _PackCache()

    def packed(self) -> ops.PackedWeight:
        _require_cuda(self.weight, "Linear")
        return self._cache.get((self.weight, self.bias), lambda: ops.pack_matrix(self.weight, self.bias, self.weight.device))

    def hip(self, x2d, residual=None, act=ops.AF_ACT_NONE, a2=None):
        return ops.gemm(x2d, self.packed(), residual=residual, act=act, a2=a2)

    def packed_bwd(self) -> ops.PackedWeight:
        _require_cuda(self.weight, "Linear")
        if not hasattr(self, "_cache_bwd"):
            self._cache_bwd = _PackCache()
        return self._cache_bwd.get((self.weight,), lambda: ops.pack_matrix(self.weight.detach().t().contiguous(), None, self.weight.device))

    def hip_dgrad(self, dy2d, residual=None):
        """dy [M, N] -> dx [M, K] (+ residual)."""
        return ops.gemm(dy2d, self.packed_bwd(), residual=residual)

    def forward(self, x):
        x2d = x.reshape(-1, x.shape[-1]).to(F16).contiguous()
        y = self.hip(x2d).reshape(*x.shape[:-1], self.out_features)
        return y if x.dtype == F16 else y.to(x.dtype)


class GroupNorm32(nn.GroupNorm):
    """GroupNorm computed with fp32 statistics (reference util.py:210-212), optionally fused with SiLU."""

    def hip(self, x, silu=False, x2=None):
        _require_cuda(self.weight, "GroupNorm32")
        return ops.groupnorm(x, self.weight, self.bias, self.eps, silu, x2=x2, groups=self.num_groups)

    def hip_train(self, x, silu=False, x2=None):
        """-> (y, stats) with stats = per-(batch, group) (mean, rstd) kept for hip_bwd."""
        _require_cuda(self.weight, "GroupNorm32")
        return ops.groupnorm_train(x, self.weight, self.bias, self.eps, silu, x2=x2, groups=self.num_groups)

    def hip_bwd(self, x, stats, dy, silu=False, x2=None, add=None):
        return ops.groupnorm_bwd(x, self.weight, self.bias, stats, dy, silu, x2=x2, add=add, groups=self.num_groups)

    def forward(self, x):
        return from_nhwc_f16(self.hip(to_nhwc_f16(x)), x.dtype)


class LayerNorm(nn.LayerNorm):
    def hip(self, x2d):
        _require_cuda(self.weight, "LayerNorm")
        return ops.layernorm(x2d, self.weight, self.bias, self.eps)

    def hip_bwd(self, x2d, dy, add=None):
        return ops.layernorm_bwd(x2d, self.weight, dy, self.eps, add=add)

    def forward(self, x):
        y = self.hip(x.reshape(-1, x.shape[-1]).to(F16).contiguous()).reshape(x.shape)
        return y if x.dtype == F16 else y.to(x.dtype)


class SiLU(nn.Module):
    def forward(self, x):
        y = ops.silu(x.to(F16).contiguous())
        return y if x.dtype == F16 else y.to(x.dtype)


def normalization(channels):
    """32-group GroupNorm, eps 1e-5 (reference util.py:195-201)."""
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the SD-1.5 hot path is 2-D)")
    return Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return Linear(*args, **kwargs)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def checkpoint(func, inputs, params, flag):
    """Pass-through, as in the reference where checkpointing is hard-disabled (util.py:115)."""
    return func(*inputs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[cos | sin] sinusoidal embedding (reference util.py:154-174) -> fp16 [N, dim] on the device."""
    if repeat_only:
        return timesteps[:, None].to(F16).repeat(1, dim)
    return ops.timestep_embedding(timesteps, dim, float(max_period))


# ----------------------------------------------------------------------------- schedules (host, numpy)
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """Reference util.py:21-43; only "linear" (SD) and its siblings that need no torch."""
    if schedule == "linear":
        # torch.linspace (not np.linspace): bit-identical fp64 table to the reference's util.py:23-25
        return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    raise ValueError(f"schedule '{schedule}' unknown.")


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphacums: fp32 numpy table.  Returns (sigmas, alphas, alphas_prev) as fp32 numpy arrays."""
    alphacums = np.asarray(alphacums, dtype=np.float32)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist(), dtype=np.float32)
    sigmas = (eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))).astype(np.float32)
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))
