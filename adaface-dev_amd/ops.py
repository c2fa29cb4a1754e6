"""Tensor-level wrappers over the C ABI (include/adaface_hip.h).

torch is plumbing here: it owns device memory and the current HIP stream; every function
below hands raw device pointers to ``libadaface_hip.so`` and raises ``RuntimeError`` on any
failure.  There is no eager / CPU fallback.

Activation convention: fp16, channels-last.  A feature map is a contiguous tensor
``[B, H, W, C]`` (equivalently tokens ``[B, H*W, C]``).

Concurrency: launches go to ``torch.cuda.current_stream()``.  The caller-owned scratch the C ABI asks for (GroupNorm partials,
split-K slabs, attention-backward copies, the zero page) is cached here PER DEVICE, not per stream: the hot path is
single-stream by design (one process per GPU, one HIP stream, the hipGraph replays on it), so two streams of one process must
not run these wrappers concurrently.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import AF_ACT_GEGLU, AF_ACT_NONE, AF_ACT_QUICKGELU, AF_ACT_SILU, AF_OUT_F32, AF_OUT_NORMAL, AF_OUT_SPLIT_T, GemmDesc

F16 = torch.float16
NEG_MAX = -torch.finfo(torch.float32).max



def param_key(p):
    """Identity + modification state of a parameter, the key of every derived-weight cache (fp16 packs, merged adapters).
    `_version` only sees autograd-visible in-place writes; parameters that live in a flat optimizer arena
    (ldm.c_adamw.FlatArena) are rewritten by a raw HIP kernel / an in-place collective, so the arena hands each of them its
    generation counter (`p._af_gen`, a shared one-element list) and bumps it on every such write."""
    if p is None:
        return None
    gen = p.__dict__.get("_af_gen")          # (a failed getattr on a tensor costs as much as the rest of this function: ~14,000 calls per Stage-2 micro-batch)
    return (p.data_ptr(), p._version, p.device, gen[0] if gen is not None else -1)


def _stream() -> int:
    # the raw handle of torch's current stream on the current device: two C calls (~0.3 us) instead of the ~3 us of
    # torch.cuda.current_stream().cuda_stream -- this runs once per launch, ~11,000 times in a Stage-2 micro-batch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk_f16(t: torch.Tensor, name: str):
    if _pending:                                  # a split-K output whose reduce pass was left to its GroupNorm is about to be read by something else
        pr = _pending.get(t.device)
        if pr is not None and pr.ptr == t.data_ptr():
            flush_pending(t.device)
    _chk_f16_raw(t, name)


def _chk_f16_raw(t: torch.Tensor, name: str):
    if t.dtype != F16 or not t.is_cuda or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous fp16 device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- weights
@dataclass
class PackedWeight:
    """fp16 [Npad, Kpad] K-contiguous weight (+ fp32 bias) laid out for af_gemm."""
    wt: torch.Tensor
    bias: Optional[torch.Tensor]
    N: int
    K: int
    kpad: int
    taps: int = 1
    cin: int = 0  # channels per tap as seen by the kernel (after any channel padding)
    ln_cs: Optional[torch.Tensor] = None   # folded LayerNorm (pack_matrix_ln): fp32 [Npad] column sums of the fp16 rows
    ln_eps: float = 0.0
    k_tail: int = 0   # pack_conv3x3_skip: plain K columns behind the nine tap blocks (the ResBlock's 1x1 shortcut inside its second convolution)
    aliased: bool = False   # wt IS the caller's tensor (pack_matrix fast path): read-only -- in-place refresh paths must replace, never copy_ into it


def pack_matrix(w2d: torch.Tensor, bias: Optional[torch.Tensor], device, taps: int = 1, cin: int = 0) -> PackedWeight:
    """w2d: [N, K] (any float dtype, any device) -> zero-padded fp16 [roundup(N,128), roundup(K,64)]."""
    N, K = w2d.shape
    npad, kpad = round_up(N, 128), round_up(K, 64)
    if ((npad, kpad) == (N, K) and w2d.dtype == F16 and w2d.is_contiguous() and w2d.device == torch.device(device) and not w2d.requires_grad
            and not isinstance(w2d, torch.nn.Parameter)):
        # already in the packed layout (an fp16 activation used as the B operand of a product: the feature-matching losses' 4096-row matrices):
        # no zero-fill, no copy.  The pack ALIASES the caller's tensor and is marked so (read-only: refresh-in-place code checks `aliased`);
        # parameters never take this path, frozen fp16 ones included
        b = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        return PackedWeight(w2d, b, N, K, kpad, taps, cin if cin else K, aliased=True)
    wt = torch.zeros((npad, kpad), dtype=F16, device=device)
    wt[:N, :K] = w2d.detach().to(device=device, dtype=F16)
    b = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
    return PackedWeight(wt, b, N, K, kpad, taps, cin if cin else K)


def pack_matrix_ln(w2d: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float, device) -> PackedWeight:
    """Weight of a Linear that consumes LayerNorm(x) (BasicTransformerBlock.norm1/2/3 -> attn1 q|k|v, attn2.to_q, ff GEGLU projection;
    reference attention.py:242-252), packed so that af_gemm can take the UN-normalised rows:
        LN(x) W^T + b = rstd * (x (gamma * W)^T - mean * colsum(gamma * W)) + (b + W beta).
    The kernel accumulates mean / rstd of each row from the A fragments of its own main loop (af_gemm_desc.ln_colsum).  Column sums are
    taken over the fp16-rounded packed rows, so the rank-one correction cancels the mean term exactly as the MFMAs saw it."""
    w = w2d.detach().to(device=device, dtype=torch.float32)
    g, bt = gamma.detach().to(device=device, dtype=torch.float32), beta.detach().to(device=device, dtype=torch.float32)
    b = (w * bt[None, :]).sum(dim=1)                       # W beta (element-wise: no vendor gemv for a one-off pack)
    if bias is not None:
        b = b + bias.detach().to(device=device, dtype=torch.float32)
    pw = pack_matrix(w * g[None, :], b, device)
    pw.ln_cs = pw.wt.float().sum(dim=1).contiguous()
    pw.ln_eps = float(eps)
    return pw


def pack_conv3x3(w: torch.Tensor, bias: Optional[torch.Tensor], device, cin_pad: int = 0) -> PackedWeight:
    """[Cout, Cin, 3, 3] -> [Cout, 9*Cin'] with K order (ky, kx, cin); Cin' = max(Cin, cin_pad)."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    w = w.detach().permute(0, 2, 3, 1)  # [Cout, 3, 3, Cin]
    cinp = max(cin, cin_pad)
    if cinp != cin:
        w = torch.nn.functional.pad(w, (0, cinp - cin))
    return pack_matrix(w.reshape(cout, 9 * cinp), bias, device, taps=9, cin=cinp)


def pack_conv3x3_skip(w3: torch.Tensor, b3: Optional[torch.Tensor], w1: torch.Tensor, b1: Optional[torch.Tensor], device) -> PackedWeight:
    """out = conv3x3(h; w3, b3) + conv1x1(x; w1, b1) as ONE K-concatenated implicit GEMM (af_gemm_desc.a3 / a4): [Cout, 9 Cin | Cs] with the 3x3
    block in (ky, kx, cin) order and the 1x1 weights behind it; the biases add up.  The ResBlock's out_layers convolution + skip_connection
    (openaimodel.py:256-276).  Cin and Cs must be multiples of 64 (every SD-1.5 block is)."""
    cout, cin, kh, kw = w3.shape
    assert kh == 3 and kw == 3 and w1.shape[0] == cout and w1.shape[2:] == (1, 1)
    cs = w1.shape[1]
    assert cin % 64 == 0 and cs % 64 == 0, "pack_conv3x3_skip: channel counts must be multiples of 64"
    w = torch.cat([w3.detach().permute(0, 2, 3, 1).reshape(cout, 9 * cin).float(), w1.detach().reshape(cout, cs).float().to(w3.device)], dim=1)
    b = None
    if b3 is not None or b1 is not None:
        b = torch.zeros(cout, dtype=torch.float32, device=w3.device)
        if b3 is not None:
            b = b + b3.detach().float()
        if b1 is not None:
            b = b + b1.detach().float().to(b.device)
    pw = pack_matrix(w, b, device, taps=9, cin=cin)
    pw.k_tail = cs
    return pw


def interleave_geglu(w: torch.Tensor, b: torch.Tensor):
    """GEGLU projection [2*inner, C]: rows [value | gate] -> 16-row groups [16 value, 16 gate, ...]
    so value and gate of a channel land in the same lane / register of adjacent MFMA tiles."""
    inner = w.shape[0] // 2
    assert inner % 16 == 0
    wv, wg = w[:inner].reshape(inner // 16, 16, -1), w[inner:].reshape(inner // 16, 16, -1)
    wi = torch.stack([wv, wg], dim=1).reshape(2 * inner, -1)
    bv, bg = b[:inner].reshape(inner // 16, 16), b[inner:].reshape(inner // 16, 16)
    bi = torch.stack([bv, bg], dim=1).reshape(2 * inner)
    return wi, bi


# ----------------------------------------------------------------------------- launch tuning
# (tile, splits) per GEMM shape, measured on an MI355X by tools/autotune_gemm.py and committed as
# tuning/gfx950_gemm.json; shapes not in the table use the library's heuristic (tile 0, no split).
import json as _json
import os as _os

_TUNE_PATH = _os.environ.get("AF_TUNE_TABLE") or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuning", "gfx950_gemm.json")
_tune_table = None
def conv_halo_eligible(d) -> bool:
    """The scope of af_gemm tile 14, the halo-resident 3x3 kernel (include/adaface_hip.h); tools/autotune_gemm.py times it against the
    tap-by-tap tiles wherever this holds.  Three forms (csrc/af_gemm3.hip::conv3h_variant): 256 x 160 tiles on whole image rows (the U-Net's levels,
    with the K-concatenated shortcut since round 6), 256 x 128 tiles on whole rows, and 256 x 128 tiles on 16 x 16-pixel patches of images wider
    than 64 pixels (the VAE's 128 / 256 / 512 levels)."""
    up = 2 if d.upsample else 1
    tail = d.c3 > 0 or d.c4 > 0
    n160 = d.N % 160 == 0
    if not (d.taps == 9 and d.stride in (0, 1) and d.upsample in (0, 1) and not d.tap_shift and d.c1 % 64 == 0 and d.c2 % 64 == 0
            and (n160 or d.N % 128 == 0) and d.M % 256 == 0 and d.Ho == up * d.H and d.Wo == up * d.W and d.act != AF_ACT_GEGLU
            and d.out_mode == AF_OUT_NORMAL):
        return False
    if tail and not (n160 and d.c3 > 0 and d.c3 % 64 == 0 and d.c4 % 64 == 0 and not d.upsample):     # round 6: the K-concatenated 1x1 shortcut
        return False
    if d.Wo > 64:
        return not n160 and not tail and d.Wo % 16 == 0 and d.Ho % 16 == 0 and d.splits <= 1
    return d.Wo in (8, 16, 32, 64) and _halo_rows_ok(d.Ho, d.Wo)


def _halo_rows_ok(Ho: int, Wo: int) -> bool:
    """A tile of the halo-resident kernel is 256 output pixels: whole rows of one image, or -- at the 8 x 8 level -- whole images whose halo blocks fit its 400-pixel buffer."""
    rows = 256 // Wo
    if Ho % rows == 0:
        return True
    return rows % Ho == 0 and (rows // Ho) * (Ho + 2) * (Wo + 2) <= 400


_tune_recorder = None      # set by tools/autotune_gemm.py: callable(key, desc, device) -> (tile, splits)
_splitk_ws = {}
SPLITK_WS_BYTES = 192 << 20


def tune_table():
    global _tune_table
    if _tune_table is None:
        try:
            with open(_TUNE_PATH) as f:
                _tune_table = {k: tuple(v) for k, v in _json.load(f).items()}
        except FileNotFoundError:
            _tune_table = {}
    return _tune_table


_zero_pages = {}


def _zero_page(device) -> torch.Tensor:
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(64, dtype=F16, device=device)
        _zero_pages[device] = z
    return z


def _splitk_workspace(device) -> torch.Tensor:
    """fp32 partial slabs + (last AF_SPLITK_COUNTER_BYTES) the per-tile arrival counters of the in-kernel reduction, which must be
    zero between launches: zeroed here once, every launch restores them."""
    key = device if _ws_lane == 0 else (device, _ws_lane)
    ws = _splitk_ws.get(key)
    if ws is None:
        ws = torch.empty((SPLITK_WS_BYTES // 4,), dtype=torch.float32, device=device)
        ws[-(_lib.AF_SPLITK_COUNTER_BYTES // 4):].zero_()
        _splitk_ws[key] = ws
    return ws


_ws_lane = 0


def set_workspace_lane(lane: int) -> int:
    """Launches issued from now on use split-K workspace number ``lane`` (0 = the default one).  The slab workspace is the one piece of scratch every
    launch of a device shares; a caller that runs two independent passes CONCURRENTLY on two streams (tools/probes/r06w_two_streams.py) gives each stream
    its own lane while it issues that stream's launches.  Returns the previous lane."""
    global _ws_lane
    prev, _ws_lane = _ws_lane, int(lane)
    return prev


# split-K with at most this many slices is reduced inside the GEMM launch by the last-arriving workgroup of each tile (one launch
# instead of two: af_gemm_desc.splitk_fused).  MEASURED SLOWER than the chip-wide reduce pass on every shape of this path but two
# tiny ones (profiles/r02e_splitk_fused.txt: e.g. M2048 N1280 K1280 x2: 29.1 vs 24.5 us, conv 8x16x16 1280->1280 x4: 89.3 vs
# 78.3 us -- the one reducer workgroup per tile reads its slabs at a dependent-latency rate while the separate pass spreads them
# over all CUs and the launch boundary costs only ~1.5 us under graph replay), so the default is 0 = never; the path stays
# available (bit-identical results, tests/test_hip_kernels.py) for callers whose launch boundaries are expensive.
VAE_HALO_DEFAULT = _os.environ.get("AF_VAE_HALO_DEFAULT", "1") != "0"     # untabled VAE-kind 3x3 shapes go to the halo-resident kernel (see _launch_gemm)
SPLITK_FUSED_MAX = int(_os.environ.get("AF_SPLITK_FUSED_MAX", "0"))


# Round 5: the same in-kernel reduction chosen by SIZE for the register-staged tiles (1 / 2): when all slabs of a launch together are at most this
# many bytes (the training legs' batch-1 passes: M 64 ... 1024 rows, a few output tiles, 2 - 16 slices) the reduce pass is a ~5 us launch that moves
# a few hundred KB, and the last-arriving workgroup of a tile reads its slabs in about a microsecond.  0 = off.
SPLITK_FUSED_BYTES = int(_os.environ.get("AF_SPLITK_FUSED_BYTES", "0"))


# Round 5: tile 18 (fragments straight from global memory, no LDS / barrier / split-K) for the SMALL plain GEMMs the table or the heuristic would
# give to the register-staged 64 x 64 tile (tile 2, often with split-K): at most this many rows, K <= SMALL_GEMM_MAX_K.  0 rows = off.
SMALL_GEMM_MAX_M = int(_os.environ.get("AF_SMALL_GEMM_MAX_M", "0"))
SMALL_GEMM_MAX_K = int(_os.environ.get("AF_SMALL_GEMM_MAX_K", "3072"))


GN_FROM_PRODUCER = _os.environ.get("AF_GN_FROM_PRODUCER", "1") != "0"   # GroupNorm statistics from the producing GEMM's epilogue (0: always a statistics pass)


class GnPartials:
    """Partial GroupNorm statistics of a tensor, written by the GEMM launch that produced it (af_gemm_desc.gn_partials).  They describe the
    bytes that launch stored: `attach` records the tensor's storage address and autograd version, and `partials_of` hands them out only
    while both still match -- an in-place write between the producer and the GroupNorm (`y.add_(1)`, `copy_`) bumps the version, the
    statistics are dropped and the consumer runs its own statistics pass."""
    __slots__ = ("ws", "nblk", "cpg", "B", "hw", "C", "ptr", "version")

    def __init__(self, ws, nblk, cpg, B, hw, C):
        self.ws, self.nblk, self.cpg, self.B, self.hw, self.C = ws, nblk, cpg, B, hw, C
        self.ptr, self.version = 0, -1

    def attach(self, t: torch.Tensor) -> torch.Tensor:
        self.ptr, self.version = t.data_ptr(), t._version
        t._gn_partials = self
        return t


def partials_of(x: torch.Tensor) -> Optional["GnPartials"]:
    """The statistics x's producer left, or None when there are none or x has been written since (views share the version counter)."""
    gn = getattr(x, "_gn_partials", None)
    if gn is None or gn.ptr != x.data_ptr() or gn.version != x._version:
        return None
    return gn


# ---- split-K reduction left to the consuming GroupNorm (round 6; af_gemm_desc.defer_reduce / af_groupnorm_splitk) ---------------------------------
# Every ResBlock convolution of the 32 x 32 / 16 x 16 / 8 x 8 levels is a split-K launch, and what reads its output first is a GroupNorm(32): the
# chip-wide reduce pass (fp32 slabs -> fp16 tensor) followed by the GroupNorm's own read of that tensor are one launch and one round trip too many.
# A caller that KNOWS its output's next reader is `groupnorm` / `groupnorm_train` on the same tensor asks for `defer_gn=True`: the split launch leaves its
# slabs in the shared workspace, the returned tensor is NOT written yet, and one PendingReduce per device remembers what is owed.  `groupnorm(x)` on that
# tensor then runs af_groupnorm_splitk -- slabs -> (bias, row bias, residual) -> x stored AND normalised, one launch, x bit-identical to the reduce
# pass.  Safety net: any other wrapper that is handed the tensor (`_chk_f16` compares storage addresses: views included) and any further GEMM launch
# (it may reuse the slab workspace) first materialises it with the plain reduce pass (af_splitk_reduce).  A torch op reading the tensor directly
# would see unwritten memory: only call sites whose next launch is the GroupNorm may ask (ResBlock.hip / hip_train).
DEFER_GN_REDUCE = _os.environ.get("AF_DEFER_GN_REDUCE", "0") != "0"      # OFF by default: bit-identical and a TIE in the step (profiles/r06c_defer_gn_ab.txt)


class PendingReduce:
    __slots__ = ("slabs", "splits", "bias", "rowbias", "ld_rowbias", "rpb", "residual", "out", "ptr", "M", "N")


_pending = {}          # device -> PendingReduce (at most one: the slabs live in the per-device split-K workspace)
pending_stats = {"deferred": 0, "fused": 0, "flushed": 0}


def flush_pending(device=None):
    """Materialise the tensor(s) whose split-K reduce pass is still owed (af_splitk_reduce: the pass af_gemm would have run)."""
    for dev in ([device] if device is not None else list(_pending)):
        pr = _pending.pop(dev, None)
        if pr is not None:
            pending_stats["flushed"] += 1
            rc = _lib.lib().af_splitk_reduce(pr.slabs, pr.splits, _p(pr.bias), _p(pr.rowbias), pr.ld_rowbias, pr.rpb, _p(pr.residual), pr.ptr, pr.M, pr.N, _stream())
            _lib.check(rc, "af_splitk_reduce")


def _take_pending(x: torch.Tensor, B: int, hw: int, c: int) -> Optional["PendingReduce"]:
    """The reduce owed on x, if x is that tensor and has the shape the producer wrote; anything else pending on the device is materialised."""
    pr = _pending.get(x.device)
    if pr is None:
        return None
    if pr.ptr == x.data_ptr() and pr.M == B * hw and pr.N == c and x.is_contiguous() and pr.rpb == hw:
        del _pending[x.device]
        return pr
    flush_pending(x.device)
    return None


class WeightPrefetcher:
    """Weight prefetch on a SECOND stream inside a captured step (round 5 experiment; bench.py --prefetch-stream).

    Inside a denoise step every GEMM meets its weights cold in HBM (1.72 GB stream through once per U-Net pass; the GEMM family takes 7.0 ms
    on warm operands and 8.6 ms in the step, profiles/r04v_*).  The in-kernel prologue touch (af_common.h) only covers the first K stages.
    This object replays the step's own launch order: `record()` + one eager pass notes (weight pointer, bytes) of every weight-consuming
    launch; during the capture, in front of launch i the side stream waits for the main stream's position (so it runs beside launch i) and
    reads the weights of launch i + depth (af_prefetch_ex: one dword per 128-byte line, a capped grid), which brings them into the
    Infinity Cache (and 1/8 of them into the consumer XCD's L2).  The main stream never waits for the side stream except once at the end
    of the capture (`join`): the prefetch is a hint, results are bit-identical with and without it."""

    def __init__(self, depth: int = 2, max_workgroups: int = 64, min_bytes: int = 1 << 18):
        self.depth, self.max_wg, self.min_bytes = depth, max_workgroups, min_bytes
        self.plan, self.mode, self.i, self.side, self.issued = [], None, 0, None, 0

    def record(self):
        self.plan, self.mode = [], "record"
        return self

    def play(self, side_stream):
        self.mode, self.i, self.side, self.issued = "play", 0, side_stream, 0
        return self

    def stop(self):
        self.mode = None

    def note(self, ptr: int, nbytes: int):
        if self.mode == "record":
            self.plan.append((ptr, nbytes))
        elif self.mode == "play":
            i = self.i
            self.i = i + 1
            j = i + self.depth
            if i < len(self.plan) and self.plan[i][0] != ptr:
                raise RuntimeError("WeightPrefetcher: the captured pass does not follow the recorded launch order")
            if j < len(self.plan) and self.plan[j][1] >= self.min_bytes and torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event()
                ev.record()                                  # the main stream's position: everything before launch i
                self.side.wait_event(ev)
                _lib.check(_lib.lib().af_prefetch_ex(self.plan[j][0], self.plan[j][1], self.max_wg, self.side.cuda_stream), "af_prefetch_ex")
                self.issued += 1

    def join(self):
        """End of the capture: the forked stream must rejoin the capturing one."""
        if self.mode == "play" and self.side is not None and self.issued:
            torch.cuda.current_stream().wait_stream(self.side)
        self.mode = None


_weight_prefetcher: Optional[WeightPrefetcher] = None


def set_weight_prefetcher(p: Optional[WeightPrefetcher]):
    global _weight_prefetcher
    _weight_prefetcher = p


def _pf_note(*weights):
    if _weight_prefetcher is not None and _weight_prefetcher.mode is not None:
        for w in weights:
            _weight_prefetcher.note(w.data_ptr(), w.numel() * w.element_size())


def _launch_gemm(d: "GemmDesc", device, what: str, tile: int = 0, splits: int = 0, gn_cpg: int = 0, defer_gn=None):
    """Pick (tile, splits) -- explicit args > recorder (autotune) > table > heuristic -- and launch.  gn_cpg > 0: the caller's output feeds a
    GroupNorm with groups of gn_cpg channels; when the chosen launch can (af_gemm_gn_stats_ok) it also writes the partial statistics and a
    GnPartials is returned (None otherwise).  defer_gn = (out tensor, bias, rowbias, residual) from a caller whose output's NEXT reader is that
    GroupNorm: a split launch then leaves its reduce pass to it (PendingReduce above)."""
    if _pending:
        flush_pending(device)                       # the slab workspace is about to be reused / the owed tensor may be an operand
    if tile == 0 and splits == 0:
        key = f"{d.taps},{d.M},{d.N},{d.K},{d.act},{d.out_mode},{d.stride},{d.upsample}"
        if d.ln_colsum:
            key += ",ln"
        if _tune_recorder is not None:
            tile, splits = _tune_recorder(key, d, device)
        else:
            tile, splits = tune_table().get(key, (0, 1))
            if d.ln_colsum and tile == 0:
                tile, splits = tune_table().get(key[:-3], (0, 1))
            if VAE_HALO_DEFAULT and tile == 0 and d.taps == 9 and d.N % 160 != 0 and d.M >= 16384 and conv_halo_eligible(d):
                # a 3x3 convolution of the VAE's kind (128-multiples of channels over many pixels) at a batch size the table has not met: the halo-resident
                # kernel's 256 x 128 forms win every such shape the table holds by 20 - 40 % (profiles/r06t_try14.log)
                tile, splits = 14, 1
    if (SMALL_GEMM_MAX_M and tile in (0, 2) and d.taps == 1 and d.M <= SMALL_GEMM_MAX_M and d.K <= SMALL_GEMM_MAX_K and not d.a2 and not d.ln_colsum
            and d.act != AF_ACT_GEGLU and d.out_mode != AF_OUT_SPLIT_T and d.K % 8 == 0 and _tune_recorder is None
            and (tile == 2 or (d.M * d.N <= (1 << 21)))):
        tile, splits = 18, 1                                  # the library falls back to tile 2 if a stride / alignment rule is not met
    if d.ln_colsum:
        # the folded LayerNorm lives in the whole-line kernel only (tiles 7 .. 13, 16, 17), unsplit
        splits = 1
        if tile < 7 or tile in (14, 15) or tile > 17 or (tile in (9, 10) and d.act != AF_ACT_GEGLU) or (tile in (16, 17) and d.act == AF_ACT_GEGLU):
            if d.act == AF_ACT_GEGLU:
                tile = 7 if d.N % 256 == 0 else 8
            else:
                tile = 7 if (d.N % 320 == 0 and d.M >= 8192) else 8
    d.tile = tile
    d.zeros = _zero_page(device).data_ptr()
    f32 = d.out_mode == AF_OUT_F32
    d.splits = max(1, splits)
    if d.splits > 1:
        if d.act == AF_ACT_GEGLU or d.out_mode == AF_OUT_SPLIT_T:
            d.splits = 1
        else:
            ws = _splitk_workspace(device)
            d.splits = max(1, min(d.splits, (ws.numel() * 4 - _lib.AF_SPLITK_COUNTER_BYTES) // (d.M * d.N * 4)))
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
            small = d.tile in (0, 1, 2) and d.splits <= 16 and d.splits * d.M * d.N * 4 <= SPLITK_FUSED_BYTES
            d.splitk_fused = int((d.splits <= SPLITK_FUSED_MAX or small) and not f32)
    gn = None
    if gn_cpg and GN_FROM_PRODUCER and _tune_recorder is None:
        rpb = d.rows_per_batch if d.rows_per_batch > 0 else d.M
        # the table's key carries no image geometry: tile 14 (halo-resident 3x3) on a latent outside its scope (W not 16 / 32 / 64, ragged rows)
        # falls back to a tap-by-tap tile INSIDE the library, which leaves no statistics -- ask for them only where tile 14 will really run
        if (d.tile not in (14, 19) or conv_halo_eligible(d)) and d.M % rpb == 0 and d.ld_out in (0, d.N) \
                and _lib.lib().af_gemm_gn_stats_ok(d.tile, d.splits, d.taps, d.act, d.out_mode, d.N, gn_cpg, rpb) == 1:
            nb = d.M // rpb
            ws = torch.empty((nb, 128, 32, 2), dtype=torch.float32, device=device)
            d.gn_partials, d.gn_cpg = ws.data_ptr(), gn_cpg
            gn = GnPartials(ws, rpb // 128, gn_cpg, nb, rpb, d.N)
    if _weight_prefetcher is not None and _weight_prefetcher.mode is not None:
        _weight_prefetcher.note(int(d.wt), int(d.kpad) * round_up(int(d.N), 128) * 2)
    left = None
    if (defer_gn is not None and DEFER_GN_REDUCE and d.splits > 1 and not d.splitk_fused and gn_cpg and _tune_recorder is None and d.act == AF_ACT_NONE
            and d.out_mode == AF_OUT_NORMAL and d.ld_out in (0, d.N) and d.N % gn_cpg == 0):
        rpb = d.rows_per_batch if d.rows_per_batch > 0 else d.M
        if d.M % rpb == 0 and _lib.lib().af_groupnorm_splitk_ok(d.M // rpb, rpb, d.N, d.N // gn_cpg) == 1:
            left = C.c_int32(0)
            d.defer_reduce = C.pointer(left)
    _lib.check(_lib.lib().af_gemm(C.byref(d), _stream()), what)
    if left is not None and left.value > 1:
        pr = PendingReduce()
        out, bias, rowbias, residual = defer_gn
        pr.slabs, pr.splits, pr.bias, pr.rowbias, pr.residual, pr.out = int(d.workspace), left.value, bias, rowbias, residual, out
        pr.ld_rowbias, pr.rpb, pr.ptr, pr.M, pr.N = int(d.ld_rowbias), (d.rows_per_batch if d.rows_per_batch > 0 else d.M), out.data_ptr(), int(d.M), int(d.N)
        _pending[device] = pr
        pending_stats["deferred"] += 1
    return gn


# ----------------------------------------------------------------------------- gemm / conv
def gemm(a1: torch.Tensor, pw: PackedWeight, *, a2: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
         rows_per_batch: int = 0, residual: Optional[torch.Tensor] = None, act: int = AF_ACT_NONE,
         split_col: int = 0, ld_out2: int = 0, tile: int = 0, splits: int = 0, out_f32: bool = False, gn_cpg: int = 0):
    """Plain-rows GEMM: a1 [M, K1] (+ a2 [M, K2], concatenated along K) x pw.  Returns out
    ([M, N], or [M, N/2] for GEGLU), or (out [M, split_col], out2 [B, N-split_col, ld_out2]).  out_f32: the fp32 accumulator is
    returned (AF_OUT_F32; weight gradients)."""
    _chk_f16(a1, "gemm.a1")
    M, k1 = a1.shape
    k2 = 0
    if a2 is not None:
        _chk_f16(a2, "gemm.a2")
        k2 = a2.shape[1]
        assert a2.shape[0] == M
    assert k1 + k2 == pw.K, f"gemm: K mismatch {k1}+{k2} vs {pw.K}"
    d = GemmDesc()
    d.a1, d.a2, d.wt, d.bias = _p(a1), _p(a2), _p(pw.wt), _p(pw.bias)
    d.rowbias, d.residual = _p(rowbias), _p(residual)
    d.M, d.N, d.K, d.kpad, d.taps = M, pw.N, pw.K, pw.kpad, 1
    d.c1, d.c2, d.lda1, d.lda2 = k1, k2, k1, k2
    d.rows_per_batch = rows_per_batch
    d.ld_rowbias = 0 if rowbias is None else rowbias.stride(0)
    d.act = act
    if pw.ln_cs is not None:
        assert a2 is None and not out_f32, "gemm: a folded LayerNorm takes one source and the fp16 epilogues"
        d.ln_colsum, d.ln_eps = _p(pw.ln_cs), pw.ln_eps
    out2 = None
    if split_col:
        assert rows_per_batch > 0 and M % rows_per_batch == 0
        nb = M // rows_per_batch
        ld_out2 = ld_out2 or round_up(rows_per_batch, 8)
        out = torch.empty((M, split_col), dtype=F16, device=a1.device)
        out2 = torch.empty((nb, pw.N - split_col, ld_out2), dtype=F16, device=a1.device)
        d.out_mode, d.split_col, d.ld_out2, d.out2 = AF_OUT_SPLIT_T, split_col, ld_out2, _p(out2)
    elif out_f32:
        assert act == AF_ACT_NONE and pw.N % 4 == 0
        out = torch.empty((M, pw.N), dtype=torch.float32, device=a1.device)
        d.out_mode = AF_OUT_F32
    else:
        n_out = pw.N // 2 if act == AF_ACT_GEGLU else pw.N
        out = torch.empty((M, n_out), dtype=F16, device=a1.device)
    if residual is not None:
        _chk_f16(residual, "gemm.residual")
        assert residual.shape == out.shape
    d.out = _p(out)
    gn = _launch_gemm(d, a1.device, "af_gemm", tile, splits, gn_cpg=0 if (split_col or out_f32) else gn_cpg)
    if gn is not None:
        gn.attach(out)                   # rides on the tensor object: GroupNorm32.hip picks it up (ops.groupnorm)
    return (out, out2) if split_col else out


def conv3x3_skip_tile(M: int, N: int, cin: int, ktail: int):
    """(tile, splits) for a 3x3 convolution with a K-concatenated 1x1 tail: its own tuned entry when the table has one on a whole-line tile,
    else what the table says for the plain convolution of the same shape (the tail only lengthens K), else a whole-line default."""
    for K in (9 * cin + ktail, 9 * cin):
        t = tune_table().get(f"9,{M},{N},{K},0,0,1,0")
        if t is not None and 7 <= t[0] <= 14:          # (14: the halo-resident kernel takes the tail since round 6; conv3x3 falls back inside the library outside its scope)
            return t
    return (7 if (N % 320 == 0 and M >= 8192) else (11 if N % 160 == 0 else 8)), 1


def conv3x3(x: torch.Tensor, pw: PackedWeight, *, x2: Optional[torch.Tensor] = None, stride: int = 1, upsample: bool = False,
            rowbias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, tile: int = 0,
            splits: int = 0, out_hw=None, tap_shift: int = 0, skip=None, gn_cpg: int = 0, defer_gn: bool = False) -> torch.Tensor:
    """3x3 / pad 1 convolution as implicit GEMM (tap_shift=1: padding (0, 1, 0, 1) instead, the VAE encoder's Downsample).  x [B,H,W,C1] (+ x2 [B,H,W,C2] channel-concat)
    -> [B,Ho,Wo,Cout].  rowbias [B, >=Cout] is added per batch item (time-embedding), residual
    [B,Ho,Wo,Cout] after it."""
    _chk_f16(x, "conv3x3.x")
    B, H, W, c1 = x.shape
    c2 = 0
    if x2 is not None:
        _chk_f16(x2, "conv3x3.x2")
        assert x2.shape[:3] == x.shape[:3]
        c2 = x2.shape[3]
    assert pw.taps == 9 and pw.cin == c1 + c2, f"conv3x3: channel mismatch {c1}+{c2} vs {pw.cin}"
    assert (pw.k_tail > 0) == (skip is not None), "conv3x3: a weight packed with a 1x1 tail (pack_conv3x3_skip) needs skip=(s1, s2 | None), and vice versa"
    he, we = (2 * H, 2 * W) if upsample else (H, W)
    ho, wo = (he + 2 - 3) // stride + 1, (we + 2 - 3) // stride + 1
    if int(upsample) == 2:      # zero-inserted image (stride-2 dgrad): output size = forward input size
        ho, wo = out_hw if out_hw is not None else (2 * H, 2 * W)
        assert ho in (2 * H, 2 * H - 1) and wo in (2 * W, 2 * W - 1)
    out = torch.empty((B, ho, wo, pw.N), dtype=F16, device=x.device)
    d = GemmDesc()
    d.a1, d.a2, d.wt, d.bias = _p(x), _p(x2), _p(pw.wt), _p(pw.bias)
    d.rowbias, d.residual, d.out = _p(rowbias), _p(residual), _p(out)
    d.M, d.N, d.K, d.kpad, d.taps = B * ho * wo, pw.N, pw.K, pw.kpad, 9
    d.c1, d.c2 = c1, c2
    d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, ho, wo
    d.stride, d.upsample, d.tap_shift = stride, int(upsample), int(tap_shift)
    d.rows_per_batch = ho * wo
    d.ld_rowbias = 0 if rowbias is None else rowbias.stride(0)
    if residual is not None:
        _chk_f16(residual, "conv3x3.residual")
        assert residual.shape == out.shape
    if skip is not None:
        # the 1x1 tail reads s1 (| s2) at the output pixel: same grid as the output, stride 1
        s1, s2 = skip
        assert stride == 1 and not upsample and not tap_shift and (ho, wo) == (H, W)
        _chk_f16(s1, "conv3x3.skip")
        assert s1.shape[:3] == out.shape[:3]
        c3, c4 = s1.shape[3], 0
        if s2 is not None:
            _chk_f16(s2, "conv3x3.skip2")
            assert s2.shape[:3] == out.shape[:3]
            c4 = s2.shape[3]
        assert c3 + c4 == pw.k_tail, f"conv3x3: skip channels {c3}+{c4} vs the packed tail {pw.k_tail}"
        d.a3, d.a4, d.c3, d.c4, d.lda3, d.lda4 = _p(s1), _p(s2), c3, c4, c3, c4
        if tile == 0 and splits == 0 and _tune_recorder is None:
            tile, splits = conv3x3_skip_tile(d.M, d.N, c1 + c2, pw.k_tail)
            if tile == 14 and not conv_halo_eligible(d):
                # the table's key has no image geometry: tile 14 on a latent outside its scope (W not 8 / 16 / 32 / 64, ragged rows) would fall back to
                # the register-staged kernel inside the library, which has no K tail -- take a whole-line tap-by-tap tile instead
                tile, splits = (7 if (d.N % 320 == 0 and d.M >= 8192) else (11 if d.N % 160 == 0 else 8)), 1
    gn = _launch_gemm(d, x.device, "af_gemm(conv3x3)", tile, splits, gn_cpg=gn_cpg,
                      defer_gn=(out, pw.bias, rowbias, residual) if defer_gn else None)
    if gn is not None:
        gn.attach(out)
    return out


def ff_fused(x2d: torch.Tensor, pw1: PackedWeight, pw2: PackedWeight, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = residual + b2 + W2 (v * gelu(g)), [v | g] = W1 LN(x) + b1 in one launch (af_ff_fused; C = 320 only).  pw1: the GEGLU-
    interleaved first projection, normally with the LayerNorm folded in (GEGLU.packed_ln); pw2: the second projection."""
    _chk_f16(x2d, "ff_fused.x")
    M, Cn = x2d.shape
    assert pw1.K == Cn and pw2.N == Cn and pw2.K * 2 == pw1.N and pw1.bias is not None
    out = torch.empty((M, Cn), dtype=F16, device=x2d.device)
    if residual is not None:
        _chk_f16(residual, "ff_fused.residual")
        assert residual.shape == out.shape
    _pf_note(pw1.wt, pw2.wt)
    rc = _lib.lib().af_ff_fused(_p(x2d), _p(pw1.wt), _p(pw1.bias), _p(pw1.ln_cs), float(pw1.ln_eps), pw1.kpad, _p(pw2.wt), _p(pw2.bias), pw2.kpad,
                                _p(residual), _p(out), M, Cn, pw2.K, _p(_zero_page(x2d.device)), _stream())
    _lib.check(rc, "af_ff_fused")
    return out


def ff_chain(x2d: torch.Tensor, pw1: PackedWeight, pw2: PackedWeight, residual: torch.Tensor, pw_p: PackedWeight, x_in: torch.Tensor, *, rows_per_batch: int,
             gn_cpg: int = 0) -> torch.Tensor:
    """af_ff_chain: `ff_fused` with the SpatialTransformer's proj_out + residual behind it, one launch -- out = x_in + b_p + W_p (residual + b2 + W2 (v * gelu(g))).
    gn_cpg > 0: `out` feeds a GroupNorm with groups of gn_cpg channels; where the shape allows, the launch leaves the partial statistics with it (GnPartials)."""
    _chk_f16(x2d, "ff_chain.x")
    _chk_f16(residual, "ff_chain.residual")
    _chk_f16(x_in, "ff_chain.x_in")
    M, Cn = x2d.shape
    assert pw1.K == Cn and pw2.N == Cn and pw2.K * 2 == pw1.N and pw1.bias is not None and pw_p.N == Cn and pw_p.K == Cn and pw_p.ln_cs is None
    assert residual.shape == x2d.shape and x_in.shape == x2d.shape and M % rows_per_batch == 0
    out = torch.empty((M, Cn), dtype=F16, device=x2d.device)
    gn, gnp = None, None
    if gn_cpg and GN_FROM_PRODUCER and gn_cpg % 2 == 0 and Cn % gn_cpg == 0 and Cn // gn_cpg <= 32 and rows_per_batch % 128 == 0 and rows_per_batch // 128 <= 128:
        nb = M // rows_per_batch
        ws = torch.empty((nb, 128, 32, 2), dtype=torch.float32, device=x2d.device)
        gn, gnp = GnPartials(ws, rows_per_batch // 128, gn_cpg, nb, rows_per_batch, Cn), ws.data_ptr()
    _pf_note(pw1.wt, pw2.wt, pw_p.wt)
    rc = _lib.lib().af_ff_chain(_p(x2d), _p(pw1.wt), _p(pw1.bias), _p(pw1.ln_cs), float(pw1.ln_eps), pw1.kpad, _p(pw2.wt), _p(pw2.bias), pw2.kpad, _p(residual),
                                _p(pw_p.wt), _p(pw_p.bias), pw_p.kpad, _p(x_in), _p(out), gnp, int(gn_cpg if gn is not None else 0), int(rows_per_batch), M, Cn, pw2.K,
                                _p(_zero_page(x2d.device)), _stream())
    _lib.check(rc, "af_ff_chain")
    if gn is not None:
        gn.attach(out)
    return out


# ----------------------------------------------------------------------------- norms
_gn_ws = {}


def xattn_fused(x2d: torch.Tensor, pw_q: PackedWeight, k: torch.Tensor, vt: torch.Tensor, pw_o: PackedWeight, *, B: int, N: int, L: int, heads: int,
                scale: float, ldk: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The whole cross-attention block of a C = 320 transformer layer in one launch (af_xattn_fused): q projection (pw_q: normally with
    the LayerNorm folded in, ops.pack_matrix_ln), 77-key attention over k [B*L, >= C] (row stride ldk) / vt [B, C, ld], output projection
    pw_o with its bias, residual."""
    _chk_f16(x2d, "xattn_fused.x")
    M, Cn = x2d.shape
    assert M == B * N and pw_q.N == Cn and pw_o.N == Cn and pw_o.K == Cn and vt.stride(2) == 1 and k.stride(1) == 1
    out = torch.empty_like(x2d)
    if residual is not None:
        _chk_f16(residual, "xattn_fused.residual")
        assert residual.shape == x2d.shape
    _pf_note(pw_q.wt, pw_o.wt)
    rc = _lib.lib().af_xattn_fused(_p(x2d), _p(pw_q.wt), _p(pw_q.bias), _p(pw_q.ln_cs), float(pw_q.ln_eps), pw_q.kpad, _p(k), int(ldk), _p(vt), int(vt.stride(0)),
                                   int(vt.stride(1)), _p(pw_o.wt), _p(pw_o.bias), pw_o.kpad, _p(residual), _p(out), B, N, L, Cn, heads, float(scale),
                                   _zero_page(x2d.device).data_ptr(), _stream())
    _lib.check(rc, "af_xattn_fused")
    return out


def xattn_chain(ao: torch.Tensor, pw_o1: PackedWeight, x0: torch.Tensor, pw_q: PackedWeight, k: torch.Tensor, vt: torch.Tensor, pw_o: PackedWeight, *,
                B: int, N: int, L: int, heads: int, scale: float, ldk: int) -> torch.Tensor:
    """af_xattn_chain: the self-attention's output projection + residual (ao, pw_o1, x0) in front of the C = 320 cross-attention block, one launch.
    Returns the block's output; the intermediate x1 = ao W1^T + b1 + x0 lives in a scratch tensor of this call."""
    _chk_f16(ao, "xattn_chain.ao")
    _chk_f16(x0, "xattn_chain.x0")
    M, Cn = ao.shape
    assert M == B * N and x0.shape == ao.shape and pw_o1.N == Cn and pw_o1.K == Cn and pw_q.N == Cn and pw_o.N == Cn and pw_o.K == Cn
    assert vt.stride(2) == 1 and k.stride(1) == 1 and pw_o1.ln_cs is None
    x1 = torch.empty_like(ao)
    out = torch.empty_like(ao)
    _pf_note(pw_o1.wt, pw_q.wt, pw_o.wt)
    rc = _lib.lib().af_xattn_chain(_p(ao), _p(pw_o1.wt), _p(pw_o1.bias), pw_o1.kpad, _p(x0), _p(x1), _p(pw_q.wt), _p(pw_q.bias), _p(pw_q.ln_cs), float(pw_q.ln_eps),
                                   pw_q.kpad, _p(k), int(ldk), _p(vt), int(vt.stride(0)), int(vt.stride(1)), _p(pw_o.wt), _p(pw_o.bias), pw_o.kpad, _p(out), B, N, L, Cn,
                                   heads, float(scale), _zero_page(ao.device).data_ptr(), _stream())
    _lib.check(rc, "af_xattn_chain")
    return out


def _grow_scratch(cache: dict, device, need: int, dtype, floor: int = 0) -> torch.Tensor:
    """Per-device scratch that grows on demand.  A hipGraph bakes in the address of whatever buffer a captured launch was given, and
    growing the cached buffer frees the old one -- so while the current stream is CAPTURING the scratch comes from the capturing graph's
    own pool instead (it lives exactly as long as that graph), and the growable buffer only ever serves eager launches."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((need,), dtype=dtype, device=device)
    sc = cache.get(device)
    if sc is None or sc.numel() < need:
        sc = torch.empty((max(need, floor),), dtype=dtype, device=device)
        cache[device] = sc
    return sc


def _gn_workspace(device, B: int) -> torch.Tensor:
    return _grow_scratch(_gn_ws, device, _lib.lib().af_groupnorm_ws_floats(max(B, 16)), torch.float32)


def _groupnorm_splitk(pr, x, gamma, beta, y, stats, B, hw, c, groups, eps, silu) -> bool:
    """x's producer left its split-K slabs (PendingReduce): finish x AND normalise it in one launch.  False (after materialising x the plain way)
    when the GroupNorm is outside the one-launch forms' scope."""
    if _lib.lib().af_groupnorm_splitk_ok(B, hw, c, groups) != 1 or gamma.data_ptr() % 16 or beta.data_ptr() % 16:
        _pending[x.device] = pr
        flush_pending(x.device)
        return False
    rc = _lib.lib().af_groupnorm_splitk(pr.slabs, pr.splits, _p(pr.bias), _p(pr.rowbias), pr.ld_rowbias, _p(pr.residual), _p(x), c, _p(gamma), _p(beta), _p(y),
                                        _p(stats), B, hw, groups, float(eps), int(silu), _stream())
    _lib.check(rc, "af_groupnorm_splitk")
    pending_stats["fused"] += 1
    return True


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, silu: bool, *,
              x2: Optional[torch.Tensor] = None, groups: int = 32) -> torch.Tensor:
    """x [B, ..., C1] (+ x2 [B, ..., C2]) -> [B, ..., C1+C2] fp16; gamma/beta fp32 [C1+C2]."""
    _chk_f16_raw(x, "groupnorm.x")               # (not _chk_f16: x may be the tensor whose reduce pass this GroupNorm is about to absorb)
    B, c1 = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c1)
    c2 = 0
    if x2 is not None:
        _chk_f16(x2, "groupnorm.x2")
        c2 = x2.shape[-1]
    y = torch.empty(tuple(x.shape[:-1]) + (c1 + c2,), dtype=F16, device=x.device)
    if _pending:
        pr = _take_pending(x, B, hw, c1) if x2 is None else flush_pending(x.device)
        if pr is not None and _groupnorm_splitk(pr, x, gamma, beta, y, None, B, hw, c1, groups, eps, silu):
            return y
    gn = partials_of(x) if x2 is None else None
    if gn is not None and gn.B == B and gn.hw == hw and gn.C == c1 and gn.cpg * groups == c1 and x.is_contiguous():
        # the launch that produced x left its partial statistics: normalise in one pass, no statistics pass (af_groupnorm_apply)
        rc = _lib.lib().af_groupnorm_apply(_p(x), c1, _p(gamma), _p(beta), _p(y), None, B, hw, groups, float(eps), int(silu), _p(gn.ws), gn.nblk, _stream())
        _lib.check(rc, "af_groupnorm_apply")
        return y
    ws = _gn_workspace(x.device, B)
    rc = _lib.lib().af_groupnorm(_p(x), _p(x2), c1, c2, _p(gamma), _p(beta), _p(y), B, hw, groups, float(eps), int(silu),
                                 _p(ws), _stream())
    _lib.check(rc, "af_groupnorm")
    return y


def gn_proj_fused(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, pw: PackedWeight, groups: int = 32) -> Optional[torch.Tensor]:
    """GroupNorm(x) followed by a 1x1 convolution / Linear at C = 320 in ONE launch (af_gn_proj_fused): x [B, ..., 320] carrying the partial
    statistics its producer left (x._gn_partials) -> [B * HW, N = 320] token-major.  Returns None when x does not qualify (no partials, another
    width, ragged image size): the caller then runs the two launches."""
    gn = partials_of(x)
    B, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c)
    if (gn is None or c != 320 or groups != 32 or pw.N != 320 or pw.K != 320 or pw.ln_cs is not None or hw % 128 != 0 or gn.B != B or gn.hw != hw or gn.C != c
            or gn.cpg * groups != c or not x.is_contiguous()):
        return None
    _chk_f16(x, "gn_proj_fused.x")
    out = torch.empty((B * hw, pw.N), dtype=F16, device=x.device)
    _pf_note(pw.wt)
    rc = _lib.lib().af_gn_proj_fused(_p(x), _p(gn.ws), gn.nblk, _p(gamma), _p(beta), float(eps), _p(pw.wt), _p(pw.bias), pw.kpad, _p(out), B, hw, c, groups,
                                     _zero_page(x.device).data_ptr(), _stream())
    _lib.check(rc, "af_gn_proj_fused")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _chk_f16(x, "layernorm.x")
    Cn = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.lib().af_layernorm(_p(x), _p(gamma), _p(beta), _p(y), x.numel() // Cn, Cn, float(eps), _stream()),
               "af_layernorm")
    return y


# ----------------------------------------------------------------------------- attention
def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, *, B: int, Nq: int, L: int, heads: int, d: int,
              ldq: int, ldk: int, keybias: Optional[torch.Tensor] = None, scale: Optional[float] = None,
              want_lse: bool = False, causal_m: int = 0):
    """q [B*Nq, ldq-wide rows], k [B*L, ldk-wide rows], vt [B, heads*d, ldv] -> o [B*Nq, heads*d]
    (and, with want_lse, the base-2 log-sum-exp fp32 [B, heads, roundup(Nq,32)] for the backward)."""
    Cn = heads * d
    o = torch.empty((B * Nq, Cn), dtype=F16, device=q.device)
    scale = d ** -0.5 if scale is None else scale
    ldb = 0 if keybias is None else keybias.stride(0)
    lse = torch.empty((B, heads, round_up(Nq, 32)), dtype=torch.float32, device=q.device) if want_lse else None
    rc = _lib.lib().af_attention_strided(_p(q), _p(k), _p(vt), _p(o), _p(lse), 0 if lse is None else lse.stride(1), _p(keybias),
                                         int(causal_m), B, Nq, L, heads, d, ldq, ldk, Cn, vt.stride(1), ldb, vt.stride(0), float(scale),
                                         _stream())
    _lib.check(rc, "af_attention")
    return (o, lse) if want_lse else o


_attn_scratch = {}


def attention_bwd(q, k, v, o, dout, lse, *, B: int, Nq: int, L: int, heads: int, d: int, ldq: int, ldk: int, ldv: int,
                  dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, lddq: int, lddk: int, lddv: int,
                  keybias: Optional[torch.Tensor] = None, scale: Optional[float] = None, causal_m: int = 0):
    """Input gradients of `attention`.  q/k/v: row-major token tensors (views allowed: row strides ldq/ldk/ldv);
    o, dout [B*Nq, heads*d] contiguous; dq/dk/dv are written in place with row strides lddq/lddk/lddv."""
    Cn = heads * d
    scale = d ** -0.5 if scale is None else scale
    need = _lib.lib().af_attention_bwd_scratch_bytes(B, Nq, L, heads, d)
    sc = _grow_scratch(_attn_scratch, q.device, need, torch.uint8, floor=64 << 20)
    ldb = 0 if keybias is None else keybias.stride(0)
    rc = _lib.lib().af_attention_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), lse.stride(1), _p(keybias), int(causal_m), _p(dq), _p(dk),
                                     _p(dv), _p(sc), sc.numel(), B, Nq, L, heads, d, ldq, ldk, ldv, Cn, Cn, lddq, lddk, lddv,
                                     ldb, float(scale), _stream())
    _lib.check(rc, "af_attention_bwd")


def attention_scores(q: torch.Tensor, k: torch.Tensor, *, B: int, Nq: int, L: int, heads: int, d: int,
                     scale: Optional[float] = None):
    """Explicit (score, prob) fp32 [B, heads, Nq, L] for the capture path; q [B*Nq, C], k [B*L, C] contiguous."""
    _chk_f16(q, "attention_scores.q")
    _chk_f16(k, "attention_scores.k")
    score = torch.empty((B, heads, Nq, L), dtype=torch.float32, device=q.device)
    prob = torch.empty_like(score)
    scale = d ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().af_attention_scores(_p(q), _p(k), _p(score), _p(prob), B, Nq, L, heads, d, float(scale), _stream()),
               "af_attention_scores")
    return score, prob


# ---- explicit cross-attention (capture / score-rewrite path; csrc/af_xattn_explicit.hip) -- q [B*Nq, >=C], k / v [B*L, >=C] fp16 rows
def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D operand expected"
    return t.stride(0)


def _chk_f16_rows(t: torch.Tensor, name: str):
    """fp16 device matrix whose rows are contiguous (a column slice of a wider buffer is fine: the kernels take leading dimensions)."""
    if t.dtype != F16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 != 0 or t.data_ptr() % 16 != 0:
        raise RuntimeError(f"{name}: expected an fp16 device matrix with contiguous, 16-byte aligned rows, got {t.dtype} {t.device} strides {t.stride()}")


def xattn_scores(q, k, *, B, Nq, L, heads, d, scale):
    """score fp32 [B, heads, Nq, L] = scale * q k^T."""
    _chk_f16_rows(q, "xattn_scores.q")
    _chk_f16_rows(k, "xattn_scores.k")
    score = torch.empty((B, heads, Nq, L), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().af_xattn_scores(_p(q), _ld(q), _p(k), _ld(k), _p(score), B, Nq, L, heads, d, float(scale), _stream()), "af_xattn_scores")
    return score


def xattn_softmax_pv(score, v, *, B, Nq, L, heads, d):
    """(prob fp32 [B, heads, Nq, L], o fp16 [B*Nq, heads*d]) from (rewritten) scores."""
    _chk_f16_rows(v, "xattn_softmax_pv.v")
    assert score.dtype == torch.float32 and score.is_contiguous() and tuple(score.shape) == (B, heads, Nq, L)
    prob = torch.empty_like(score)
    o = torch.empty((B * Nq, heads * d), dtype=F16, device=v.device)
    _lib.check(_lib.lib().af_xattn_softmax_pv(_p(score), _p(v), _ld(v), _p(prob), _p(o), heads * d, B, Nq, L, heads, d, _stream()), "af_xattn_softmax_pv")
    return prob, o


def xattn_softmax_pv_bwd(prob, v, dout, dprob_ext, *, B, Nq, L, heads, d):
    """dscore fp32 [B, heads, Nq, L]; dprob_ext: gradient arriving on the captured probabilities (fp32, same shape) or None."""
    _chk_f16_rows(dout, "xattn_softmax_pv_bwd.dout")
    dscore = torch.empty_like(prob)
    if dprob_ext is not None:
        dprob_ext = dprob_ext.to(torch.float32).contiguous()
    _lib.check(_lib.lib().af_xattn_softmax_pv_bwd(_p(prob), _p(v), _ld(v), _p(dout), _ld(dout), _p(dprob_ext), _p(dscore), B, Nq, L, heads, d,
                                                  _stream()), "af_xattn_softmax_pv_bwd")
    return dscore


def xattn_rowmix(w, x, alpha, *, B, Nq, L, heads, d):
    """out fp16 [B*Nq, heads*d] = alpha * w x  (w fp32 [B, heads, Nq, L], x fp16 [B*L, >=C])."""
    out = torch.empty((B * Nq, heads * d), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_xattn_rowmix(_p(w), _p(x), _ld(x), _p(out), heads * d, float(alpha), B, Nq, L, heads, d, _stream()), "af_xattn_rowmix")
    return out


def xattn_colmix(w, x, alpha, *, B, Nq, L, heads, d):
    """out fp16 [B*L, heads*d] = alpha * w^T x  (x fp16 [B*Nq, >=C]): the reductions over the queries (dk, dv)."""
    out = torch.empty((B * L, heads * d), dtype=F16, device=x.device)
    nb = int(_lib.lib().af_xattn_colmix_ws_bytes(B, L, heads, d))
    ws = torch.empty((nb // 4,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().af_xattn_colmix(_p(w), _p(x), _ld(x), _p(out), heads * d, float(alpha), _p(ws), nb, B, Nq, L, heads, d, _stream()),
               "af_xattn_colmix")
    return out


def make_keybias(mask: torch.Tensor, L: int) -> torch.Tensor:
    """mask [B, L] (nonzero = keep) -> fp32 [B, roundup(L, 64)] additive bias: 0 / -FLT_MAX
    (== masked_fill_(~mask, -finfo.max), attention.py:188-194)."""
    B = mask.shape[0]
    kb = torch.zeros((B, round_up(L, 64)), dtype=torch.float32, device=mask.device)
    kb[:, :L] = torch.where(mask.bool(), 0.0, NEG_MAX)
    return kb


# ----------------------------------------------------------------------------- element-wise
def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    t = timesteps.to(dtype=torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=F16, device=t.device)
    _lib.check(_lib.lib().af_timestep_embedding(_p(t), _p(out), t.shape[0], dim, float(max_period), _stream()),
               "af_timestep_embedding")
    return out


def nchw_f32_to_nhwc_f16(x: torch.Tensor, cpad: int = 0) -> torch.Tensor:
    x = x.to(torch.float32).contiguous()
    B, Cn, H, W = x.shape
    cpad = max(cpad, Cn)
    y = torch.empty((B, H, W, cpad), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_nchw_f32_to_nhwc_f16(_p(x), _p(y), B, Cn, H * W, cpad, _stream()), "af_nchw_f32_to_nhwc_f16")
    return y


def nhwc_f16_to_nchw_f32(x: torch.Tensor, Cn: Optional[int] = None) -> torch.Tensor:
    _chk_f16(x, "nhwc_f16_to_nchw_f32.x")
    B, H, W, cs = x.shape
    Cn = Cn or cs
    y = torch.empty((B, Cn, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().af_nhwc_f16_to_nchw_f32(_p(x), _p(y), B, Cn, H * W, cs, _stream()), "af_nhwc_f16_to_nchw_f32")
    return y


def silu(x: torch.Tensor) -> torch.Tensor:
    _chk_f16(x, "silu.x")
    y = torch.empty_like(x)
    _lib.check(_lib.lib().af_silu_f16(_p(x), _p(y), x.numel(), _stream()), "af_silu_f16")
    return y


def cfg_ddim_step(eps2: torch.Tensor, x: torch.Tensor, guidance: float, a_t: float, a_prev: float, has_uncond: bool = True):
    """eps2 fp32 [2n or n] = [e_cond ; e_uncond], x fp32 [n] -> (x_prev, pred_x0), ddim.py:253-302 (sigma = 0)."""
    assert eps2.dtype == torch.float32 and x.dtype == torch.float32 and eps2.is_contiguous() and x.is_contiguous()
    n = x.numel()
    assert eps2.numel() == (2 * n if has_uncond else n)
    x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.lib().af_cfg_ddim_step(_p(eps2), _p(x), _p(x_prev), _p(pred_x0), n, int(has_uncond), float(guidance),
                                           float(a_t), float(a_prev), _stream()), "af_cfg_ddim_step")
    return x_prev, pred_x0


def q_sample(x0: torch.Tensor, noise: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    """x_t = sa[b] x0 + sb[b] noise (ddpm.py:395-398); fp32."""
    x0, noise = x0.to(torch.float32).contiguous(), noise.to(torch.float32).contiguous()
    sa, sb = sa.to(torch.float32).contiguous(), sb.to(torch.float32).contiguous()
    B = x0.shape[0]
    xt = torch.empty_like(x0)
    _lib.check(_lib.lib().af_q_sample(_p(x0), _p(noise), _p(sa), _p(sb), _p(xt), B, x0.numel() // B, _stream()), "af_q_sample")
    return xt


# ----------------------------------------------------------------------------- backward ops
def groupnorm_train(x, gamma, beta, eps, silu, *, x2=None, groups=32):
    """groupnorm() that also returns the (mean, rstd) statistics fp32 [B, groups, 2] for the backward."""
    _chk_f16_raw(x, "groupnorm.x")
    B, c1 = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c1)
    c2 = 0 if x2 is None else x2.shape[-1]
    y = torch.empty(tuple(x.shape[:-1]) + (c1 + c2,), dtype=F16, device=x.device)
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    if _pending:
        pr = _take_pending(x, B, hw, c1) if x2 is None else flush_pending(x.device)
        if pr is not None and _groupnorm_splitk(pr, x, gamma, beta, y, stats, B, hw, c1, groups, eps, silu):
            return y, stats
    gn = partials_of(x) if x2 is None else None
    if gn is not None and gn.B == B and gn.hw == hw and gn.C == c1 and gn.cpg * groups == c1 and x.is_contiguous():
        rc = _lib.lib().af_groupnorm_apply(_p(x), c1, _p(gamma), _p(beta), _p(y), _p(stats), B, hw, groups, float(eps), int(silu), _p(gn.ws), gn.nblk, _stream())
        _lib.check(rc, "af_groupnorm_apply")
        return y, stats
    rc = _lib.lib().af_groupnorm_stats(_p(x), _p(x2), c1, c2, _p(gamma), _p(beta), _p(y), _p(stats), B, hw, groups, float(eps),
                                       int(silu), _p(_gn_workspace(x.device, B)), _stream())
    _lib.check(rc, "af_groupnorm_stats")
    return y, stats


def groupnorm_bwd(x, gamma, beta, stats, dy, silu, *, x2=None, add=None, groups=32):
    """Input gradient of groupnorm(+SiLU): returns dx (or (dx1, dx2) when x2 is given)."""
    B, c1 = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c1)
    c2 = 0 if x2 is None else x2.shape[-1]
    _chk_f16(dy, "groupnorm_bwd.dy")
    dx1 = torch.empty_like(x)
    dx2 = None if x2 is None else torch.empty_like(x2)
    rc = _lib.lib().af_groupnorm_bwd(_p(x), _p(x2), c1, c2, _p(gamma), _p(beta), _p(stats), _p(dy), _p(add), _p(dx1), _p(dx2), B,
                                     hw, groups, int(silu), _p(_gn_workspace(x.device, B)), _stream())
    _lib.check(rc, "af_groupnorm_bwd")
    return dx1 if x2 is None else (dx1, dx2)


def layernorm_bwd(x, gamma, dy, eps=1e-5, add=None):
    dx = torch.empty_like(x)
    Cn = x.shape[-1]
    _lib.check(_lib.lib().af_layernorm_bwd(_p(x), _p(gamma), _p(dy), _p(add), _p(dx), x.numel() // Cn, Cn, float(eps), _stream()),
               "af_layernorm_bwd")
    return dx


def layernorm_param_grads(x, dy, eps=1e-5):
    """(dgamma, dbeta) fp32 [C] of a LayerNorm over the last dim from its input and output gradient (rows <= 2048, C <= 1536)."""
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    out = torch.empty((2 * Cn + 2 * rows,), dtype=torch.float32, device=x.device)        # dgamma | dbeta | row statistics scratch
    dg, db, st = out[:Cn], out[Cn:2 * Cn], out[2 * Cn:]
    _lib.check(_lib.lib().af_layernorm_param_grads(_p(x), _p(dy), _p(dg), _p(db), _p(st), rows, Cn, float(eps), _stream()),
               "af_layernorm_param_grads")
    return dg, db


def geglu_fwd(hp):
    M, two_i = hp.shape
    out = torch.empty((M, two_i // 2), dtype=F16, device=hp.device)
    _lib.check(_lib.lib().af_geglu_fwd(_p(hp), _p(out), M, two_i // 2, _stream()), "af_geglu_fwd")
    return out


def geglu_bwd(hp, dout):
    dhp = torch.empty_like(hp)
    _lib.check(_lib.lib().af_geglu_bwd(_p(hp), _p(dout), _p(dhp), hp.shape[0], hp.shape[1] // 2, _stream()), "af_geglu_bwd")
    return dhp


def sumpool2x2(x):
    B, H2, W2, Cn = x.shape
    y = torch.empty((B, H2 // 2, W2 // 2, Cn), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_sumpool2x2(_p(x), _p(y), B, H2 // 2, W2 // 2, Cn, _stream()), "af_sumpool2x2")
    return y


def add(a, b):
    _chk_f16(a, "add.a")
    _chk_f16(b, "add.b")
    assert a.shape == b.shape
    out = torch.empty_like(a)
    _lib.check(_lib.lib().af_add_f16(_p(a), _p(b), _p(out), a.numel(), _stream()), "af_add_f16")
    return out


def axpy(a, b, alpha):
    """a + alpha * b (fp16)."""
    _chk_f16(a, "axpy.a")
    _chk_f16(b, "axpy.b")
    assert a.shape == b.shape
    out = torch.empty_like(a)
    _lib.check(_lib.lib().af_axpy_f16(_p(a), _p(b), float(alpha), _p(out), a.numel(), _stream()), "af_axpy_f16")
    return out


def transpose_tokens(x, B, N, Cn, ldx):
    """x: rows of a [B*N, ldx] token tensor (Cn columns from x's first column) -> [B, Cn, roundup(N, 8)]."""
    ldy = round_up(N, 8)
    y = torch.empty((B, Cn, ldy), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_transpose_tokens(_p(x), _p(y), B, N, Cn, ldx, ldy, _stream()), "af_transpose_tokens")
    return y


def cadamw_step(p, g, m, v, seg_offsets, counts, *, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, step=1, correct_bias=True):
    """Fused cautious-AdamW step (c_adamw.py:65-123) on flat fp32 buffers; seg_offsets int64 [nseg+1] on the device."""
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous()
    rc = _lib.lib().af_cadamw_step(_p(p), _p(g), _p(m), _p(v), _p(seg_offsets), seg_offsets.numel() - 1, _p(counts), float(lr),
                                   float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), int(correct_bias),
                                   _stream())
    _lib.check(rc, "af_cadamw_step")


def colsum(a, b=None, out=None, accumulate=False):
    """fp32 [C] = sum over rows of a (* b): bias / LayerNorm-gamma gradients."""
    rows, Cn = a.shape
    if out is None:
        out = torch.empty((Cn,), dtype=torch.float32, device=a.device)
    if rows >= 2048:                                           # tall: spread the rows over workgroups (two deterministic passes)
        chunks = min(256, (rows + 255) // 256)
        partial = torch.empty((chunks, Cn), dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().af_colsum_tall(_p(a), _p(b), _p(partial), _p(out), rows, Cn, chunks, int(accumulate), _stream()), "af_colsum_tall")
        return out
    _lib.check(_lib.lib().af_colsum(_p(a), _p(b), _p(out), rows, Cn, int(accumulate), _stream()), "af_colsum")
    return out


def quickgelu_fwd(x):
    y = torch.empty_like(x)
    _lib.check(_lib.lib().af_quickgelu_fwd(_p(x), _p(y), x.numel(), _stream()), "af_quickgelu_fwd")
    return y


def quickgelu_bwd(x, dy):
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().af_quickgelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "af_quickgelu_bwd")
    return dx


def clamp_f32_(a, lo, hi):
    _lib.check(_lib.lib().af_clamp_f32(_p(a), float(lo), float(hi), a.numel(), _stream()), "af_clamp_f32")
    return a


def scale_f32_(a, s):
    _lib.check(_lib.lib().af_scale_f32(_p(a), float(s), a.numel(), _stream()), "af_scale_f32")
    return a


def softmax_rows(x):
    """Row softmax of an fp16 [rows, L] score matrix (fp32 inside)."""
    _chk_f16(x, "softmax_rows.x")
    rows, L = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().af_softmax_rows(_p(x), _p(y), rows, L, _stream()), "af_softmax_rows")
    return y


def softmax_rows_bwd(p, dp):
    """ds = p * (dp - rowsum(p * dp)) for fp16 [rows, L] probabilities and their gradient."""
    _chk_f16(p, "softmax_rows_bwd.p")
    _chk_f16(dp, "softmax_rows_bwd.dp")
    assert p.shape == dp.shape
    ds = torch.empty_like(p)
    _lib.check(_lib.lib().af_softmax_rows_bwd(_p(p), _p(dp), _p(ds), p.shape[0], p.shape[1], _stream()), "af_softmax_rows_bwd")
    return ds


def mask_pairs_(p, cls):
    """In place: p [N, N] fp16 *= ((cls[i] & cls[j]) != 0); cls uint8 [N] (bit 0 foreground, bit 1 background)."""
    assert p.shape[0] == p.shape[1] == cls.numel() and cls.dtype == torch.uint8
    _lib.check(_lib.lib().af_mask_pairs(_p(p), _p(cls), p.shape[0], _stream()), "af_mask_pairs")
    return p


# ----------------------------------------------------------------------------- trainable DoRA adapters
def dora_combine(y0, c2, lb, u, v):
    """y0 + u[c] * c2 + v[c] * lb over the last (channel) axis; u, v fp32 [C]."""
    Cn = y0.shape[-1]
    out = torch.empty_like(y0)
    _lib.check(_lib.lib().af_dora_combine(_p(y0), _p(c2), _p(lb), _p(u), _p(v), _p(out), y0.numel() // Cn, Cn, _stream()), "af_dora_combine")
    return out


def mul(a, b):
    _chk_f16(a, "mul.a")
    _chk_f16(b, "mul.b")
    assert a.shape == b.shape
    out = torch.empty_like(a)
    _lib.check(_lib.lib().af_mul_f16(_p(a), _p(b), _p(out), a.numel(), _stream()), "af_mul_f16")
    return out


def im2col3x3(x, stride=1):
    """x [B,H,W,C] fp16 -> [B*Ho*Wo, 9*C], column order (ky, kx, c)."""
    B, H, W, Cn = x.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out = torch.empty((B * Ho * Wo, 9 * Cn), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_im2col3x3(_p(x), _p(out), B, H, W, Cn, stride, _stream()), "af_im2col3x3")
    return out


# ----------------------------------------------------------------------------- ArcFace ResNetFace encoder
def affine_prelu(x, scale=None, shift=None, slope=None):
    """y = prelu(x * scale[c] + shift[c]) over the last (channel) axis; any of the two stages may be absent."""
    Cn = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.lib().af_affine_prelu(_p(x), _p(scale), _p(shift), _p(slope), _p(y), x.numel() // Cn, Cn, _stream()), "af_affine_prelu")
    return y


def maxpool2x2(x):
    B, H2, W2, Cn = x.shape
    y = torch.empty((B, H2 // 2, W2 // 2, Cn), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_maxpool2x2(_p(x), _p(y), B, H2 // 2, W2 // 2, Cn, _stream()), "af_maxpool2x2")
    return y


def global_avgpool(x):
    B, H, W, Cn = x.shape
    out = torch.empty((B, Cn), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_global_avgpool(_p(x), _p(out), B, H * W, Cn, _stream()), "af_global_avgpool")
    return out


def se_residual_prelu(x, se_logits, residual, slope):
    B, H, W, Cn = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().af_se_residual_prelu(_p(x), _p(se_logits), _p(residual), _p(slope), _p(y), B, H * W, Cn, _stream()),
               "af_se_residual_prelu")
    return y


def affine_prelu_bwd(dy, x=None, scale=None, shift=None, slope=None):
    """Input gradient of ``affine_prelu``; x (the forward input) is only read when there is a PReLU."""
    _chk_f16(dy, "affine_prelu_bwd.dy")
    Cn = dy.shape[-1]
    dx = torch.empty_like(dy)
    _lib.check(_lib.lib().af_affine_prelu_bwd(_p(x), _p(scale), _p(shift), _p(slope), _p(dy), _p(dx), dy.numel() // Cn, Cn, _stream()),
               "af_affine_prelu_bwd")
    return dx


def maxpool2x2_bwd(x, dy):
    _chk_f16(dy, "maxpool2x2_bwd.dy")
    B, H2, W2, Cn = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().af_maxpool2x2_bwd(_p(x), _p(dy), _p(dx), B, H2 // 2, W2 // 2, Cn, _stream()), "af_maxpool2x2_bwd")
    return dx


def se_gate_grad(x, se_logits, residual, slope, dy):
    """-> fp16 [B, C]: gradient of the SE logits divided by H*W (see include/adaface_hip.h)."""
    _chk_f16(dy, "se_gate_grad.dy")
    B, H, W, Cn = x.shape
    dgl = torch.empty((B, Cn), dtype=F16, device=x.device)
    _lib.check(_lib.lib().af_se_gate_grad(_p(x), _p(se_logits), _p(residual), _p(slope), _p(dy), _p(dgl), B, H * W, Cn, _stream()),
               "af_se_gate_grad")
    return dgl


def se_residual_prelu_bwd(x, se_logits, residual, slope, dy, dpool=None):
    """-> (dx, dresidual) of ``se_residual_prelu``; dpool [B, C] is the squeeze branch's gradient (1/HW included)."""
    _chk_f16(dy, "se_residual_prelu_bwd.dy")
    B, H, W, Cn = x.shape
    dx, dres = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.lib().af_se_residual_prelu_bwd(_p(x), _p(se_logits), _p(residual), _p(slope), _p(dy), _p(dpool), _p(dx), _p(dres), B,
                                                   H * W, Cn, _stream()), "af_se_residual_prelu_bwd")
    return dx, dres


# ----------------------------------------------------------------------------- profiling hook
def prof_enable(on: bool):
    _lib.check(_lib.lib().af_prof_enable(int(on)), "af_prof_enable")


def prof_reset():
    _lib.check(_lib.lib().af_prof_reset(), "af_prof_reset")


def prof_read(family: int):
    n, ms = C.c_int(0), C.c_double(0.0)
    _lib.check(_lib.lib().af_prof_read(family, C.byref(n), C.byref(ms)), "af_prof_read")
    return n.value, ms.value
