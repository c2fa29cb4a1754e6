"""ctypes binding of csrc/libadaface_hip.so (C ABI declared in include/adaface_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  ``build()`` (re)compiles the library in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# AF_LIB: another build of the SAME sources to load instead (measurement only: tools/ab_lib.sh compares two builds of the library, e.g. one compiled
# with -DAF_GEMM3W_DIET=0, in alternating runs on one box).  Unset in every judged run; a missing file fails as loudly as the default path.
LIB_PATH = os.environ.get("AF_LIB") or os.path.join(CSRC, "libadaface_hip.so")

AF_OK, AF_E_BADARG, AF_E_UNSUPPORTED, AF_E_HIP = 0, -1, -2, -3
AF_ACT_NONE, AF_ACT_SILU, AF_ACT_GEGLU, AF_ACT_QUICKGELU = 0, 1, 2, 3
AF_OUT_NORMAL, AF_OUT_SPLIT_T, AF_OUT_F32 = 0, 1, 2
AF_SPLITK_COUNTER_BYTES = 4096 * 4
AF_FAM_GEMM, AF_FAM_ATTN, AF_FAM_GNORM, AF_FAM_LNORM, AF_FAM_ELEM, AF_FAM_XATTN = 0, 1, 2, 3, 4, 5

# every symbol include/adaface_hip.h declares (tests check the .so exports exactly these)
EXPORTS = (
    "af_last_error", "af_version", "af_device_count", "af_prof_enable", "af_prof_reset", "af_prof_read",
    "af_gemm", "af_groupnorm_ws_floats", "af_groupnorm", "af_layernorm", "af_attention", "af_attention_scores",
    "af_timestep_embedding", "af_nchw_f32_to_nhwc_f16", "af_nhwc_f16_to_nchw_f32", "af_cfg_ddim_step", "af_q_sample",
    "af_silu_f16", "af_attention_lse", "af_groupnorm_stats", "af_attention_bwd_scratch_bytes", "af_attention_bwd",
    "af_groupnorm_bwd", "af_layernorm_bwd", "af_geglu_fwd", "af_geglu_bwd", "af_sumpool2x2", "af_add_f16",
    "af_transpose_tokens", "af_cadamw_step", "af_attention_ex", "af_colsum", "af_quickgelu_fwd", "af_quickgelu_bwd",
    "af_scale_f32", "af_affine_prelu", "af_maxpool2x2", "af_global_avgpool", "af_se_residual_prelu", "af_axpy_f16", "af_dora_combine", "af_mul_f16", "af_im2col3x3", "af_colsum_tall", "af_softmax_rows", "af_attention_strided", "af_clamp_f32", "af_mask_pairs", "af_prefetch", "af_prefetch_ex",
    "af_xattn_scores", "af_xattn_softmax_pv", "af_xattn_softmax_pv_bwd", "af_xattn_rowmix", "af_xattn_colmix_ws_bytes", "af_xattn_colmix",
    "af_layernorm_param_grads", "af_transpose_tokens_pair", "af_ff_fused", "af_ff_chain", "af_xattn_fused",
    "af_softmax_rows_bwd", "af_affine_prelu_bwd", "af_maxpool2x2_bwd", "af_se_gate_grad", "af_se_residual_prelu_bwd",
    "af_groupnorm_apply", "af_gemm_gn_stats_ok", "af_gemm_halo_variant", "af_gn_proj_fused",
    "af_splitk_reduce", "af_groupnorm_splitk_ok", "af_groupnorm_splitk", "af_xattn_chain",
)


class GemmDesc(C.Structure):
    _fields_ = [
        ("a1", C.c_void_p), ("a2", C.c_void_p), ("wt", C.c_void_p), ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("kpad", C.c_int32), ("taps", C.c_int32),
        ("c1", C.c_int32), ("c2", C.c_int32), ("lda1", C.c_int32), ("lda2", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("rows_per_batch", C.c_int32), ("ld_rowbias", C.c_int32),
        ("act", C.c_int32), ("out_mode", C.c_int32), ("ld_out", C.c_int32), ("split_col", C.c_int32),
        ("ld_out2", C.c_int32), ("tile", C.c_int32), ("splits", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("zeros", C.c_void_p),
        ("tap_shift", C.c_int32), ("splitk_fused", C.c_int32),
        ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("a3", C.c_void_p), ("a4", C.c_void_p), ("c3", C.c_int32), ("c4", C.c_int32), ("lda3", C.c_int32), ("lda4", C.c_int32),
        ("gn_partials", C.c_void_p), ("gn_cpg", C.c_int32),
        ("defer_reduce", C.POINTER(C.c_int32)),
    ]


def _csrc_sha() -> str:
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    files += [os.path.join(CSRC, "Makefile"), os.path.join(os.path.dirname(_HERE), "include", "adaface_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into csrc/libadaface_hip.so (make, in-tree).  The objects are stamped with a hash
    of ALL sources: if the stamp does not match the tree (a stale or foreign .o / .so), everything is rebuilt from scratch;
    otherwise make's incremental rules decide."""
    stamp = os.path.join(CSRC, ".build_sha")
    want = _csrc_sha()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    cmd = ["make", "-C", CSRC, "-j8"] + ([] if have == want and os.path.exists(LIB_PATH) else ["-B"])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libadaface_hip.so failed:\n" + r.stdout + r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    if verbose:
        print(r.stdout)
    return LIB_PATH


def sources_sha() -> str:
    """Hash of everything that decides what the GEMM family executes (kernel sources + the tuned tile table).  Measurements
    that cannot be taken in-process (PMC traffic) record it, and bench.py reports them only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("csrc/af_common.h", "csrc/af_gemm.hip", "csrc/af_gemm3.hip", "csrc/af_runtime.hip", "tuning/gfx950_gemm.json"):
        with open(os.path.join(_HERE, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raise loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be the HIP runtime of this process, so load it
    # first (the .so then binds to the already-loaded runtime instead of a second copy).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C adaface-dev_amd/csrc`."
        )
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.af_last_error.restype = C.c_char_p
    L.af_last_error.argtypes = []
    L.af_version.argtypes = []
    L.af_device_count.argtypes = []
    L.af_prof_enable.argtypes = [i32]
    L.af_prof_reset.argtypes = []
    L.af_prof_read.argtypes = [i32, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.af_gemm.argtypes = [C.POINTER(GemmDesc), vp]
    L.af_groupnorm_ws_floats.argtypes = [i32]
    L.af_groupnorm.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]
    L.af_groupnorm_apply.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, i32, vp]
    L.af_gemm_gn_stats_ok.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32]
    L.af_gemm_halo_variant.argtypes = [C.POINTER(GemmDesc)]
    L.af_splitk_reduce.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, i32, i32, vp]
    L.af_groupnorm_splitk_ok.argtypes = [i32, i32, i32, i32]
    L.af_groupnorm_splitk.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]
    L.af_gn_proj_fused.argtypes = [vp, vp, i32, vp, vp, f32, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp]
    L.af_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp]
    L.af_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]
    L.af_attention_scores.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    L.af_timestep_embedding.argtypes = [vp, vp, i32, i32, f32, vp]
    L.af_nchw_f32_to_nhwc_f16.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.af_nhwc_f16_to_nchw_f32.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.af_cfg_ddim_step.argtypes = [vp, vp, vp, vp, i64, i32, f32, f32, f32, vp]
    L.af_q_sample.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp]
    L.af_silu_f16.argtypes = [vp, vp, i64, vp]
    L.af_attention_lse.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]
    L.af_attention_ex.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]
    L.af_attention_strided.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, f32, vp]
    L.af_groupnorm_stats.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]
    L.af_attention_bwd_scratch_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.af_attention_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i64] + [i32] * 14 + [f32, vp]
    L.af_axpy_f16.argtypes = [vp, vp, f32, vp, i64, vp]
    L.af_dora_combine.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, vp]
    L.af_mul_f16.argtypes = [vp, vp, vp, i64, vp]
    L.af_im2col3x3.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.af_softmax_rows.argtypes = [vp, vp, i64, i32, vp]
    L.af_mask_pairs.argtypes = [vp, vp, i32, vp]
    L.af_prefetch.argtypes = [vp, i64, vp]
    L.af_prefetch_ex.argtypes = [vp, i64, i32, vp]
    L.af_xattn_scores.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, f32, vp]
    L.af_xattn_softmax_pv.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.af_xattn_softmax_pv_bwd.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]
    L.af_xattn_rowmix.argtypes = [vp, vp, i32, vp, i32, f32, i32, i32, i32, i32, i32, vp]
    L.af_xattn_colmix_ws_bytes.argtypes = [i32, i32, i32, i32]
    L.af_xattn_colmix.argtypes = [vp, vp, i32, vp, i32, f32, vp, i64, i32, i32, i32, i32, i32, vp]
    L.af_colsum.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.af_colsum_tall.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.af_quickgelu_fwd.argtypes = [vp, vp, i64, vp]
    L.af_quickgelu_bwd.argtypes = [vp, vp, vp, i64, vp]
    L.af_scale_f32.argtypes = [vp, f32, i64, vp]
    L.af_clamp_f32.argtypes = [vp, f32, f32, i64, vp]
    L.af_affine_prelu.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp]
    L.af_maxpool2x2.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.af_global_avgpool.argtypes = [vp, vp, i32, i32, i32, vp]
    L.af_se_residual_prelu.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.af_softmax_rows_bwd.argtypes = [vp, vp, vp, i64, i32, vp]
    L.af_affine_prelu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, vp]
    L.af_maxpool2x2_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.af_se_gate_grad.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.af_se_residual_prelu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.af_groupnorm_bwd.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.af_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, vp]
    L.af_layernorm_param_grads.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, vp]
    L.af_geglu_fwd.argtypes = [vp, vp, i64, i32, vp]
    L.af_geglu_bwd.argtypes = [vp, vp, vp, i64, i32, vp]
    L.af_sumpool2x2.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.af_add_f16.argtypes = [vp, vp, vp, i64, vp]
    L.af_transpose_tokens.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.af_transpose_tokens_pair.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp]
    L.af_cadamw_step.argtypes = [vp, vp, vp, vp, vp, i32, vp, f32, f32, f32, f32, f32, i32, i32, vp]
    L.af_ff_fused.argtypes = [vp, vp, vp, vp, f32, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp]
    L.af_ff_chain.argtypes = [vp, vp, vp, vp, f32, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.af_xattn_fused.argtypes = [vp, vp, vp, vp, f32, i32, vp, i32, vp, i64, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp]
    L.af_xattn_chain.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, f32, i32, vp, i32, vp, i64, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, vp, vp]
    for name in EXPORTS:
        if name != "af_last_error":
            getattr(L, name).restype = C.c_int
    L.af_attention_bwd_scratch_bytes.restype = C.c_int64
    L.af_xattn_colmix_ws_bytes.restype = C.c_int64
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc < 0:
        msg = lib().af_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
