// af_common.h -- shared device/host helpers for libadaface_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/adaface_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4_t __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define AF_WAVE 64

// ---- host side: error reporting + launch bracketing -------------------------------------
void af_set_error(const std::string& msg);
int af_fail(int code, const std::string& msg);

// RAII bracket used by every launcher: records hipEvents around the launch when profiling
// of `family` is enabled (bench.py roofline leg), and turns a failed launch into AF_E_HIP.
struct AfLaunchScope {
  int family;
  hipStream_t stream;
  hipEvent_t ev_stop;      // the scope owns its stop event handle (no index into shared state); nullptr = not recording
  unsigned epoch;          // af_prof_reset() generation the pair belongs to
  AfLaunchScope(int family, void* stream);
  ~AfLaunchScope();
};
int af_check_launch(const char* what);
// Dynamic LDS above the 64 KB default needs hipFuncAttributeMaxDynamicSharedMemorySize once per kernel.  `done` is the caller's static flag (set only on
// success).  A refusal is remembered per thread and reported by the next af_check_launch(); callers skip the launch when this returns false.
bool af_allow_dyn_lds(const void* kernel, size_t bytes, bool& done, const char* what);
// af_bwd.hip: [B, N, C (row stride ldx)] -> [B, C, ldy], token index contiguous, tokens N .. ldy zero-filled (16-byte accesses when aligned)
void af_launch_transpose_tokens(const _Float16* x, _Float16* y, int B, int N, int C, int ldx, int ldy, hipStream_t stream);
struct AfTransposeJob {
  const _Float16* x;
  _Float16* y;
  int N, C, ldx, ldy;
};
// up to three such transposes (same batch count) in one launch when all qualify for the 16-byte form
void af_launch_transpose_tokens_multi(const AfTransposeJob* jobs, int n, int B, hipStream_t stream);

#define AF_REQUIRE(cond, msg)                                      \
  do {                                                             \
    if (!(cond)) return af_fail(AF_E_BADARG, std::string(msg));    \
  } while (0)
#define AF_SUPPORTED(cond, msg)                                      \
  do {                                                               \
    if (!(cond)) return af_fail(AF_E_UNSUPPORTED, std::string(msg)); \
  } while (0)

// ---- device helpers ------------------------------------------------------------------------
// sigmoid with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 VALU instructions): every caller rounds the result to fp16.
// The element-wise kernels are VALU-co-bound, not purely HBM-bound: GroupNorm + SiLU [8, 256, 1280] 10.2 -> 6.7 us, [8, 4096, 320] 22.2 -> 19.8 us,
// its backward [4, 4096, 640+320] 87 -> 76 us (tools/bench_gn.py, tools/bench_gn_bwd.py, same box).
__device__ __forceinline__ float af_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float af_silu(float x) { return x * af_sigmoid(x); }
// exact (erf) GELU as F.gelu default (attention.py:38)
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of the result): branch-free, 9 VALU + rcp +
// exp2 -- the libm erff is ~45 VALU with two divergent ranges, which made the GEGLU epilogue cost more than its tile's MFMAs.
__device__ __forceinline__ float af_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float af_gelu_erf(float x) { return 0.5f * x * (1.0f + af_erf(x * 0.70710678118654752f)); }

__device__ __forceinline__ float af_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float af_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- GroupNorm partial statistics (round 5) --------------------------------------------------------------------------------------------------
// A partial is (sum, M2) of a block of n values, M2 = sum of squares ABOUT THE BLOCK'S OWN MEAN.  Blocks are merged pairwise (Chan et al.):
//     M2 = M2_a + M2_b + (mean_b - mean_a)^2 n_a n_b / (n_a + n_b)
// so nothing of the form E[x^2] - mean^2 is ever formed across blocks: that difference loses log2((mean / sigma)^2) bits, which in fp32 is every bit of
// the variance of a group with |mean| / sigma ~ 1000 and a 10 % error at 100 (the reference is torch's fp32 group_norm, util.py:195-212, which
// does not cancel).  A producer gets its (sum, M2) from sums SHIFTED by a pivot p taken inside the block (any element: |p - mean| is a few sigma):
//     s' = sum(x - p), q' = sum((x - p)^2)  ->  sum = n p + s',  M2 = q' - s'^2 / n.
struct GnAcc {
  float n, s, m2;
};
__device__ __forceinline__ GnAcc gn_acc_merge(const GnAcc a, const GnAcc b) {
  if (b.n <= 0.f) return a;
  if (a.n <= 0.f) return b;
  const float n = a.n + b.n;
  const float d = b.s * __builtin_amdgcn_rcpf(b.n) - a.s * __builtin_amdgcn_rcpf(a.n);
  GnAcc r;
  r.n = n;
  r.s = a.s + b.s;
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n * __builtin_amdgcn_rcpf(n));
  return r;
}
// shifted sums of n values about pivot p -> (sum, M2)
__device__ __forceinline__ GnAcc gn_acc_from_shifted(float n, float p, float s1, float q1) {
  GnAcc r;
  r.n = n;
  r.s = n * p + s1;
  r.m2 = n > 0.f ? fmaxf(q1 - s1 * s1 * __builtin_amdgcn_rcpf(n), 0.f) : 0.f;
  return r;
}

// Packed-fp16 shifted difference for the statistics kernels: HALF of (x - p), formed as 0.5 x + (-0.5 p).  x - p itself overflows fp16 when an
// outlier and the pivot have opposite signs and |x - p| > 65504 (inf -> M2 = inf - inf = NaN -> the whole group NaN; the SD fp16 VAE carries
// activations in the 1e4 range), the halved form cannot.  Halving is exact for normal fp16 values, so 0.5 x - 0.5 p rounds to the same mantissa as
// x - p: callers accumulate s' = sum(d), q' = sum(d^2) of the halves and scale by 2 and 4 (exact in fp32) at the end -- bit-identical to the
// un-halved form wherever that one was finite.  nhp = -0.5 p, prepared once per pivot.
__device__ __forceinline__ half2_t gn_half_diff(const half2_t x, const half2_t nhp) {
  const half2_t h = {(half_t)0.5f, (half_t)0.5f};
  return x * h + nhp;
}

// ---- weight-tile prefetch into the XCD's L2 (used by the GEMM kernels at kernel start) ---------------------------------------------------
// Inside a denoise / training step every GEMM meets its weights cold in HBM (1.72 GB of weights are read once per U-Net pass) while
// the operand pipelines request a K stage only ~one stage (~0.3 us) before it is needed, so the first workgroup of an XCD to touch a
// weight line pays the HBM latency at EVERY stage (profiles/r01w_cold_operands.txt: 7.99 ms with hot operands vs 9.41 ms with cold
// weights per step).  The workgroups that share a weight tile (same tile_n, consecutive tile_m: neighbours on one XCD under the
// XCD-aware tile mapping) each touch a 1 / coop share of its 128-byte lines once, right at kernel start: one dword per line, 64 lines
// per wave instruction, earliest K stages first, at most `budget` instructions per wave.  The loads are older than the first
// stage's operand loads, so that stage's wait covers them.  Their data is never used: they are LDS-DMA loads (no destination VGPR) into
// a 256-byte dump area of the kernel's LDS that nothing reads.  (The first form loaded into "sink" VGPRs by inline asm; the compiler does
// not know such a load completes later, so under register pressure it could spill the sink and re-use the register before the data
// landed -- the folded-LayerNorm 256 x 320 GEGLU tile faulted that way, tools/probes/r03ae_ln_geglu_fault.py.)  Measured: 12.31 -> 11.61 ms
// per denoise step (profiles/r03h_weight_prefetch.txt).
constexpr int AF_WPF_MAX = 8;
constexpr int AF_WPF_DUMP_BYTES = 256;                             // LDS the caller sets aside: 64 lanes x 4 bytes
__device__ __forceinline__ void af_prefetch_weight_tile(const half_t* wt, int kpad, int npad, int row0, int rows, int kt0, int nk, int coop, int me,
                                                        int budget, int nw, int wave, int lane, char* lds_dump) {
  const int total = rows * nk;                                   // 128-byte lines (64 halves) of these weight rows over this K range
  const int per = (total + coop - 1) / coop;
  const int begin = me * per;
  const int end = min(total, begin + min(per, budget * nw * 64));
#pragma unroll
  for (int j = 0; j < AF_WPF_MAX; ++j) {
    const int line = begin + (j * nw + wave) * 64 + lane;
    if (j < budget && line < end) {
      const int st = line / rows, row = line - st * rows;        // stage-major: the first lines cover stage 0 of every row
      const int n = row0 + row;
      if (n < npad)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wt + (size_t)n * kpad + (size_t)(kt0 + st) * 64),
                                         (__attribute__((address_space(3))) void*)lds_dump, 4, 0, 0);
    }
  }
}
