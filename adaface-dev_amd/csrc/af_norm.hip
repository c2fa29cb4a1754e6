// af_norm.hip -- GroupNorm(32)[+SiLU] over NHWC and LayerNorm over the channel dim.
//
// Both are HBM-bound (45.06 M GroupNorm and 34.65 M LayerNorm elements per
// sample).  Design for gfx950:
//   * every global access is a 16-byte (8 x fp16) chunk; a thread keeps the SAME channel chunk
//     for every pixel it visits, so per-channel scale/shift (and the per-channel partial sums
//     of the statistics pass) live in registers and the inner loop is load-fma-(silu)-store;
//   * GroupNorm statistics are per (batch, group) over HW x C/32 elements that are strided in
//     NHWC: pass 1 writes deterministic per-block partial sums (no float atomics), pass 2
//     folds them (a few KB, L2-resident) and normalises; the activation tensor of one layer
//     (<= 21 MB at batch 8) stays in the 256 MB Infinity Cache between the two passes;
//   * the channel concat of the U-Net skip connections (openaimodel.py:918) is fused: the
//     input is read from two sources, the output is one tensor.
#include <stdlib.h>

#include "af_common.h"

namespace {

constexpr int GN_NBLK = 128;   // max statistic partial blocks per batch item (the launch uses a.nblk <= GN_NBLK)
constexpr int GN_MAXG = 32;

struct GnArgs {
  const half_t* x1;
  const half_t* x2;
  int c1, c2, C, CP;  // CP = C / 8 chunks per pixel
  const float* gamma;
  const float* beta;
  half_t* y;
  int B, HW, groups, cpg;
  float eps;
  int silu;
  float* ws;  // [B][GN_NBLK][GN_MAXG][2]
  float* stats;  // optional out [B][groups][2] = (mean, rstd)
  int ppb;    // pixel slots per iteration (CT == 1)
  int nblk;   // partial blocks per batch item in this launch
  // split-K producer folded in (af_groupnorm_splitk; the SLAB instantiations of the one-launch kernels): x1 is not read.  The value at (row, c) is
  // fp16(sum_sp slab[sp][row][c] + bias[c] + rowbias[b][c] + residual[row][c]) -- af_splitk_reduce_kernel's arithmetic in its order -- stored to
  // xout and normalised.  Slabs are [splits][B * HW][C] fp32, slab_stride = B * HW * C.
  const float* slab;
  size_t slab_stride;
  int splits;
  const float* bias;
  const half_t* rowbias;
  int ld_rowbias;
  const half_t* residual;
  half_t* xout;
};

__device__ __forceinline__ half8_t gn_load(const GnArgs& a, int b, int pix, int c0) {
  const size_t row = (size_t)b * a.HW + pix;
  if (c0 < a.c1) return *reinterpret_cast<const half8_t*>(a.x1 + row * a.c1 + c0);
  return *reinterpret_cast<const half8_t*>(a.x2 + row * a.c2 + (c0 - a.c1));
}

__device__ __forceinline__ float gn_load1(const GnArgs& a, int b, int pix, int c) {
  const size_t row = (size_t)b * a.HW + pix;
  return c < a.c1 ? (float)a.x1[row * a.c1 + c] : (float)a.x2[row * a.c2 + (c - a.c1)];
}

// pass 1: per-block partial (sum, M2 about the block's own mean: af_common.h, GroupNorm partial statistics) per group.  The sums are taken
// SHIFTED by a pivot -- the value at the block's first pixel, the same for every thread of the block, so the per-thread sums still add up --
// and re-based to the group's pivot when the units of a group are folded.  PAIR (cpg even: every SD-1.5 width): the unit is a PAIR of
// adjacent channels sharing the first one's pivot, and the arithmetic is packed fp16 + v_dot2_f32_f16 -- d2 = x2 - p2 (exact whenever x is
// within a factor 2 of the pivot, i.e. exactly where the shift matters; rounded to 11 bits of d otherwise), s += d2 . (1, 1), q += d2 . d2:
// 1.5 instructions per pair of elements where the unshifted per-channel form took 6 (the pass is VALU-co-bound, af_common.h).
template <int CT, bool PAIR>
__global__ __launch_bounds__(256) void gn_partial_kernel(GnArgs a) {
  constexpr int U = PAIR ? 2 : 1, NS = 8 / U;         // channels per unit, units per 8-channel chunk
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  float* red = reinterpret_cast<float*>(af_smem);  // [2][slots][C / U]
  const int t = threadIdx.x, b = blockIdx.y, blk = blockIdx.x;
  const int slots = CT == 1 ? a.ppb : 1;
  const int slot = CT == 1 ? t / a.CP : 0;
  const int chunk0 = CT == 1 ? t - slot * a.CP : t;
  const bool active = CT == 1 ? (slot < slots) : true;
  const int per = (a.HW + a.nblk - 1) / a.nblk;
  const int p0 = blk * per, p1 = min(a.HW, p0 + per);
  const int CU = a.C / U;                            // units per pixel

  float s[CT][NS], q[CT][NS];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int e = 0; e < NS; ++e) s[j][e] = q[j][e] = 0.f;

  if (active && p0 < p1) {
    half8_t pv[CT];                                    // pivots: this thread's channels at the block's first pixel (PAIR: the even channel's, twice)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ch = chunk0 + 256 * j;
      pv[j] = ch < a.CP ? gn_load(a, b, p0, ch * 8) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      if (PAIR) {
#pragma unroll
        for (int e = 1; e < 8; e += 2) pv[j][e] = pv[j][e - 1];
      }
    }
    // PF pixels per thread in flight: the loop is pure load -> fma, so bytes in flight per CU are what sets the rate
    constexpr int PF = CT == 1 ? 4 : 2;
    const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
    half8_t npv[CT];                                   // -0.5 pivot (PAIR)
#pragma unroll
    for (int j = 0; j < CT; ++j) npv[j] = pv[j] * (half_t)-0.5f;
    for (int pix = p0 + slot; pix < p1; pix += slots * PF) {
      half8_t v[PF][CT];
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int ch = chunk0 + 256 * j;
          const int px = pix + u * slots;
          v[u][j] = (px < p1 && ch < a.CP) ? gn_load(a, b, px, ch * 8) : pv[j];      // out of range: the pivot itself (adds 0 to both sums)
        }
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          if (PAIR) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const half2_t x2 = {v[u][j][2 * e], v[u][j][2 * e + 1]}, np2 = {npv[j][2 * e], npv[j][2 * e + 1]};
              const half2_t d2 = gn_half_diff(x2, np2);           // (x - p) / 2: cannot overflow fp16 (af_common.h); sums rescaled below
              s[j][e] = __builtin_amdgcn_fdot2(d2, one2, s[j][e], false);
              q[j][e] = __builtin_amdgcn_fdot2(d2, d2, q[j][e], false);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float f = (float)v[u][j][e] - (float)pv[j][e];
              s[j][e % NS] += f;
              q[j][e % NS] += f * f;
            }
          }
        }
    }
  }
  float* rs = red;
  float* rq = red + slots * CU;
  if (active) {
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ch = chunk0 + 256 * j;
      if (ch < a.CP) {
#pragma unroll
        for (int e = 0; e < NS; ++e) {
          rs[slot * CU + ch * NS + e] = PAIR ? 2.f * s[j][e] : s[j][e];       // PAIR summed halves of the differences
          rq[slot * CU + ch * NS + e] = PAIR ? 4.f * q[j][e] : q[j][e];
        }
      }
    }
  }
  __syncthreads();
  // fold pixel slots -> per-unit (into slot 0): same pivot in every slot, plain sums
  for (int c = t; c < CU; c += 256) {
    float ss = 0.f, qq = 0.f;
    for (int sl = 0; sl < slots; ++sl) {
      ss += rs[sl * CU + c];
      qq += rq[sl * CU + c];
    }
    rs[c] = ss;
    rq[c] = qq;
  }
  __syncthreads();
  if (t < a.groups) {
    float* w = a.ws + (((size_t)b * GN_NBLK + blk) * GN_MAXG + t) * 2;
    if (p0 >= p1) {                                    // a block past the image (HW < nblk * per): n = 0, the consumer skips it
      w[0] = w[1] = 0.f;
      return;
    }
    // units -> group: re-base unit u's sums from its own pivot p_u to the group's pivot P (that of the group's first channel):
    //   sum(x - P) = s_u + n_u d,  sum((x - P)^2) = q_u + 2 d s_u + n_u d^2,  d = p_u - P,  n_u = pixels x channels of the unit
    const float n = (float)(p1 - p0) * (float)U;
    const float P = gn_load1(a, b, p0, t * a.cpg);
    float ss = 0.f, qq = 0.f;
    for (int u = t * a.cpg / U; u < (t + 1) * a.cpg / U; ++u) {
      const float d = gn_load1(a, b, p0, u * U) - P;
      ss += rs[u] + n * d;
      qq += rq[u] + 2.f * d * rs[u] + n * d * d;
    }
    const GnAcc acc = gn_acc_from_shifted((float)(p1 - p0) * (float)a.cpg, P, ss, qq);
    w[0] = acc.s;
    w[1] = acc.m2;
  }
}

// Fold the partials of batch item b: thread (g = t & 31, kl = t >> 5) merges blocks kl, kl + NL, ...; block k of the launch that wrote them covers
// pixels [k per, min(HW, (k + 1) per)), per = ceil(HW / nblk) (the producing GEMM's 128-row tiles: per = 128).  Returns this thread's share.
__device__ __forceinline__ GnAcc gn_fold_partials(const float* ws_b, int nblk, int HW, int cpg, int g, int kl, int NL) {
  const int per = (HW + nblk - 1) / nblk;
  GnAcc acc = {0.f, 0.f, 0.f};
  for (int k = kl; k < nblk; k += NL) {
    const float* w = ws_b + ((size_t)k * GN_MAXG + g) * 2;
    const int cnt = min(HW, (k + 1) * per) - k * per;
    GnAcc pk;
    pk.n = cnt > 0 ? (float)cnt * (float)cpg : 0.f;
    pk.s = w[0];
    pk.m2 = w[1];
    acc = gn_acc_merge(acc, pk);
  }
  return acc;
}

// pass 2: fold partials, normalise (+ SiLU), write.  The launch is short (two pixel iterations per thread at [8, 4096, 320]), so its
// latency chain matters more than its bytes: the first iteration's loads and the thread's gamma / beta are requested BEFORE the partial
// sums are folded (they do not depend on the statistics), and each iteration's loads are issued ahead of the previous one's arithmetic.
template <int CT>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a) {
  __shared__ float mr[GN_MAXG][2];
  __shared__ float fold[8][GN_MAXG][3];
  const int t = threadIdx.x, b = blockIdx.y;
  const int slots = CT == 1 ? a.ppb : 1;
  const int slot = CT == 1 ? t / a.CP : 0;
  const int chunk0 = CT == 1 ? t - slot * a.CP : t;
  const bool active = !(CT == 1 && slot >= slots);
  const int per = (a.HW + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = min(a.HW, p0 + per);
  constexpr int PF = CT == 1 ? 4 : 2;
  const int step = slots * PF;
  half8_t v[PF][CT];
  auto load = [&](int pix, half8_t (&dst)[PF][CT]) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int ch = chunk0 + 256 * j;
        const int px = pix + u * slots;
        if (px < p1 && ch < a.CP) dst[u][j] = gn_load(a, b, px, ch * 8);
      }
  };
  int pix = p0 + slot;
  floatx4 gm[CT][2], bt[CT][2];
  if (active) {
    if (pix < p1) load(pix, v);
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int c8 = min((chunk0 + 256 * j) * 8, a.C - 8);
      gm[j][0] = *reinterpret_cast<const floatx4*>(a.gamma + c8);
      gm[j][1] = *reinterpret_cast<const floatx4*>(a.gamma + c8 + 4);
      bt[j][0] = *reinterpret_cast<const floatx4*>(a.beta + c8);
      bt[j][1] = *reinterpret_cast<const floatx4*>(a.beta + c8 + 4);
    }
  }
  {
    // 8 lanes of partial blocks x 32 groups: every thread merges nblk / 8 partials, then 8 -> 1 through LDS
    const int g = t & 31, kl = t >> 5;
    GnAcc acc = {0.f, 0.f, 0.f};
    if (g < a.groups) acc = gn_fold_partials(a.ws + (size_t)b * GN_NBLK * GN_MAXG * 2, a.nblk, a.HW, a.cpg, g, kl, 8);
    fold[kl][g][0] = acc.n;
    fold[kl][g][1] = acc.s;
    fold[kl][g][2] = acc.m2;
  }
  __syncthreads();
  if (t < a.groups) {
    GnAcc acc = {fold[0][t][0], fold[0][t][1], fold[0][t][2]};
#pragma unroll
    for (int k = 1; k < 8; ++k) acc = gn_acc_merge(acc, GnAcc{fold[k][t][0], fold[k][t][1], fold[k][t][2]});
    const float inv_n = 1.0f / ((float)a.HW * (float)a.cpg);
    const float mean = acc.s * inv_n;
    const float var = fmaxf(acc.m2 * inv_n, 0.f);
    mr[t][0] = mean;
    mr[t][1] = rsqrtf(var + a.eps);
    if (a.stats && blockIdx.x == 0) {
      a.stats[((size_t)b * a.groups + t) * 2 + 0] = mean;
      a.stats[((size_t)b * a.groups + t) * 2 + 1] = mr[t][1];
    }
  }
  __syncthreads();
  if (!active) return;

  float sc[CT][8], sh[CT][8];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int c8 = min((chunk0 + 256 * j) * 8, a.C - 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (c8 + e) / a.cpg;
      const float k = mr[g][1] * gm[j][e >> 2][e & 3];
      sc[j][e] = k;
      sh[j][e] = bt[j][e >> 2][e & 3] - mr[g][0] * k;
    }
  }
  for (; pix < p1; pix += step) {
    half8_t vn[PF][CT];
    if (pix + step < p1) load(pix + step, vn);
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int ch = chunk0 + 256 * j;
        const int px = pix + u * slots;
        if (px < p1 && ch < a.CP) {
          half8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = (float)v[u][j][e] * sc[j][e] + sh[j][e];
            if (a.silu) f = af_silu(f);
            o[e] = (half_t)f;
          }
          *reinterpret_cast<half8_t*>(a.y + ((size_t)b * a.HW + px) * a.C + ch * 8) = o;
        }
      }
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int j = 0; j < CT; ++j) v[u][j] = vn[u][j];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small-tensor GroupNorm in ONE launch: one workgroup per (batch item, group) holds the group's HW x cpg elements in registers
// (statistics, then normalise: 1 read + 1 write, no workspace, no second launch).  Used for the 16x16 / 8x8 levels, where a
// GroupNorm call is two ~7 us latency-bound launches over a few MB; needs the group's channels to be whole 16-byte chunks
// (cpg % 8 == 0: C = 1280 / 2560) that lie in ONE of the two sources, and HW * cpg / 8 <= 256 * GS_IT chunks.
constexpr int GS_IT = 10;

// SLAB instantiations: IT = items per thread the launch needs (2 at the 8 x 8 level, 5 at 16 x 16, at most GS_IT), UNR = slabs requested together
template <bool SLAB, int IT = GS_IT, int UNR = 1>
__global__ __launch_bounds__(256) void gn_small_kernel(GnArgs a) {
  constexpr int GS_IT = IT;                           // (shadows the file-level bound inside this kernel)
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int cpb = a.cpg >> 3;                          // 16-byte chunks per pixel in this group
  const int c0 = g * a.cpg;
  const half_t* src = c0 < a.c1 ? a.x1 + c0 : a.x2 + (c0 - a.c1);
  const int ld = c0 < a.c1 ? a.c1 : a.c2;
  const int items = a.HW * cpb;
  half8_t v[GS_IT];
  float s = 0.f, q = 0.f;
  // item i = t + 256 u -> (pixel, chunk) = divmod(i, cpb): ONE integer division per thread, then steps of divmod(256, cpb) -- a runtime
  // integer division costs ~25 VALU instructions, as much as the arithmetic of a whole 8-element chunk
  const int pix0 = t / cpb, ch0 = t - pix0 * cpb;
  const int dq = 256 / cpb, dr = 256 - dq * cpb;
  if constexpr (!SLAB) {
    int pix = pix0, ch = ch0;
#pragma unroll
    for (int u = 0; u < GS_IT; ++u) {
      if (t + 256 * u < items) {
        v[u] = *reinterpret_cast<const half8_t*>(src + ((size_t)b * a.HW + pix) * ld + ch * 8);
      } else {
        v[u] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      }
      pix += dq;
      ch += dr;
      if (ch >= cpb) ch -= cpb, ++pix;
    }
  } else {
    // the producer's split-K slabs: every item's two float4 of a slab are requested before any is used (a slab is one memory round trip for the
    // whole thread, not one per item), slabs are added in slice order starting from 0 (0 + s0 == s0), then bias, row bias, residual -- exactly
    // af_splitk_reduce_kernel -- and the fp16 value is stored (the convolution's output) and kept for the statistics
    size_t off[GS_IT];
    floatx4 lo[GS_IT], hi[GS_IT];
    {
      int pix = pix0, ch = ch0;
#pragma unroll
      for (int u = 0; u < GS_IT; ++u) {
        off[u] = t + 256 * u < items ? ((size_t)b * a.HW + pix) * a.C + c0 + ch * 8 : (size_t)b * a.HW * a.C + c0;   // out of range: the group's first chunk (loaded, dropped)
        lo[u] = hi[u] = floatx4{0.f, 0.f, 0.f, 0.f};
        pix += dq;
        ch += dr;
        if (ch >= cpb) ch -= cpb, ++pix;
      }
    }
    // UNR slabs are requested together (one memory round trip per UNR slabs, not per slab: at the 8 x 8 level a launch has 12 - 16 of them) and
    // added in slice order; out-of-range items read item 0's address (always valid) and are dropped below
    int sp = 0;
    for (; sp + UNR <= a.splits; sp += UNR) {
      floatx4 l0[UNR][GS_IT], l1[UNR][GS_IT];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const float* sl = a.slab + (size_t)(sp + k) * a.slab_stride;
#pragma unroll
        for (int u = 0; u < GS_IT; ++u) {
          l0[k][u] = *reinterpret_cast<const floatx4*>(sl + off[u]);
          l1[k][u] = *reinterpret_cast<const floatx4*>(sl + off[u] + 4);
        }
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k)
#pragma unroll
        for (int u = 0; u < GS_IT; ++u) {
          lo[u] += l0[k][u];
          hi[u] += l1[k][u];
        }
    }
    for (; sp < a.splits; ++sp) {
      const float* sl = a.slab + (size_t)sp * a.slab_stride;
      floatx4 l0[GS_IT], l1[GS_IT];
#pragma unroll
      for (int u = 0; u < GS_IT; ++u) {
        l0[u] = *reinterpret_cast<const floatx4*>(sl + off[u]);
        l1[u] = *reinterpret_cast<const floatx4*>(sl + off[u] + 4);
      }
#pragma unroll
      for (int u = 0; u < GS_IT; ++u) {
        lo[u] += l0[u];
        hi[u] += l1[u];
      }
    }
    int ch = ch0;
#pragma unroll
    for (int u = 0; u < GS_IT; ++u) {
      if (t + 256 * u < items) {
        const int c = c0 + ch * 8;
        if (a.bias) {
          lo[u] += *reinterpret_cast<const floatx4*>(a.bias + c);
          hi[u] += *reinterpret_cast<const floatx4*>(a.bias + c + 4);
        }
        if (a.rowbias) {
          const half8_t rv = *reinterpret_cast<const half8_t*>(a.rowbias + (size_t)b * a.ld_rowbias + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[u][e] += (float)rv[e], hi[u][e] += (float)rv[e + 4];
        }
        if (a.residual) {
          const half8_t rv = *reinterpret_cast<const half8_t*>(a.residual + off[u]);
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[u][e] += (float)rv[e], hi[u][e] += (float)rv[e + 4];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = (half_t)lo[u][e], v[u][e + 4] = (half_t)hi[u][e];
        *reinterpret_cast<half8_t*>(a.xout + off[u]) = v[u];
      } else {
        v[u] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      }
      ch += dr;
      if (ch >= cpb) ch -= cpb;
    }
  }
  // sums shifted by a pivot (the group's first element, the same for every thread): q / n - (s / n)^2 then cancels only |pivot - mean| / sigma,
  // not |mean| / sigma (af_common.h, GroupNorm partial statistics)
  float pivot;
  if constexpr (SLAB) {
    // the group's first element of this batch item is item 0's first channel: thread 0 holds it
    if (t == 0) red[0][0] = (float)v[0][0];
    __syncthreads();
    pivot = red[0][0];
    __syncthreads();
  } else {
    pivot = (float)src[(size_t)b * a.HW * ld];
  }
  {
    const half_t nhp = (half_t)pivot * (half_t)-0.5f;
    const half2_t np2 = {nhp, nhp}, one2 = {(half_t)1.0f, (half_t)1.0f};    // packed fp16 + dot2, as gn_partial_kernel
#pragma unroll
    for (int u = 0; u < GS_IT; ++u)
      if (t + 256 * u < items) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const half2_t x2 = {v[u][e], v[u][e + 1]};
          const half2_t d2 = gn_half_diff(x2, np2);
          s = __builtin_amdgcn_fdot2(d2, one2, s, false);
          q = __builtin_amdgcn_fdot2(d2, d2, q, false);
        }
      }
    s *= 2.f;                                           // the sums were taken over (x - p) / 2
    q *= 4.f;
  }
  // gamma / beta of this thread's chunks: requested BEFORE the reduction (they do not depend on the statistics; loaded behind it they were one more
  // dependent memory round trip in a launch that is nothing but a chain of them).  A thread's chunk index cycles with period cpb / gcd(dr, cpb): every
  // item's own registers when the launch needs few items (the slab-fed forms, IT <= 5), the first chunk's otherwise re-loaded per item as before
  constexpr bool kHoist = GS_IT <= 5;
  floatx4 hg0[kHoist ? GS_IT : 1], hg1[kHoist ? GS_IT : 1], hb0[kHoist ? GS_IT : 1], hb1[kHoist ? GS_IT : 1];
  if constexpr (kHoist) {
    int ch = ch0;
#pragma unroll
    for (int u = 0; u < GS_IT; ++u) {
      const int c = c0 + ch * 8;
      hg0[u] = *reinterpret_cast<const floatx4*>(a.gamma + c);
      hg1[u] = *reinterpret_cast<const floatx4*>(a.gamma + c + 4);
      hb0[u] = *reinterpret_cast<const floatx4*>(a.beta + c);
      hb1[u] = *reinterpret_cast<const floatx4*>(a.beta + c + 4);
      ch += dr;
      if (ch >= cpb) ch -= cpb;
    }
  }
  s = af_wave_sum(s);
  q = af_wave_sum(q);
  if (lane == 0) {
    red[0][w] = s;
    red[1][w] = q;
  }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const float inv_n = 1.0f / ((float)a.HW * (float)a.cpg);
  const float dm = s * inv_n;
  const float mean = pivot + dm;
  const float rstd = rsqrtf(fmaxf(q * inv_n - dm * dm, 0.f) + a.eps);
  if (a.stats && t == 0) {
    a.stats[((size_t)b * a.groups + g) * 2 + 0] = mean;
    a.stats[((size_t)b * a.groups + g) * 2 + 1] = rstd;
  }
  int pix = pix0, ch = ch0;
#pragma unroll
  for (int u = 0; u < GS_IT; ++u) {
    if (t + 256 * u < items) {
      const int c = c0 + ch * 8;
      floatx4 g0, g1, b0, b1;
      if constexpr (kHoist) {
        g0 = hg0[u], g1 = hg1[u], b0 = hb0[u], b1 = hb1[u];
      } else {
        g0 = *reinterpret_cast<const floatx4*>(a.gamma + c), g1 = *reinterpret_cast<const floatx4*>(a.gamma + c + 4);
        b0 = *reinterpret_cast<const floatx4*>(a.beta + c), b1 = *reinterpret_cast<const floatx4*>(a.beta + c + 4);
      }
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float k = rstd * (e < 4 ? g0[e & 3] : g1[e & 3]);
        float f = (float)v[u][e] * k + ((e < 4 ? b0[e & 3] : b1[e & 3]) - mean * k);
        if (a.silu) f = af_silu(f);
        o[e] = (half_t)f;
      }
      *reinterpret_cast<half8_t*>(a.y + ((size_t)b * a.HW + pix) * a.C + c) = o;
    }
    pix += dq;
    ch += dr;
    if (ch >= cpb) ch -= cpb, ++pix;
  }
}

// Same idea for ANY group width (cpg even: 10 / 20 / 30 / 60 channels at C = 320 / 640 / 960 / 1920), at 4-byte granularity:
// 1024 threads per (batch item, group), each holding up to 12 channel pairs.  A group's slice of a pixel is 20 .. 120 bytes of a
// 128-byte line that the neighbouring groups' workgroups read too, so all 32 groups of a batch item are mapped to ONE XCD
// (bid % 8 picks the batch item's XCD) and share those lines in its L2 instead of each XCD fetching them again.

template <int IT, bool SLAB>
__global__ __launch_bounds__(1024) void gn_pair_kernel(GnArgs a) {
  __shared__ float red[2][16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int bid = blockIdx.x;
  const int k = bid >> 3;
  const int b = (bid & 7) + 8 * (k / a.groups), g = k % a.groups;
  if (b >= a.B) return;
  const int ppp = a.cpg >> 1;                           // channel pairs per pixel in this group
  const int c0 = g * a.cpg;
  const int items = a.HW * ppp;
  unsigned int v[IT];                                   // fully unrolled below: stays in registers
  float s = 0.f, q = 0.f;
  // (pixel, pair) of item t + 1024 u by steps of divmod(1024, ppp): one integer division per thread instead of one per item and phase
  const int pix0 = t / ppp, pr0 = t - pix0 * ppp;
  const int dq = 1024 / ppp, dr = 1024 - dq * ppp;
  float pivot;
  if constexpr (!SLAB) {
    int pix = pix0, pr = pr0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      unsigned int x = 0;
      if (t + 1024 * u < items) {
        const int c = c0 + 2 * pr;
        const size_t row = (size_t)b * a.HW + pix;
        x = c < a.c1 ? *reinterpret_cast<const unsigned int*>(a.x1 + row * a.c1 + c)
                     : *reinterpret_cast<const unsigned int*>(a.x2 + row * a.c2 + (c - a.c1));
      }
      v[u] = x;
      pix += dq;
      pr += dr;
      if (pr >= ppp) pr -= ppp, ++pix;
    }
    // shifted sums, as gn_small_kernel: the pivot is the group's first element of this batch item
    pivot = c0 < a.c1 ? (float)a.x1[(size_t)b * a.HW * a.c1 + c0] : (float)a.x2[(size_t)b * a.HW * a.c2 + (c0 - a.c1)];
  } else {
    // the producer's split-K slabs at 8-byte granularity (gn_small_kernel<true> has the scheme): all items of a slab in flight together,
    // slice order, then bias / row bias / residual, the fp16 pair stored and kept
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    size_t off[IT];
    floatx2 acc[IT];
    {
      int pix = pix0, pr = pr0;
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        off[u] = t + 1024 * u < items ? ((size_t)b * a.HW + pix) * a.C + c0 + 2 * pr : (size_t)b * a.HW * a.C + c0;   // out of range: the group's first pair (loaded, dropped)
        acc[u] = floatx2{0.f, 0.f};
        pix += dq;
        pr += dr;
        if (pr >= ppp) pr -= ppp, ++pix;
      }
    }
    int sp = 0;
    constexpr int UNR = IT > 8 ? 1 : 2;                 // slabs per memory round trip (1024 threads: 128 registers each), added in slice order
    for (; UNR > 1 && sp + UNR <= a.splits; sp += UNR) {
      const float* sl = a.slab + (size_t)sp * a.slab_stride;
      floatx2 l[UNR][IT];
#pragma unroll
      for (int k = 0; k < UNR; ++k)
#pragma unroll
        for (int u = 0; u < IT; ++u) l[k][u] = *reinterpret_cast<const floatx2*>(sl + (size_t)k * a.slab_stride + off[u]);
#pragma unroll
      for (int k = 0; k < UNR; ++k)
#pragma unroll
        for (int u = 0; u < IT; ++u) acc[u] += l[k][u];
    }
    for (; sp < a.splits; ++sp) {
      const float* sl = a.slab + (size_t)sp * a.slab_stride;
      floatx2 l[IT];
#pragma unroll
      for (int u = 0; u < IT; ++u) l[u] = *reinterpret_cast<const floatx2*>(sl + off[u]);
#pragma unroll
      for (int u = 0; u < IT; ++u) acc[u] += l[u];
    }
    int pr = pr0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      unsigned int x = 0;
      if (t + 1024 * u < items) {
        const int c = c0 + 2 * pr;
        if (a.bias) acc[u] += *reinterpret_cast<const floatx2*>(a.bias + c);
        if (a.rowbias) {
          const half2_t rv = *reinterpret_cast<const half2_t*>(a.rowbias + (size_t)b * a.ld_rowbias + c);
          acc[u][0] += (float)rv[0], acc[u][1] += (float)rv[1];
        }
        if (a.residual) {
          const half2_t rv = *reinterpret_cast<const half2_t*>(a.residual + off[u]);
          acc[u][0] += (float)rv[0], acc[u][1] += (float)rv[1];
        }
        const half2_t h = {(half_t)acc[u][0], (half_t)acc[u][1]};
        x = *reinterpret_cast<const unsigned int*>(&h);
        *reinterpret_cast<unsigned int*>(a.xout + off[u]) = x;
      }
      v[u] = x;
      pr += dr;
      if (pr >= ppp) pr -= ppp;
    }
    if (t == 0) {
      const half2_t h = *reinterpret_cast<const half2_t*>(&v[0]);
      red[0][0] = (float)h[0];
    }
    __syncthreads();
    pivot = red[0][0];
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    if (t + 1024 * u < items) {
      const half2_t h = *reinterpret_cast<const half2_t*>(&v[u]);
      const half_t nhp = (half_t)pivot * (half_t)-0.5f;
      const half2_t np2 = {nhp, nhp}, one2 = {(half_t)1.0f, (half_t)1.0f};
      const half2_t d2 = gn_half_diff(h, np2);
      s = __builtin_amdgcn_fdot2(d2, one2, s, false);
      q = __builtin_amdgcn_fdot2(d2, d2, q, false);
    }
  }
  s = af_wave_sum(2.f * s);                             // the sums were taken over (x - p) / 2
  q = af_wave_sum(4.f * q);
  if (lane == 0) {
    red[0][w] = s;
    red[1][w] = q;
  }
  __syncthreads();
  s = 0.f;
  q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s += red[0][i];
    q += red[1][i];
  }
  const float inv_n = 1.0f / ((float)a.HW * (float)a.cpg);
  const float dm = s * inv_n;
  const float mean = pivot + dm;
  const float rstd = rsqrtf(fmaxf(q * inv_n - dm * dm, 0.f) + a.eps);
  if (a.stats && t == 0) {
    a.stats[((size_t)b * a.groups + g) * 2 + 0] = mean;
    a.stats[((size_t)b * a.groups + g) * 2 + 1] = rstd;
  }
  int pix = pix0, pr = pr0;
#pragma unroll
  for (int u = 0; u < IT; ++u, pix += dq, pr += dr) {
    if (pr >= ppp) pr -= ppp, ++pix;
    if (t + 1024 * u < items) {
      const int c = c0 + 2 * pr;
      const half2_t h = *reinterpret_cast<const half2_t*>(&v[u]);
      const float2 gm = *reinterpret_cast<const float2*>(a.gamma + c), bt = *reinterpret_cast<const float2*>(a.beta + c);   // c is even
      const float k0 = rstd * gm.x, k1 = rstd * gm.y;
      float f0 = (float)h[0] * k0 + (bt.x - mean * k0);
      float f1 = (float)h[1] * k1 + (bt.y - mean * k1);
      if (a.silu) {
        f0 = af_silu(f0);
        f1 = af_silu(f1);
      }
      const half2_t o = {(half_t)f0, (half_t)f1};
      *reinterpret_cast<half2_t*>(a.y + ((size_t)b * a.HW + pix) * a.C + c) = o;
    }
  }
}

// LayerNorm: one wave per row, row held in registers (two-pass mean / variance).
template <int CT>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, half_t* __restrict__ y, int rows,
                                                        int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int CP = C >> 3;
  const half_t* xr = x + (size_t)row * C;
  half8_t v[CT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    if (ch < CP) {
      v[j] = *reinterpret_cast<const half8_t*>(xr + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[j][e];
    }
  }
  const float mean = af_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    if (ch < CP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[j][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(af_wave_sum(q) / (float)C + eps);
  half_t* yr = y + (size_t)row * C;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    if (ch < CP) {
      const floatx4 g0 = *reinterpret_cast<const floatx4*>(gamma + ch * 8);
      const floatx4 g1 = *reinterpret_cast<const floatx4*>(gamma + ch * 8 + 4);
      const floatx4 b0 = *reinterpret_cast<const floatx4*>(beta + ch * 8);
      const floatx4 b1 = *reinterpret_cast<const floatx4*>(beta + ch * 8 + 4);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (half_t)(((float)v[j][e] - mean) * rstd * g0[e] + b0[e]);
        o[e + 4] = (half_t)(((float)v[j][e + 4] - mean) * rstd * g1[e] + b1[e]);
      }
      *reinterpret_cast<half8_t*>(yr + ch * 8) = o;
    }
  }
}

}  // namespace

extern "C" int af_groupnorm_ws_floats(int B) { return B > 0 ? B * GN_NBLK * GN_MAXG * 2 : 0; }

extern "C" int af_groupnorm(const void* x1, const void* x2, int c1, int c2, const void* gamma, const void* beta, void* y,
                            int B, int HW, int groups, float eps, int silu, void* workspace, void* stream) {
  return af_groupnorm_stats(x1, x2, c1, c2, gamma, beta, y, nullptr, B, HW, groups, eps, silu, workspace, stream);
}

extern "C" int af_groupnorm_stats(const void* x1, const void* x2, int c1, int c2, const void* gamma, const void* beta,
                                  void* y, void* stats, int B, int HW, int groups, float eps, int silu, void* workspace,
                                  void* stream) {
  AF_REQUIRE(x1 && gamma && beta && y && workspace, "af_groupnorm: null pointer");
  AF_REQUIRE(B > 0 && HW > 0 && c1 > 0 && c2 >= 0, "af_groupnorm: bad sizes");
  AF_REQUIRE(c1 % 8 == 0 && c2 % 8 == 0, "af_groupnorm: c1/c2 must be multiples of 8");
  AF_REQUIRE(c2 == 0 || x2 != nullptr, "af_groupnorm: x2 is null but c2 > 0");
  // the kernels read x / y in 16-byte chunks and gamma / beta as float4: state-dict slices at 4-byte offsets would be misaligned vector loads
  AF_REQUIRE((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
             "af_groupnorm: x1 / x2 / y / gamma / beta must be 16-byte aligned");
  const int C = c1 + c2;
  AF_REQUIRE(groups > 0 && groups <= GN_MAXG && C % groups == 0, "af_groupnorm: groups must divide C and be <= 32");
  AF_SUPPORTED(C <= 4096, "af_groupnorm: C > 4096");
  GnArgs a;
  a.x1 = (const half_t*)x1;
  a.x2 = (const half_t*)x2;
  a.c1 = c1;
  a.c2 = c2;
  a.C = C;
  a.CP = C / 8;
  a.gamma = (const float*)gamma;
  a.beta = (const float*)beta;
  a.y = (half_t*)y;
  a.B = B;
  a.HW = HW;
  a.groups = groups;
  a.cpg = C / groups;
  a.eps = eps;
  a.silu = silu;
  a.ws = (float*)workspace;
  a.stats = (float*)stats;
  const int ct = (a.CP + 255) / 256;
  a.ppb = ct == 1 ? 256 / a.CP : 1;
  const int slots = ct == 1 ? a.ppb : 1;
  const size_t lds = (size_t)2 * slots * C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_GNORM, stream);
  // small tensors: one launch, one workgroup per (batch item, group)
  static const bool no_small = getenv("AF_GN_NO_SMALL") != nullptr;      // A/B switch
  if (!no_small && a.cpg % 8 == 0 && (c2 == 0 || c1 % a.cpg == 0) && (long)HW * (a.cpg / 8) <= 256L * GS_IT) {
    const long nu = ((long)HW * (a.cpg / 8) + 255) / 256;     // items per thread: the instantiation with that many registers' worth of gamma / beta hoisted
    if (nu <= 2) hipLaunchKernelGGL((gn_small_kernel<false, 2>), dim3(groups, B), dim3(256), 0, s, a);
    else if (nu <= 5) hipLaunchKernelGGL((gn_small_kernel<false, 5>), dim3(groups, B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gn_small_kernel<false>), dim3(groups, B), dim3(256), 0, s, a);
    return af_check_launch("af_groupnorm(small)");
  }
  static const bool no_pair = getenv("AF_GN_NO_PAIR") != nullptr;        // A/B switch
  // (4-byte accesses only pay while the group is small: measured 15.0 vs 19.2 us at [8, 1024, 640] but 32.8 vs 26.3 at [8, 4096, 320])
  if (!no_small && !no_pair && a.cpg % 2 == 0 && c1 % 2 == 0 && (long)HW * (a.cpg / 2) <= 1024L * 12) {
    const int b8 = (B + 7) / 8 * 8;
    const long nu = ((long)HW * (a.cpg / 2) + 1023) / 1024;
    dim3 gp(b8 * groups), bp(1024);
    if (nu <= 4) hipLaunchKernelGGL((gn_pair_kernel<4, false>), gp, bp, 0, s, a);
    else if (nu <= 8) hipLaunchKernelGGL((gn_pair_kernel<8, false>), gp, bp, 0, s, a);
    else hipLaunchKernelGGL((gn_pair_kernel<12, false>), gp, bp, 0, s, a);
    return af_check_launch("af_groupnorm(pair)");
  }
  // enough workgroups to keep >= 4 per CU streaming (B * nblk >= 1024 when the tensor is large enough), each thread with
  // several 16-byte loads in flight; small levels keep >= 4 pixel iterations per thread
  const int iters = (HW + slots - 1) / slots;          // pixel iterations if one block took a whole batch item
  int nblk = (iters + 7) / 8;
  nblk = nblk < 1 ? 1 : (nblk > GN_NBLK ? GN_NBLK : nblk);
  a.nblk = nblk;
  dim3 g1(nblk, B), blk(256);
  int nb2 = (iters + 7) / 8;
  nb2 = nb2 < 1 ? 1 : (nb2 > 256 ? 256 : nb2);
  dim3 g2(nb2, B);
  const bool pair = a.cpg % 2 == 0;                    // every SD-1.5 width; odd group widths keep the per-channel form
  if (ct == 1) {
    if (pair) hipLaunchKernelGGL((gn_partial_kernel<1, true>), g1, blk, lds, s, a);
    else hipLaunchKernelGGL((gn_partial_kernel<1, false>), g1, blk, lds, s, a);
    hipLaunchKernelGGL(gn_apply_kernel<1>, g2, blk, 0, s, a);
  } else if (ct == 2) {
    if (pair) hipLaunchKernelGGL((gn_partial_kernel<2, true>), g1, blk, lds, s, a);
    else hipLaunchKernelGGL((gn_partial_kernel<2, false>), g1, blk, lds, s, a);
    hipLaunchKernelGGL(gn_apply_kernel<2>, g2, blk, 0, s, a);
  } else {
    return af_fail(AF_E_UNSUPPORTED, "af_groupnorm: C > 4096");
  }
  return af_check_launch("af_groupnorm");
}

// scope of the slab-fed one-launch forms: 1 = gn_small_kernel (16-byte chunks), 2 = gn_pair_kernel (4-byte pairs), 0 = neither
static int gn_splitk_form(int HW, int C, int groups) {
  if (groups <= 0 || groups > GN_MAXG || C % groups != 0 || C % 8 != 0 || C > 4096 || HW <= 0) return 0;
  const int cpg = C / groups;
  if (cpg % 8 == 0 && (long)HW * (cpg / 8) <= 256L * GS_IT) return 1;
  if (cpg % 2 == 0 && (long)HW * (cpg / 2) <= 1024L * 12) return 2;
  return 0;
}

extern "C" int af_groupnorm_splitk_ok(int B, int HW, int C, int groups) { return B > 0 && gn_splitk_form(HW, C, groups) != 0 ? 1 : 0; }

extern "C" int af_groupnorm_splitk(const void* slabs, int splits, const void* bias, const void* rowbias, int ld_rowbias, const void* residual, void* x_out,
                                   int C, const void* gamma, const void* beta, void* y, void* stats, int B, int HW, int groups, float eps, int silu,
                                   void* stream) {
  AF_REQUIRE(slabs && x_out && gamma && beta && y, "af_groupnorm_splitk: null pointer");
  AF_REQUIRE(splits >= 1 && B > 0 && HW > 0 && C > 0, "af_groupnorm_splitk: bad sizes");
  AF_REQUIRE((((uintptr_t)slabs | (uintptr_t)bias | (uintptr_t)rowbias | (uintptr_t)residual | (uintptr_t)x_out | (uintptr_t)y | (uintptr_t)gamma |
               (uintptr_t)beta) & 15) == 0 && (rowbias == nullptr || (ld_rowbias >= C && ld_rowbias % 8 == 0)),
             "af_groupnorm_splitk: every pointer must be 16-byte aligned and ld_rowbias a multiple of 8 that covers C");
  const int form = gn_splitk_form(HW, C, groups);
  AF_SUPPORTED(form != 0, "af_groupnorm_splitk: outside the one-launch forms' scope (af_groupnorm_splitk_ok)");
  GnArgs a{};
  a.x1 = (const half_t*)x_out;                       // never read: the SLAB kernels take their values from the slabs
  a.c1 = C;
  a.C = C;
  a.CP = C / 8;
  a.gamma = (const float*)gamma;
  a.beta = (const float*)beta;
  a.y = (half_t*)y;
  a.B = B;
  a.HW = HW;
  a.groups = groups;
  a.cpg = C / groups;
  a.eps = eps;
  a.silu = silu;
  a.stats = (float*)stats;
  a.slab = (const float*)slabs;
  a.slab_stride = (size_t)B * HW * C;
  a.splits = splits;
  a.bias = (const float*)bias;
  a.rowbias = (const half_t*)rowbias;
  a.ld_rowbias = ld_rowbias;
  a.residual = (const half_t*)residual;
  a.xout = (half_t*)x_out;
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_GNORM, stream);
  if (form == 1) {
    const long nu = ((long)HW * (a.cpg / 8) + 255) / 256;       // items per thread
    const dim3 g(groups, B), blk(256);
    if (nu <= 2) hipLaunchKernelGGL((gn_small_kernel<true, 2, 8>), g, blk, 0, s, a);
    else if (nu <= 5) hipLaunchKernelGGL((gn_small_kernel<true, 5, 4>), g, blk, 0, s, a);
    else hipLaunchKernelGGL((gn_small_kernel<true, GS_IT, 2>), g, blk, 0, s, a);
    return af_check_launch("af_groupnorm_splitk(small)");
  }
  const int b8 = (B + 7) / 8 * 8;
  const long nu = ((long)HW * (a.cpg / 2) + 1023) / 1024;
  dim3 gp(b8 * groups), bp(1024);
  if (nu <= 4) hipLaunchKernelGGL((gn_pair_kernel<4, true>), gp, bp, 0, s, a);
  else if (nu <= 8) hipLaunchKernelGGL((gn_pair_kernel<8, true>), gp, bp, 0, s, a);
  else hipLaunchKernelGGL((gn_pair_kernel<12, true>), gp, bp, 0, s, a);
  return af_check_launch("af_groupnorm_splitk(pair)");
}

extern "C" int af_groupnorm_apply(const void* x, int C, const void* gamma, const void* beta, void* y, void* stats, int B, int HW, int groups,
                                  float eps, int silu, const void* partials, int nblk, void* stream) {
  AF_REQUIRE(x && gamma && beta && y && partials, "af_groupnorm_apply: null pointer");
  AF_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0, "af_groupnorm_apply: bad sizes");
  AF_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "af_groupnorm_apply: x / y / gamma / beta must be 16-byte aligned");
  AF_REQUIRE(groups > 0 && groups <= GN_MAXG && C % groups == 0, "af_groupnorm_apply: groups must divide C and be <= 32");
  AF_REQUIRE(nblk > 0 && nblk <= GN_NBLK, "af_groupnorm_apply: nblk must be 1 .. 128");
  AF_SUPPORTED(C <= 4096, "af_groupnorm_apply: C > 4096");
  GnArgs a;
  a.x1 = (const half_t*)x;
  a.x2 = nullptr;
  a.c1 = C;
  a.c2 = 0;
  a.C = C;
  a.CP = C / 8;
  a.gamma = (const float*)gamma;
  a.beta = (const float*)beta;
  a.y = (half_t*)y;
  a.B = B;
  a.HW = HW;
  a.groups = groups;
  a.cpg = C / groups;
  a.eps = eps;
  a.silu = silu;
  a.ws = const_cast<float*>((const float*)partials);
  a.stats = (float*)stats;
  a.nblk = nblk;
  const int ct = (a.CP + 255) / 256;
  a.ppb = ct == 1 ? 256 / a.CP : 1;
  const int slots = ct == 1 ? a.ppb : 1;
  const int iters = (HW + slots - 1) / slots;
  int nb2 = (iters + 7) / 8;
  nb2 = nb2 < 1 ? 1 : (nb2 > 256 ? 256 : nb2);
  AfLaunchScope scope(AF_FAM_GNORM, stream);
  hipStream_t s = (hipStream_t)stream;
  if (ct == 1) hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(nb2, B), dim3(256), 0, s, a);
  else if (ct == 2) hipLaunchKernelGGL(gn_apply_kernel<2>, dim3(nb2, B), dim3(256), 0, s, a);
  else return af_fail(AF_E_UNSUPPORTED, "af_groupnorm_apply: C > 4096");
  return af_check_launch("af_groupnorm_apply");
}

extern "C" int af_layernorm(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                            void* stream) {
  AF_REQUIRE(x && gamma && beta && y, "af_layernorm: null pointer");
  AF_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "af_layernorm: C must be a positive multiple of 8");
  AF_SUPPORTED(C <= 2048, "af_layernorm: C > 2048");
  const int ct = (C / 8 + 63) / 64;
  dim3 grid((rows + 3) / 4), blk(256);
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_LNORM, stream);
  const half_t* xx = (const half_t*)x;
  const float* g = (const float*)gamma;
  const float* b = (const float*)beta;
  half_t* yy = (half_t*)y;
  switch (ct) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, blk, 0, s, xx, g, b, yy, rows, C, eps); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, blk, 0, s, xx, g, b, yy, rows, C, eps); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, blk, 0, s, xx, g, b, yy, rows, C, eps); break;
    default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, blk, 0, s, xx, g, b, yy, rows, C, eps); break;
  }
  return af_check_launch("af_layernorm");
}
