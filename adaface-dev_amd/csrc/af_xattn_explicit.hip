// af_xattn_explicit.hip -- the EXPLICIT cross-attention of the capture / score-rewrite path, forward and backward.
//
// The three captured cross-attention layers of the live path (diffusers_attn_lora_capture.py:309-315: layers 22-24, 64x64 latents,
// C = 320, 8 heads of 40, 77 / 97 keys) do not run flash attention: the scores are materialised [B, heads, N, L] in fp32, optionally
// rewritten between the product and the softmax (:108-133: SC/MC mixing, subject-token normalisation with a learnable scale), and
// scores, probabilities, q, k, v and the attention output are captured -- in Stage 2 WITH gradients, losses are taken on them.
// So the path is split where the reference splits it:
//
//     af_xattn_scores      score = scale * q k^T                                  (backward: af_xattn_rowmix for dq, af_xattn_colmix for dk)
//     <rewrite of the scores: index bookkeeping on [B, heads, N, L], host side>
//     af_xattn_softmax_pv  prob = softmax(score), o = prob v                      (backward: af_xattn_softmax_pv_bwd -> dscore,
//                                                                                   af_xattn_colmix for dv)
//
// Work per layer-sample is 0.25 GFLOP against 12.7 MB of fp32 scores: HBM / latency bound by construction, so these are plain
// wave-per-row kernels (one wave owns one (batch, head, query) row; lane = key, two keys per lane, L <= 128), fp32 arithmetic,
// coalesced row accesses; no MFMA -- except the two reductions over the 4096 queries (dk, dv), which are [L x Nq] . [Nq x d] GEMMs per head on the
// f32-input matrix instruction (xattn_colmix_mfma_kernel), two-pass and deterministic (no atomics).
#include "af_common.h"

namespace {

constexpr int ROWS_PER_WG = 4;      // waves per workgroup

struct RowId {
  int b, h, i;
  bool ok;
};
__device__ __forceinline__ RowId row_of(int Nq, int heads, long nrows) {
  const long row = (long)blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
  RowId r;
  r.ok = row < nrows;
  const long rr = r.ok ? row : 0;
  r.i = (int)(rr % Nq);
  r.h = (int)((rr / Nq) % heads);
  r.b = (int)(rr / ((long)Nq * heads));
  return r;
}

// score[b,h,i,j] = scale * sum_c q[b,i,h*d+c] k[b,j,h*d+c]
__global__ __launch_bounds__(256) void xattn_scores_kernel(const half_t* __restrict__ q, int ldq, const half_t* __restrict__ k, int ldk,
                                                           float* __restrict__ score, int B, int Nq, int L, int heads, int d, float scale) {
  const int lane = threadIdx.x & 63;
  const long nrows = (long)B * heads * Nq;
  const RowId r = row_of(Nq, heads, nrows);
  if (!r.ok) return;
  const half_t* qp = q + ((size_t)r.b * Nq + r.i) * ldq + r.h * d;
  float* sp = score + (((size_t)r.b * heads + r.h) * Nq + r.i) * L;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = lane + 64 * jj;
    if (j >= L) continue;
    const half_t* kp = k + ((size_t)r.b * L + j) * ldk + r.h * d;
    float acc = 0.f;
    for (int c0 = 0; c0 < d; c0 += 8) {
      const half8_t qv = *reinterpret_cast<const half8_t*>(qp + c0);
      const half8_t kv = *reinterpret_cast<const half8_t*>(kp + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)qv[e] * (float)kv[e];
    }
    sp[j] = acc * scale;
  }
}

// out[c] = alpha * sum_j w[j] x[b,j,h*d+c] for the wave's row; w[j] lives in lanes (j = lane, lane + 64).  Lanes then act as channels.
__device__ __forceinline__ void row_mix(float w0, float w1, const half_t* __restrict__ xb, int ldx, int L, int d, float alpha, float* wsh,
                                        int lane, half_t* __restrict__ outp) {
  wsh[lane] = w0;
  wsh[lane + 64] = w1;
  __builtin_amdgcn_wave_barrier();
  for (int c0 = 0; c0 < d; c0 += 64) {
    const int c = c0 + lane;
    if (c < d) {
      float acc = 0.f;
      for (int j = 0; j < L; ++j) acc += wsh[j] * (float)xb[(size_t)j * ldx + c];
      outp[c] = (half_t)(acc * alpha);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// prob = softmax_j(score); o[b,i,h*d+c] = sum_j prob_j v[b,j,h*d+c]
__global__ __launch_bounds__(256) void xattn_softmax_pv_kernel(const float* __restrict__ score, const half_t* __restrict__ v, int ldv,
                                                               float* __restrict__ prob, half_t* __restrict__ o, int ldo, int B, int Nq, int L,
                                                               int heads, int d) {
  __shared__ float wsh_all[ROWS_PER_WG][128];
  const int lane = threadIdx.x & 63;
  const long nrows = (long)B * heads * Nq;
  const RowId r = row_of(Nq, heads, nrows);
  if (!r.ok) return;
  const size_t ro = (((size_t)r.b * heads + r.h) * Nq + r.i) * L;
  const float s0 = lane < L ? score[ro + lane] : -INFINITY;
  const float s1 = lane + 64 < L ? score[ro + lane + 64] : -INFINITY;
  const float mx = af_wave_max(fmaxf(s0, s1));
  // a row that is -inf everywhere (every key masked with -inf) would give NaN as torch.softmax does; keep the arithmetic identical
  const float e0 = lane < L ? __expf(s0 - mx) : 0.f, e1 = lane + 64 < L ? __expf(s1 - mx) : 0.f;
  const float inv = 1.0f / af_wave_sum(e0 + e1);
  const float p0 = e0 * inv, p1 = e1 * inv;
  if (lane < L) prob[ro + lane] = p0;
  if (lane + 64 < L) prob[ro + lane + 64] = p1;
  row_mix(p0, p1, v + (size_t)r.b * L * ldv + r.h * d, ldv, L, d, 1.0f, wsh_all[threadIdx.x >> 6], lane,
          o + ((size_t)r.b * Nq + r.i) * ldo + r.h * d);
}

// dscore_j = prob_j (dP_j - sum_l prob_l dP_l),   dP_j = sum_c do[b,i,h*d+c] v[b,j,h*d+c] (+ dprob_ext[b,h,i,j])
__global__ __launch_bounds__(256) void xattn_softmax_pv_bwd_kernel(const float* __restrict__ prob, const half_t* __restrict__ v, int ldv,
                                                                   const half_t* __restrict__ dout, int lddo, const float* __restrict__ dprob_ext,
                                                                   float* __restrict__ dscore, int B, int Nq, int L, int heads, int d) {
  const int lane = threadIdx.x & 63;
  const long nrows = (long)B * heads * Nq;
  const RowId r = row_of(Nq, heads, nrows);
  if (!r.ok) return;
  const size_t ro = (((size_t)r.b * heads + r.h) * Nq + r.i) * L;
  const half_t* dop = dout + ((size_t)r.b * Nq + r.i) * lddo + r.h * d;
  float dp[2], p[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = lane + 64 * jj;
    dp[jj] = 0.f;
    p[jj] = 0.f;
    if (j >= L) continue;
    const half_t* vp = v + ((size_t)r.b * L + j) * ldv + r.h * d;
    float acc = 0.f;
    for (int c0 = 0; c0 < d; c0 += 8) {
      const half8_t gv = *reinterpret_cast<const half8_t*>(dop + c0);
      const half8_t vv = *reinterpret_cast<const half8_t*>(vp + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)gv[e] * (float)vv[e];
    }
    if (dprob_ext) acc += dprob_ext[ro + j];
    dp[jj] = acc;
    p[jj] = prob[ro + j];
  }
  const float t = af_wave_sum(p[0] * dp[0] + p[1] * dp[1]);
  if (lane < L) dscore[ro + lane] = p[0] * (dp[0] - t);
  if (lane + 64 < L) dscore[ro + lane + 64] = p[1] * (dp[1] - t);
}

// out[b,i,h*d+c] = alpha * sum_j w[b,h,i,j] x[b,j,h*d+c]       (dq = scale * dscore k)
__global__ __launch_bounds__(256) void xattn_rowmix_kernel(const float* __restrict__ w, const half_t* __restrict__ x, int ldx, half_t* __restrict__ out,
                                                           int ldout, float alpha, int B, int Nq, int L, int heads, int d) {
  __shared__ float wsh_all[ROWS_PER_WG][128];
  const int lane = threadIdx.x & 63;
  const long nrows = (long)B * heads * Nq;
  const RowId r = row_of(Nq, heads, nrows);
  if (!r.ok) return;
  const size_t ro = (((size_t)r.b * heads + r.h) * Nq + r.i) * L;
  const float w0 = lane < L ? w[ro + lane] : 0.f, w1 = lane + 64 < L ? w[ro + lane + 64] : 0.f;
  row_mix(w0, w1, x + (size_t)r.b * L * ldx + r.h * d, ldx, L, d, alpha, wsh_all[threadIdx.x >> 6], lane,
          out + ((size_t)r.b * Nq + r.i) * ldout + r.h * d);
}

// partial[z][b][j][h*d+c] = sum over the z-th query chunk of w[b,h,i,j] x[b,i,h*d+c]      (dv: w = prob, x = do;  dk: w = dscore, x = q)
//
// Per (batch item, head) this is the GEMM  [L keys x Nq queries] . [Nq queries x d channels]  with the REDUCTION over the queries -- both operands have
// the reduction index as their slow (row) index in memory.  It runs on the f32-input matrix instruction v_mfma_f32_16x16x4_f32 (round 5; the scalar
// loop it replaces took 343 us per call at [4, 8, 4096, 97] x d 40, one thread per (key, channel) walking the queries at stride L):
//   * f32 in, f32 accumulate: bit for bit a k-ordered fmaf chain, so the fp32 probabilities / score gradients are NOT rounded to fp16 (they can be
//     anywhere in fp32's range: a gradient) and the result is deterministic;
//   * a lane holds ONE operand element per instruction -- A[key = lane & 15][query = lane >> 4], B[query = lane >> 4][channel = lane & 15] -- so the
//     fragments are plain coalesced loads of the natural layouts: 4 rows x 16 consecutive floats of w, 4 rows x 16 consecutive halves of x; nothing
//     is transposed, staged in LDS or shared between waves;
//   * a WORKGROUP owns one of 16 query chunks of one (batch item, head, group of NT 16-channel tiles) and all ceil(L / 16) key tiles; its four
//     waves take a quarter of the chunk each: up to 8 x NT accumulators per wave, 8 + NT loads per 4 queries for 8 NT MFMAs, the fragments of the
//     next TWO steps in flight under this step's MFMAs (the loads are HBM-latency bound: with one step in flight the kernel took 50 us, 5x its
//     MFMA time); the four waves' sums meet in LDS in wave order (deterministic) and leave as one fp32 slab.
// Grid (16 * ngroups, heads, B); the 16 chunk slabs are summed by xattn_colmix_final_kernel (two deterministic passes, no atomics, as before).
// At full size ([4, 8, 4096, 97] x 40): 512 workgroups, 21 MFMAs per step, ~0.7 M MFMAs of 32 cycles over 1024 SIMDs.
// The loop is BRANCH-FREE: every load is unconditional on a clamped (always valid) address and the out-of-range lanes are zeroed by a select when the
// step is consumed -- with `cond ? load : 0` hipcc built a branch per load and an s_waitcnt vmcnt(0) behind every one of them (50 us per call).
template <int NT, int MT>
__global__ __launch_bounds__(256) void xattn_colmix_mfma_kernel(const float* __restrict__ w, const half_t* __restrict__ x, int ldx,
                                                                float* __restrict__ partial, int B, int Nq, int L, int heads, int d, int NZ) {
  __shared__ float red[MT * 16][NT * 16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int zc = blockIdx.x % NZ;                              // this workgroup's query chunk
  const int ng = blockIdx.x / NZ;                              // channel group: tiles ng * NT ..
  const int h = blockIdx.y, b = blockIdx.z;
  const int per = (Nq + NZ - 1) / NZ;
  const int sub = ((per + 15) >> 4) << 2;                      // queries per wave: a quarter of the chunk, rounded up to whole steps of 4
  const int c1 = min(Nq, (zc + 1) * per);
  const int i0 = zc * per + wave * sub, i1 = min(c1, i0 + sub);
  const int lr = lane & 15, lk = lane >> 4;
  const int cbase = ng * NT * 16;
  const float* wp = w + (((size_t)b * heads + h) * Nq) * L;
  const half_t* xp = x + (size_t)b * Nq * ldx + h * d;
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zf;
  // per-lane column offsets (clamped) and validity, fixed for the whole kernel
  int jo[MT], co[NT];
  bool jv[MT], cv[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int j = mt * 16 + lr;
    jv[mt] = j < L;
    jo[mt] = min(j, L - 1);
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = cbase + nt * 16 + lr;
    cv[nt] = c < d;
    co[nt] = min(c, d - 1);
  }
  struct Frag {
    float a[MT];
    half_t b[NT];
    bool ok;
  };
  auto load = [&](int i, Frag& f) {
    const int ii = i + lk;
    f.ok = ii < i1;
    const int ic = min(ii, Nq - 1);
    const float* wr = wp + (size_t)ic * L;
    const half_t* xr = xp + (size_t)ic * ldx;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.a[mt] = wr[jo[mt]];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) f.b[nt] = xr[co[nt]];
  };
  auto mma = [&](const Frag& f) {
    float bf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[nt] = (f.ok && cv[nt]) ? (float)f.b[nt] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float av = (f.ok && jv[mt]) ? f.a[mt] : 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[nt], acc[mt][nt], 0, 0, 0);
    }
  };
  // three fragment sets rotate, two steps in flight under the MFMAs of the third; steps past the wave's range read a clamped row and multiply zeros
  // (sched_barrier: without it hipcc sinks each load group down to its consumers -- fewer live registers, and every step a full memory round trip)
  Frag f0, f1, f2;
  load(i0, f0);
  load(i0 + 4, f1);
  for (int i = i0; i < i1; i += 12) {
    load(i + 8, f2);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    load(i + 12, f0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
    __builtin_amdgcn_sched_barrier(0);
    load(i + 16, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f2);
    __builtin_amdgcn_sched_barrier(0);
  }
  // D: column (channel) = lane & 15, row (key) = 4 (lane >> 4) + r.  The four waves add into the LDS tile in wave order (deterministic).
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* cell = &red[mt * 16 + 4 * lk + r][nt * 16 + lr];
            *cell = wv == 0 ? acc[mt][nt][r] : *cell + acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  // an empty chunk still writes its zeros: the final pass sums all NZ slabs
  const int Cn = heads * d;
  for (int e = threadIdx.x; e < MT * 16 * NT * 16; e += 256) {
    const int j = e / (NT * 16), cc = e - j * (NT * 16);
    const int c = cbase + cc;
    if (j < L && c < d) partial[(((size_t)zc * B + b) * L + j) * Cn + h * d + c] = red[j][cc];
  }
}

template <int NT>
static void launch_colmix_mfma(dim3 grid, hipStream_t s, const float* w, const half_t* x, int ldx, float* ws, int B, int Nq, int L, int heads, int d, int NZ) {
  if (L <= 80) hipLaunchKernelGGL((xattn_colmix_mfma_kernel<NT, 5>), grid, dim3(256), 0, s, w, x, ldx, ws, B, Nq, L, heads, d, NZ);
  else if (L <= 112) hipLaunchKernelGGL((xattn_colmix_mfma_kernel<NT, 7>), grid, dim3(256), 0, s, w, x, ldx, ws, B, Nq, L, heads, d, NZ);
  else hipLaunchKernelGGL((xattn_colmix_mfma_kernel<NT, 8>), grid, dim3(256), 0, s, w, x, ldx, ws, B, Nq, L, heads, d, NZ);
}

__global__ __launch_bounds__(256) void xattn_colmix_final_kernel(const float* __restrict__ partial, half_t* __restrict__ out, int ldout, float alpha,
                                                                 int rows, int C, int NZ) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)rows * C) return;
  const int row = (int)(idx / C), c = (int)(idx - (long)row * C);
  float acc = 0.f;
  for (int z = 0; z < NZ; ++z) acc += partial[((size_t)z * rows + row) * C + c];
  out[(size_t)row * ldout + c] = (half_t)(acc * alpha);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the four wave-per-row kernels above as MFMA kernels (they are kept as the fallback outside the MFMA forms' scope: L > 128 never
// reaches this file, d % 8 == 0 always holds).  On v_mfma_f32_16x16x4_f32 a lane holds ONE element of each operand (f32 in, f32 accumulate:
// bit for bit an fmaf chain over the reduction index), so every fragment is a plain load of the natural layout -- nothing is staged or transposed
// -- and the fp32 scores / probabilities / score gradients are never rounded.  A WAVE owns 16 queries of one (batch item, head).
//
//   xattn_qk_mfma_kernel<MT, MODE>   D[query][key] = sum_c a[query][c] b[key][c]  (a = q or dO rows, b = k or v rows; reduction over the head dimension)
//       MODE 0: score = scale * D;   MODE 1: dscore = P o (dP - sum_keys P dP), dP = D (+ dprob_ext)      -- lanes = 16 consecutive keys: 64-byte runs
//   xattn_mix_mfma_kernel<DT, S, SM> D[query][channel] = sum_key w[query][key] x[key][channel]             (reduction over the keys)
//       SM = 1: w = softmax(score row), written out as prob;   SM = 0: w as given (dq = scale * dscore k)
template <int MT, int MODE>
__global__ __launch_bounds__(256) void xattn_qk_mfma_kernel(const half_t* __restrict__ a, int lda, const half_t* __restrict__ bm, int ldb,
                                                            float* __restrict__ out, const float* __restrict__ prob, const float* __restrict__ ext,
                                                            int B, int Nq, int L, int heads, int d, float scale) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  const int h = blockIdx.y, b = blockIdx.z;
  if (i0 >= Nq) return;
  const half_t* ap = a + ((size_t)b * Nq + min(i0 + lr, Nq - 1)) * lda + h * d + lk;
  const half_t* bp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bp[mt] = bm + ((size_t)b * L + min(16 * mt + lr, L - 1)) * ldb + h * d + lk;
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = zf;
  for (int c = 0; c < d; c += 4) {                             // unconditional loads of clamped rows; rows / keys out of range are dropped below
    const float av = (float)ap[c];
    float bv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bv[mt] = (float)bp[mt][c];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[mt], acc[mt], 0, 0, 0);
  }
  // D: row (query) = 4 lk + r, column (key) = lr
  const size_t row0 = ((size_t)b * heads + h) * Nq;
  if (MODE == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int j = 16 * mt + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * lk + r;
        if (i < Nq && j < L) out[(row0 + i) * L + j] = acc[mt][r] * scale;
      }
    }
  } else {
    float p[MT][4], t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int j = 16 * mt + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * lk + r;
        const bool ok = i < Nq && j < L;
        const size_t o = (row0 + min(i, Nq - 1)) * L + min(j, L - 1);
        const float pv = prob[o];
        float dp = acc[mt][r];
        if (ext != nullptr) dp += ext[o];
        p[mt][r] = ok ? pv : 0.f;
        acc[mt][r] = dp;
        t[r] += p[mt][r] * dp;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                              // the query's keys sit in the 16 lanes of its lk group
      float v = t[r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      t[r] = v;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int j = 16 * mt + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * lk + r;
        if (i < Nq && j < L) out[(row0 + i) * L + j] = p[mt][r] * (acc[mt][r] - t[r]);
      }
    }
  }
}

template <int DT, int S, int SM>
__global__ __launch_bounds__(256) void xattn_mix_mfma_kernel(const float* __restrict__ w, const half_t* __restrict__ x, int ldx, float* __restrict__ prob,
                                                             half_t* __restrict__ out, int ldo, float alpha, int B, int Nq, int L, int heads, int d) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  const int h = blockIdx.y, b = blockIdx.z;
  if (i0 >= Nq) return;
  // A operand: w[query = i0 + lr][key = 4 s + lk]; the query's S x 4 keys are spread over the four lanes lr, lr + 16, lr + 32, lr + 48
  const bool qok = i0 + lr < Nq;
  const size_t wrow = (((size_t)b * heads + h) * Nq + min(i0 + lr, Nq - 1)) * L;
  float wv[S];
#pragma unroll
  for (int st = 0; st < S; ++st) {
    const int j = 4 * st + lk;
    const float v = w[wrow + min(j, L - 1)];
    wv[st] = j < L ? v : (SM ? -INFINITY : 0.f);
  }
  if (SM) {
    float mx = wv[0];
#pragma unroll
    for (int st = 1; st < S; ++st) mx = fmaxf(mx, wv[st]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int st = 0; st < S; ++st) {
      wv[st] = 4 * st + lk < L ? __expf(wv[st] - mx) : 0.f;    // (a row that is -inf everywhere gives NaN, as torch.softmax does)
      sum += wv[st];
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int st = 0; st < S; ++st) {
      wv[st] *= inv;
      const int j = 4 * st + lk;
      if (qok && j < L) prob[wrow + j] = wv[st];
    }
  }
  const half_t* xp = x + (size_t)b * L * ldx + h * d;
  int co[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) co[dt] = min(16 * dt + lr, d - 1);
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = zf;
#pragma unroll
  for (int st = 0; st < S; ++st) {
    const half_t* xr = xp + (size_t)min(4 * st + lk, L - 1) * ldx;          // B operand: x[key = 4 st + lk][channel = 16 dt + lr]
    float xv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) xv[dt] = (float)xr[co[dt]];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[st], xv[dt], acc[dt], 0, 0, 0);
  }
  // D: row (query) = 4 lk + r, column (channel) = lr
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int c = 16 * dt + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * lk + r;
      if (i < Nq && c < d) out[((size_t)b * Nq + i) * ldo + h * d + c] = (half_t)(acc[dt][r] * alpha);
    }
  }
}

template <int MODE>
static void launch_qk_mfma(hipStream_t s, const half_t* a, int lda, const half_t* bm, int ldb, float* out, const float* prob, const float* ext, int B, int Nq,
                           int L, int heads, int d, float scale) {
  const dim3 grid((Nq + 63) / 64, heads, B), block(256);
  if (L <= 16) hipLaunchKernelGGL((xattn_qk_mfma_kernel<1, MODE>), grid, block, 0, s, a, lda, bm, ldb, out, prob, ext, B, Nq, L, heads, d, scale);
  else if (L <= 80) hipLaunchKernelGGL((xattn_qk_mfma_kernel<5, MODE>), grid, block, 0, s, a, lda, bm, ldb, out, prob, ext, B, Nq, L, heads, d, scale);
  else if (L <= 112) hipLaunchKernelGGL((xattn_qk_mfma_kernel<7, MODE>), grid, block, 0, s, a, lda, bm, ldb, out, prob, ext, B, Nq, L, heads, d, scale);
  else hipLaunchKernelGGL((xattn_qk_mfma_kernel<8, MODE>), grid, block, 0, s, a, lda, bm, ldb, out, prob, ext, B, Nq, L, heads, d, scale);
}

template <int DT, int SM>
static void launch_mix_mfma_s(hipStream_t s, const float* w, const half_t* x, int ldx, float* prob, half_t* out, int ldo, float alpha, int B, int Nq, int L,
                              int heads, int d) {
  const dim3 grid((Nq + 63) / 64, heads, B), block(256);
  if (L <= 16) hipLaunchKernelGGL((xattn_mix_mfma_kernel<DT, 4, SM>), grid, block, 0, s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
  else if (L <= 80) hipLaunchKernelGGL((xattn_mix_mfma_kernel<DT, 20, SM>), grid, block, 0, s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
  else if (L <= 100) hipLaunchKernelGGL((xattn_mix_mfma_kernel<DT, 25, SM>), grid, block, 0, s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
  else hipLaunchKernelGGL((xattn_mix_mfma_kernel<DT, 32, SM>), grid, block, 0, s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
}

template <int SM>
static void launch_mix_mfma(hipStream_t s, const float* w, const half_t* x, int ldx, float* prob, half_t* out, int ldo, float alpha, int B, int Nq, int L,
                            int heads, int d) {
  if (d <= 16) launch_mix_mfma_s<1, SM>(s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
  else if (d <= 48) launch_mix_mfma_s<3, SM>(s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
  else if (d <= 80) launch_mix_mfma_s<5, SM>(s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
  else launch_mix_mfma_s<10, SM>(s, w, x, ldx, prob, out, ldo, alpha, B, Nq, L, heads, d);
}

static bool xattn_use_mfma(int d) {
  static const bool off = getenv("AF_XATTN_EXPLICIT_MFMA") != nullptr && atoi(getenv("AF_XATTN_EXPLICIT_MFMA")) == 0;     // A/B switch: 0 = the wave-per-row kernels
  return !off && d <= 160;
}

bool bad_common(int B, int Nq, int L, int heads, int d) { return !(B > 0 && Nq > 0 && L > 0 && L <= 128 && heads > 0 && d > 0 && d % 8 == 0); }

}  // namespace

extern "C" int af_xattn_scores(const void* q, int ldq, const void* k, int ldk, void* score, int B, int Nq, int L, int heads, int d, float scale,
                               void* stream) {
  AF_REQUIRE(q && k && score, "af_xattn_scores: null pointer");
  AF_REQUIRE(!bad_common(B, Nq, L, heads, d), "af_xattn_scores: bad sizes (L <= 128, d % 8 == 0)");
  AF_REQUIRE(ldq >= heads * d && ldk >= heads * d && ldq % 8 == 0 && ldk % 8 == 0, "af_xattn_scores: bad leading dimensions");
  const long rows = (long)B * heads * Nq;
  AfLaunchScope scope(AF_FAM_XATTN, stream);
  if (xattn_use_mfma(d)) {
    launch_qk_mfma<0>((hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (float*)score, nullptr, nullptr, B, Nq, L, heads, d, scale);
    return af_check_launch("af_xattn_scores");
  }
  hipLaunchKernelGGL(xattn_scores_kernel, dim3((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)q, ldq, (const half_t*)k, ldk, (float*)score, B, Nq, L, heads, d, scale);
  return af_check_launch("af_xattn_scores");
}

extern "C" int af_xattn_softmax_pv(const void* score, const void* v, int ldv, void* prob, void* o, int ldo, int B, int Nq, int L, int heads, int d,
                                   void* stream) {
  AF_REQUIRE(score && v && prob && o, "af_xattn_softmax_pv: null pointer");
  AF_REQUIRE(!bad_common(B, Nq, L, heads, d), "af_xattn_softmax_pv: bad sizes (L <= 128, d % 8 == 0)");
  AF_REQUIRE(ldv >= heads * d && ldo >= heads * d, "af_xattn_softmax_pv: bad leading dimensions");
  const long rows = (long)B * heads * Nq;
  AfLaunchScope scope(AF_FAM_XATTN, stream);
  if (xattn_use_mfma(d)) {
    launch_mix_mfma<1>((hipStream_t)stream, (const float*)score, (const half_t*)v, ldv, (float*)prob, (half_t*)o, ldo, 1.0f, B, Nq, L, heads, d);
    return af_check_launch("af_xattn_softmax_pv");
  }
  hipLaunchKernelGGL(xattn_softmax_pv_kernel, dim3((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)score, (const half_t*)v, ldv, (float*)prob, (half_t*)o, ldo, B, Nq, L, heads, d);
  return af_check_launch("af_xattn_softmax_pv");
}

extern "C" int af_xattn_softmax_pv_bwd(const void* prob, const void* v, int ldv, const void* dout, int lddo, const void* dprob_ext, void* dscore,
                                       int B, int Nq, int L, int heads, int d, void* stream) {
  AF_REQUIRE(prob && v && dout && dscore, "af_xattn_softmax_pv_bwd: null pointer");
  AF_REQUIRE(!bad_common(B, Nq, L, heads, d), "af_xattn_softmax_pv_bwd: bad sizes (L <= 128, d % 8 == 0)");
  AF_REQUIRE(ldv >= heads * d && lddo >= heads * d && ldv % 8 == 0 && lddo % 8 == 0, "af_xattn_softmax_pv_bwd: bad leading dimensions");
  const long rows = (long)B * heads * Nq;
  AfLaunchScope scope(AF_FAM_XATTN, stream);
  if (xattn_use_mfma(d)) {
    launch_qk_mfma<1>((hipStream_t)stream, (const half_t*)dout, lddo, (const half_t*)v, ldv, (float*)dscore, (const float*)prob, (const float*)dprob_ext, B, Nq, L,
                      heads, d, 1.0f);
    return af_check_launch("af_xattn_softmax_pv_bwd");
  }
  hipLaunchKernelGGL(xattn_softmax_pv_bwd_kernel, dim3((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)prob, (const half_t*)v, ldv, (const half_t*)dout, lddo, (const float*)dprob_ext, (float*)dscore, B, Nq, L,
                     heads, d);
  return af_check_launch("af_xattn_softmax_pv_bwd");
}

extern "C" int af_xattn_rowmix(const void* w, const void* x, int ldx, void* out, int ldout, float alpha, int B, int Nq, int L, int heads, int d,
                               void* stream) {
  AF_REQUIRE(w && x && out, "af_xattn_rowmix: null pointer");
  AF_REQUIRE(!bad_common(B, Nq, L, heads, d), "af_xattn_rowmix: bad sizes (L <= 128, d % 8 == 0)");
  AF_REQUIRE(ldx >= heads * d && ldout >= heads * d, "af_xattn_rowmix: bad leading dimensions");
  const long rows = (long)B * heads * Nq;
  AfLaunchScope scope(AF_FAM_XATTN, stream);
  if (xattn_use_mfma(d)) {
    launch_mix_mfma<0>((hipStream_t)stream, (const float*)w, (const half_t*)x, ldx, nullptr, (half_t*)out, ldout, alpha, B, Nq, L, heads, d);
    return af_check_launch("af_xattn_rowmix");
  }
  hipLaunchKernelGGL(xattn_rowmix_kernel, dim3((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)w, (const half_t*)x, ldx, (half_t*)out, ldout, alpha, B, Nq, L, heads, d);
  return af_check_launch("af_xattn_rowmix");
}

extern "C" int64_t af_xattn_colmix_ws_bytes(int B, int L, int heads, int d) { return (int64_t)AF_XATTN_COLMIX_CHUNKS * B * L * heads * d * 4; }

extern "C" int af_xattn_colmix(const void* w, const void* x, int ldx, void* out, int ldout, float alpha, void* workspace, int64_t workspace_bytes,
                               int B, int Nq, int L, int heads, int d, void* stream) {
  AF_REQUIRE(w && x && out && workspace, "af_xattn_colmix: null pointer");
  AF_REQUIRE(!bad_common(B, Nq, L, heads, d), "af_xattn_colmix: bad sizes (L <= 128, d % 8 == 0)");
  AF_REQUIRE(ldx >= heads * d && ldout >= heads * d, "af_xattn_colmix: bad leading dimensions");
  AF_REQUIRE(workspace_bytes >= af_xattn_colmix_ws_bytes(B, L, heads, d), "af_xattn_colmix: workspace too small (af_xattn_colmix_ws_bytes)");
  const int NZ = AF_XATTN_COLMIX_CHUNKS;
    AfLaunchScope scope(AF_FAM_XATTN, stream);
  const int nt = d <= 16 ? 1 : (d <= 32 ? 2 : (d <= 48 ? 3 : 4));
  const dim3 grid(NZ * ((d + nt * 16 - 1) / (nt * 16)), heads, B);
  switch (nt) {
    case 1: launch_colmix_mfma<1>(grid, (hipStream_t)stream, (const float*)w, (const half_t*)x, ldx, (float*)workspace, B, Nq, L, heads, d, NZ); break;
    case 2: launch_colmix_mfma<2>(grid, (hipStream_t)stream, (const float*)w, (const half_t*)x, ldx, (float*)workspace, B, Nq, L, heads, d, NZ); break;
    case 3: launch_colmix_mfma<3>(grid, (hipStream_t)stream, (const float*)w, (const half_t*)x, ldx, (float*)workspace, B, Nq, L, heads, d, NZ); break;
    default: launch_colmix_mfma<4>(grid, (hipStream_t)stream, (const float*)w, (const half_t*)x, ldx, (float*)workspace, B, Nq, L, heads, d, NZ); break;
  }
  const long n = (long)B * L * heads * d;
  hipLaunchKernelGGL(xattn_colmix_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     (half_t*)out, ldout, alpha, B * L, heads * d, NZ);
  return af_check_launch("af_xattn_colmix");
}
