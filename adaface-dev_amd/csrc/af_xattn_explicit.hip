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
// coalesced row accesses; no MFMA.  Reductions over the 4096 queries (dk, dv) are two-pass and deterministic (no atomics).
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
// grid (ceil(L*d / 256), heads, B * NZ); thread = one (j, c) output of its head.
__global__ __launch_bounds__(256) void xattn_colmix_partial_kernel(const float* __restrict__ w, const half_t* __restrict__ x, int ldx,
                                                                   float* __restrict__ partial, int B, int Nq, int L, int heads, int d, int NZ) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y;
  const int b = blockIdx.z / NZ, z = blockIdx.z - b * NZ;
  if (idx >= L * d) return;
  const int j = idx / d, c = idx - j * d;
  const int per = (Nq + NZ - 1) / NZ;
  const int i0 = z * per, i1 = min(Nq, i0 + per);
  const float* wp = w + (((size_t)b * heads + h) * Nq) * L + j;
  const half_t* xp = x + (size_t)b * Nq * ldx + h * d + c;
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) acc += wp[(size_t)i * L] * (float)xp[(size_t)i * ldx];
  partial[(((size_t)z * B + b) * L + j) * (heads * d) + h * d + c] = acc;
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

bool bad_common(int B, int Nq, int L, int heads, int d) { return !(B > 0 && Nq > 0 && L > 0 && L <= 128 && heads > 0 && d > 0 && d % 8 == 0); }

}  // namespace

extern "C" int af_xattn_scores(const void* q, int ldq, const void* k, int ldk, void* score, int B, int Nq, int L, int heads, int d, float scale,
                               void* stream) {
  AF_REQUIRE(q && k && score, "af_xattn_scores: null pointer");
  AF_REQUIRE(!bad_common(B, Nq, L, heads, d), "af_xattn_scores: bad sizes (L <= 128, d % 8 == 0)");
  AF_REQUIRE(ldq >= heads * d && ldk >= heads * d && ldq % 8 == 0 && ldk % 8 == 0, "af_xattn_scores: bad leading dimensions");
  const long rows = (long)B * heads * Nq;
  AfLaunchScope scope(AF_FAM_XATTN, stream);
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
  hipLaunchKernelGGL(xattn_colmix_partial_kernel, dim3((unsigned)((L * d + 255) / 256), heads, B * NZ), dim3(256), 0, (hipStream_t)stream,
                     (const float*)w, (const half_t*)x, ldx, (float*)workspace, B, Nq, L, heads, d, NZ);
  const long n = (long)B * L * heads * d;
  hipLaunchKernelGGL(xattn_colmix_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     (half_t*)out, ldout, alpha, B * L, heads * d, NZ);
  return af_check_launch("af_xattn_colmix");
}
