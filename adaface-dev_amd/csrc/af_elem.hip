// af_elem.hip -- small element-wise kernels around the U-Net: timestep embedding, layout
// conversion at the NCHW fp32 API boundary, classifier-free guidance + DDIM update, q_sample.
#include "af_common.h"

namespace {

__global__ void temb_kernel(const int64_t* __restrict__ t, half_t* __restrict__ out, int B, int dim, float max_period) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half_dim = dim / 2;
  if (idx >= B * dim) return;
  const int b = idx / dim, j = idx - b * dim;
  float v = 0.f;
  if (j < 2 * half_dim) {
    const int kk = j < half_dim ? j : j - half_dim;
    // freqs = exp(-ln(max_period) * k / half)   (util.py:165-167), args = t * freqs
    const float f = expf(-logf(max_period) * (float)kk / (float)half_dim);
    const float arg = (float)t[b] * f;
    v = j < half_dim ? cosf(arg) : sinf(arg);
  }
  out[idx] = (half_t)v;
}

// row softmax of an explicit fp16 score matrix (the VAE decoder's single-head 512-dim attention, model.py:179-189, runs as
// GEMM -> softmax -> GEMM: its head dim is beyond the flash kernel's register budget and it is one layer at N = 4096).
// One wave per row, 16-byte accesses; rows up to 64 x 8 x RS elements are held in registers between the passes.
template <int RS>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long rows, int L) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const half_t* xr = x + row * L;
  half8_t v[RS];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < RS; ++j) {
    const int c = (lane + 64 * j) * 8;
    if (c < L) {
      v[j] = *reinterpret_cast<const half8_t*>(xr + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[j][e]);
    }
  }
  mx = af_wave_max(mx);
  float sum = 0.f;
  float p[RS][8];
#pragma unroll
  for (int j = 0; j < RS; ++j) {
    const int c = (lane + 64 * j) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p[j][e] = c < L ? __expf((float)v[j][e] - mx) : 0.f;
      sum += p[j][e];
    }
  }
  const float inv = 1.0f / af_wave_sum(sum);
  half_t* yr = y + row * L;
#pragma unroll
  for (int j = 0; j < RS; ++j) {
    const int c = (lane + 64 * j) * 8;
    if (c < L) {
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)(p[j][e] * inv);
      *reinterpret_cast<half8_t*>(yr + c) = o;
    }
  }
}

// the VAE encoder's masked attention (model.py:191-209): AFTER the softmax, weights between tokens of different classes are zeroed
// (no renormalisation): p[i][j] *= ((cls[i] & cls[j]) != 0), cls bit 0 = foreground (fg*aug != 0), bit 1 = background ((1-fg)*aug != 0)
__global__ __launch_bounds__(256) void mask_pairs_kernel(half_t* __restrict__ p, const unsigned char* __restrict__ cls, int N, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int per_row = N >> 3;
  const int row = (int)(i / per_row), c0 = (int)(i - (long)row * per_row) * 8;
  const unsigned char ci = cls[row];
  half8_t v = *reinterpret_cast<half8_t*>(p + i * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if ((cls[c0 + e] & ci) == 0) v[e] = (half_t)0;
  *reinterpret_cast<half8_t*>(p + i * 8) = v;
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, half_t* __restrict__ y, int B, int C, int HW, int cpad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * HW * cpad;
  if (idx >= n) return;
  const int c = (int)(idx % cpad);
  const long pix = idx / cpad;
  const int b = (int)(pix / HW);
  const int p = (int)(pix - (long)b * HW);
  y[idx] = c < C ? (half_t)x[((long)b * C + c) * HW + p] : (half_t)0;
}

__global__ void nhwc_to_nchw_kernel(const half_t* __restrict__ x, float* __restrict__ y, int B, int C, int HW, int cstride) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * C * HW;
  if (idx >= n) return;
  const int p = (int)(idx % HW);
  const long bc = idx / HW;
  const int c = (int)(bc % C);
  const int b = (int)(bc / C);
  y[idx] = (float)x[((long)b * HW + p) * cstride + c];
}

__global__ void cfg_ddim_kernel(const float* __restrict__ eps2, const float* __restrict__ x, float* __restrict__ x_prev,
                                float* __restrict__ pred_x0, long n, int has_uncond, float g, float sqrt_one_minus_at,
                                float sqrt_at, float sqrt_aprev, float dir_coef) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e = eps2[i];
  if (has_uncond) {
    const float eu = eps2[n + i];
    e = eu + g * (e - eu);
  }
  const float p0 = (x[i] - sqrt_one_minus_at * e) / sqrt_at;
  if (pred_x0) pred_x0[i] = p0;
  x_prev[i] = sqrt_aprev * p0 + dir_coef * e;
}

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const float* __restrict__ sa,
                                const float* __restrict__ sb, float* __restrict__ xt, int B, long per) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * per) return;
  const int b = (int)(i / per);
  xt[i] = sa[b] * x0[i] + sb[b] * noise[i];
}

__global__ void silu_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (half_t)af_silu((float)x[i]);
}

inline dim3 grid1d(long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

extern "C" int af_timestep_embedding(const void* timesteps_i64, void* out, int B, int dim, float max_period, void* stream) {
  AF_REQUIRE(timesteps_i64 && out && B > 0 && dim > 0, "af_timestep_embedding: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(temb_kernel, grid1d((long)B * dim), dim3(256), 0, (hipStream_t)stream, (const int64_t*)timesteps_i64,
                     (half_t*)out, B, dim, max_period);
  return af_check_launch("af_timestep_embedding");
}

extern "C" int af_nchw_f32_to_nhwc_f16(const void* x, void* y, int B, int C, int HW, int cpad, void* stream) {
  AF_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && cpad >= C, "af_nchw_f32_to_nhwc_f16: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid1d((long)B * HW * cpad), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                     (half_t*)y, B, C, HW, cpad);
  return af_check_launch("af_nchw_f32_to_nhwc_f16");
}

extern "C" int af_nhwc_f16_to_nchw_f32(const void* x, void* y, int B, int C, int HW, int cstride, void* stream) {
  AF_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && cstride >= C, "af_nhwc_f16_to_nchw_f32: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid1d((long)B * C * HW), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (float*)y, B, C, HW, cstride);
  return af_check_launch("af_nhwc_f16_to_nchw_f32");
}

extern "C" int af_cfg_ddim_step(const void* eps2, const void* x, void* x_prev, void* pred_x0, int64_t n, int has_uncond,
                                float guidance, float a_t, float a_prev, void* stream) {
  AF_REQUIRE(eps2 && x && x_prev && n > 0, "af_cfg_ddim_step: bad argument");
  AF_REQUIRE(a_t > 0.f && a_t <= 1.f && a_prev > 0.f && a_prev <= 1.f, "af_cfg_ddim_step: alphas must be in (0, 1]");
  // fp32 scalar arithmetic exactly as ddim.py:279-301 (torch.full(..., fp32).sqrt())
  const float sqrt_one_minus_at = sqrtf(1.0f - a_t);
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_coef = sqrtf(1.0f - a_prev);
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(cfg_ddim_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const float*)eps2, (const float*)x,
                     (float*)x_prev, (float*)pred_x0, (long)n, has_uncond, guidance, sqrt_one_minus_at, sqrt_at, sqrt_aprev,
                     dir_coef);
  return af_check_launch("af_cfg_ddim_step");
}

extern "C" int af_q_sample(const void* x0, const void* noise, const void* sa, const void* sb, void* xt, int B, int64_t per,
                           void* stream) {
  AF_REQUIRE(x0 && noise && sa && sb && xt && B > 0 && per > 0, "af_q_sample: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(q_sample_kernel, grid1d((long)B * per), dim3(256), 0, (hipStream_t)stream, (const float*)x0,
                     (const float*)noise, (const float*)sa, (const float*)sb, (float*)xt, B, (long)per);
  return af_check_launch("af_q_sample");
}

// dS = P * (dP - rowsum(P * dP)): the softmax backward of the VAE decoder's single-head attention (rows like af_softmax_rows)
template <int RS>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const half_t* __restrict__ p, const half_t* __restrict__ dp,
                                                               half_t* __restrict__ ds, long rows, int L) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const half_t *pr = p + row * L, *dr = dp + row * L;
  half8_t pv[RS], dv[RS];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < RS; ++j) {
    const int c = (lane + 64 * j) * 8;
    if (c < L) {
      pv[j] = *reinterpret_cast<const half8_t*>(pr + c);
      dv[j] = *reinterpret_cast<const half8_t*>(dr + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += (float)pv[j][e] * (float)dv[j][e];
    }
  }
  dot = af_wave_sum(dot);
  half_t* sr = ds + row * L;
#pragma unroll
  for (int j = 0; j < RS; ++j) {
    const int c = (lane + 64 * j) * 8;
    if (c < L) {
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)pv[j][e] * ((float)dv[j][e] - dot));
      *reinterpret_cast<half8_t*>(sr + c) = o;
    }
  }
}

extern "C" int af_softmax_rows(const void* x, void* y, int64_t rows, int L, void* stream) {
  AF_REQUIRE(x && y && rows > 0 && L > 0 && L % 8 == 0, "af_softmax_rows: L must be a positive multiple of 8");
  AF_SUPPORTED(L <= 64 * 8 * 8, "af_softmax_rows: L > 4096");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (L <= 512) hipLaunchKernelGGL(softmax_rows_kernel<1>, grid, blk, 0, s, (const half_t*)x, (half_t*)y, (long)rows, L);
  else if (L <= 1024) hipLaunchKernelGGL(softmax_rows_kernel<2>, grid, blk, 0, s, (const half_t*)x, (half_t*)y, (long)rows, L);
  else if (L <= 2048) hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, blk, 0, s, (const half_t*)x, (half_t*)y, (long)rows, L);
  else hipLaunchKernelGGL(softmax_rows_kernel<8>, grid, blk, 0, s, (const half_t*)x, (half_t*)y, (long)rows, L);
  return af_check_launch("af_softmax_rows");
}

extern "C" int af_softmax_rows_bwd(const void* p, const void* dp, void* ds, int64_t rows, int L, void* stream) {
  AF_REQUIRE(p && dp && ds && rows > 0 && L > 0 && L % 8 == 0, "af_softmax_rows_bwd: L must be a positive multiple of 8");
  AF_SUPPORTED(L <= 64 * 8 * 8, "af_softmax_rows_bwd: L > 4096");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const half_t *pp = (const half_t*)p, *dd = (const half_t*)dp;
  if (L <= 512) hipLaunchKernelGGL(softmax_rows_bwd_kernel<1>, grid, blk, 0, s, pp, dd, (half_t*)ds, (long)rows, L);
  else if (L <= 1024) hipLaunchKernelGGL(softmax_rows_bwd_kernel<2>, grid, blk, 0, s, pp, dd, (half_t*)ds, (long)rows, L);
  else if (L <= 2048) hipLaunchKernelGGL(softmax_rows_bwd_kernel<4>, grid, blk, 0, s, pp, dd, (half_t*)ds, (long)rows, L);
  else hipLaunchKernelGGL(softmax_rows_bwd_kernel<8>, grid, blk, 0, s, pp, dd, (half_t*)ds, (long)rows, L);
  return af_check_launch("af_softmax_rows_bwd");
}

// Pull a byte range (the next layers' packed weights) toward the GPU's caches: every 128-byte line is read once, nothing is written.
// Launched on a side stream ahead of the GEMM that will stream the range (memory-side Infinity Cache hits instead of HBM misses).
__global__ __launch_bounds__(256) void prefetch_kernel(const unsigned* __restrict__ p, long nlines, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nlines; i += (long)gridDim.x * 256) {
    acc ^= p[i * 32];                                                  // one dword per 128-byte line fetches the line
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;                       // keeps the loads alive; practically never taken
}

extern "C" int af_prefetch(const void* ptr, int64_t bytes, void* stream) {
  AF_REQUIRE(ptr && bytes >= 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0, "af_prefetch: 16-byte aligned range");
  const long nlines = bytes / 128;
  if (nlines == 0) return 0;
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const unsigned grid = (unsigned)((nlines + 255) / 256 < 2048 ? (nlines + 255) / 256 : 2048);
  hipLaunchKernelGGL(prefetch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned*)ptr, nlines, (unsigned*)nullptr);
  return af_check_launch("af_prefetch");
}

extern "C" int af_prefetch_ex(const void* ptr, int64_t bytes, int max_workgroups, void* stream) {
  AF_REQUIRE(ptr && bytes >= 0 && max_workgroups > 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0, "af_prefetch_ex: 16-byte aligned range, max_workgroups > 0");
  const long nlines = bytes / 128;
  if (nlines == 0) return 0;
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long want = (nlines + 255) / 256;
  const unsigned grid = (unsigned)(want < max_workgroups ? want : max_workgroups);
  hipLaunchKernelGGL(prefetch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned*)ptr, nlines, (unsigned*)nullptr);
  return af_check_launch("af_prefetch_ex");
}

extern "C" int af_mask_pairs(void* p, const void* cls, int N, void* stream) {
  AF_REQUIRE(p && cls && N > 0 && N % 8 == 0, "af_mask_pairs: N must be a positive multiple of 8");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = (long)N * (N / 8);
  hipLaunchKernelGGL(mask_pairs_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (half_t*)p,
                     (const unsigned char*)cls, N, n8);
  return af_check_launch("af_mask_pairs");
}

extern "C" int af_silu_f16(const void* x, void* y, int64_t n, void* stream) {
  AF_REQUIRE(x && y && n > 0, "af_silu_f16: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(silu_kernel, grid1d((long)n), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, (long)n);
  return af_check_launch("af_silu_f16");
}
