// af_face.hip -- the non-matmul kernels of the ArcFace ResNetFace-18 IR-SE face encoder (reference
// evaluation/arcface_resnet.py:62-97 IRBlock, :139-154 SEBlock, :157-217 ResNetFace).  The convolutions and the FC layers run
// on af_gemm (eval-mode BatchNorm folded into weights / bias on the host); what is left is HBM-bound element-wise work on
// NHWC fp16 tensors, 8 channels (16 B) per lane so every wave issues full 1 KiB coalesced loads:
//   af_affine_prelu      y = prelu(x * scale[c] + shift[c])          (pre-conv BatchNorm bn0; post-conv PReLU)
//   af_maxpool2x2        2x2 / stride 2 max
//   af_global_avgpool    mean over H*W per (b, c)                    (SE squeeze)
//   af_se_residual_prelu y = prelu(x * sigmoid(s[b,c]) + residual)   (SE excite + shortcut + PReLU in one pass)
#include "af_common.h"

namespace {

__device__ __forceinline__ float prelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

__global__ __launch_bounds__(256) void affine_prelu_kernel(const half_t* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ slope,
                                                           half_t* __restrict__ y, long n8, int C8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c0 = (int)(i % C8) * 8;
  const half8_t v = *reinterpret_cast<const half8_t*>(x + i * 8);
  const float sl = slope ? slope[0] : 1.0f;
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f = (float)v[e];
    if (scale) f = f * scale[c0 + e] + shift[c0 + e];
    o[e] = (half_t)prelu(f, sl);
  }
  *reinterpret_cast<half8_t*>(y + i * 8) = o;
}

__global__ __launch_bounds__(256) void maxpool2x2_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int Ho, int Wo,
                                                         int C8, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)(i % C8);
  long p = i / C8;
  const int wo = (int)(p % Wo);
  p /= Wo;
  const int ho = (int)(p % Ho);
  const long b = p / Ho;
  const long W = 2L * Wo, C = 8L * C8;
  const half_t* s = x + ((b * 2 * Ho + 2 * ho) * W + 2 * wo) * C + c * 8;
  const half8_t a = *reinterpret_cast<const half8_t*>(s), bq = *reinterpret_cast<const half8_t*>(s + C);
  const half8_t cq = *reinterpret_cast<const half8_t*>(s + W * C), d = *reinterpret_cast<const half8_t*>(s + W * C + C);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)fmaxf(fmaxf((float)a[e], (float)bq[e]), fmaxf((float)cq[e], (float)d[e]));
  *reinterpret_cast<half8_t*>(y + i * 8) = o;
}

// one block per (b, 64-channel slab): 4 waves stride over pixels, each lane owns one channel; LDS fold
__global__ __launch_bounds__(256) void global_avgpool_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, int HW, int C) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C)
    for (int p = w; p < HW; p += 4) s += (float)x[((size_t)b * HW + p) * C + c];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C) out[(size_t)b * C + c] = (half_t)((red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)HW);
}

__global__ __launch_bounds__(256) void se_residual_prelu_kernel(const half_t* __restrict__ x, const half_t* __restrict__ se,
                                                                const half_t* __restrict__ residual, const float* __restrict__ slope,
                                                                half_t* __restrict__ y, long n8, int C8, long per_batch8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c0 = (int)(i % C8) * 8;
  const long b = i / per_batch8;
  const half8_t v = *reinterpret_cast<const half8_t*>(x + i * 8), r = *reinterpret_cast<const half8_t*>(residual + i * 8);
  const float sl = slope[0];
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float g = 1.0f;
    if (se) g = 1.0f / (1.0f + __expf(-(float)se[b * C8 * 8 + c0 + e]));
    o[e] = (half_t)prelu((float)v[e] * g + (float)r[e], sl);
  }
  *reinterpret_cast<half8_t*>(y + i * 8) = o;
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int af_affine_prelu(const void* x, const void* scale, const void* shift, const void* slope, void* y, int64_t rows, int C,
                               void* stream) {
  AF_REQUIRE(x && y && rows > 0 && C > 0 && C % 8 == 0, "af_affine_prelu: C must be a positive multiple of 8");
  AF_REQUIRE((scale == nullptr) == (shift == nullptr), "af_affine_prelu: scale and shift go together");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = rows * (C / 8);
  hipLaunchKernelGGL(affine_prelu_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const float*)scale,
                     (const float*)shift, (const float*)slope, (half_t*)y, n8, C / 8);
  return af_check_launch("af_affine_prelu");
}

extern "C" int af_maxpool2x2(const void* x, void* y, int B, int Ho, int Wo, int C, void* stream) {
  AF_REQUIRE(x && y && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "af_maxpool2x2: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = (long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool2x2_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, Ho, Wo, C / 8, n8);
  return af_check_launch("af_maxpool2x2");
}

extern "C" int af_global_avgpool(const void* x, void* out, int B, int HW, int C, void* stream) {
  AF_REQUIRE(x && out && B > 0 && HW > 0 && C > 0, "af_global_avgpool: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)out,
                     HW, C);
  return af_check_launch("af_global_avgpool");
}

extern "C" int af_se_residual_prelu(const void* x, const void* se_logits, const void* residual, const void* slope, void* y, int B,
                                    int HW, int C, void* stream) {
  AF_REQUIRE(x && residual && slope && y && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "af_se_residual_prelu: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = (long)B * HW * (C / 8);
  hipLaunchKernelGGL(se_residual_prelu_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)se_logits,
                     (const half_t*)residual, (const float*)slope, (half_t*)y, n8, C / 8, (long)HW * (C / 8));
  return af_check_launch("af_se_residual_prelu");
}
