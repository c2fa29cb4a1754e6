// af_face.hip -- the non-matmul kernels of the ArcFace ResNetFace-18 IR-SE face encoder (reference
// evaluation/arcface_resnet.py:62-97 IRBlock, :139-154 SEBlock, :157-217 ResNetFace).  The convolutions and the FC layers run
// on af_gemm (eval-mode BatchNorm folded into weights / bias on the host); what is left is HBM-bound element-wise work on
// NHWC fp16 tensors, 8 channels (16 B) per lane so every wave issues full 1 KiB coalesced loads:
//   af_affine_prelu      y = prelu(x * scale[c] + shift[c])          (pre-conv BatchNorm bn0; post-conv PReLU)
//   af_maxpool2x2        2x2 / stride 2 max
//   af_global_avgpool    mean over H*W per (b, c)                    (SE squeeze)
//   af_se_residual_prelu y = prelu(x * sigmoid(s[b,c]) + residual)   (SE excite + shortcut + PReLU in one pass)
// and their input-gradient kernels (the encoder is frozen -- arcface_wrapper.py:65-76 -- but the alignment loss differentiates
// THROUGH it into the decoded image, ddpm.py:2511-2535, so only d/dx is ever needed, never a parameter gradient):
//   af_affine_prelu_bwd       dx = dy * prelu'(x * scale + shift) * scale
//   af_maxpool2x2_bwd         dy routed to the first maximum of each 2x2 window (torch's argmax rule)
//   af_se_gate_grad           d(se logits)[b,c] / HW = sigmoid' * mean_hw(dpre * x)      (pre = x * sigmoid(s) + residual)
//   af_se_residual_prelu_bwd  dx = dpre * sigmoid(s) + dpool[b,c] ;  dresidual = dpre    (dpool: the squeeze branch's gradient)
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

// one block per (b, 64-channel slab): 1024 threads = 8 channel chunks (16-byte loads) x 128 pixel slots, LDS fold.  (The first form -- 4 waves, one
// channel per lane, 2-byte loads, HW / 4 dependent iterations -- took 83 us per call at [B, 4096, 64]: a handful of blocks walking 512 KB each in
// 1,024 two-byte steps.)
__global__ __launch_bounds__(1024) void global_avgpool_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, int HW, int C, int vec) {
  __shared__ float red[128][65];
  const int t = threadIdx.x, ch = t & 7, ps = t >> 3;
  const int b = blockIdx.y, c0 = blockIdx.x * 64 + ch * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (vec && c0 + 8 <= C) {                          // vec: C % 8 == 0 and a 16-byte aligned base
    const half_t* xp = x + (size_t)b * HW * C + c0;
    for (int p = ps; p < HW; p += 128) {
      const half8_t v = *reinterpret_cast<const half8_t*>(xp + (size_t)p * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  } else {
    for (int e = 0; e < 8; ++e)
      if (c0 + e < C)
        for (int p = ps; p < HW; p += 128) s[e] += (float)x[((size_t)b * HW + p) * C + c0 + e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ps][ch * 8 + e] = s[e];
  __syncthreads();
  if (t < 64 && blockIdx.x * 64 + t < C) {
    float a = 0.f;
    for (int k = 0; k < 128; ++k) a += red[k][t];
    out[(size_t)b * C + blockIdx.x * 64 + t] = (half_t)(a / (float)HW);
  }
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
    if (se) g = af_sigmoid((float)se[b * C8 * 8 + c0 + e]);
    o[e] = (half_t)prelu((float)v[e] * g + (float)r[e], sl);
  }
  *reinterpret_cast<half8_t*>(y + i * 8) = o;
}

__global__ __launch_bounds__(256) void affine_prelu_bwd_kernel(const half_t* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ slope,
                                                               const half_t* __restrict__ dy, half_t* __restrict__ dx, long n8, int C8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c0 = (int)(i % C8) * 8;
  const half8_t g = *reinterpret_cast<const half8_t*>(dy + i * 8);
  half8_t v = g;
  if (slope) v = *reinterpret_cast<const half8_t*>(x + i * 8);
  const float sl = slope ? slope[0] : 1.0f;
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f = (float)v[e], d = (float)g[e];
    if (scale) {
      f = f * scale[c0 + e] + shift[c0 + e];
      d *= scale[c0 + e];
    }
    o[e] = (half_t)((slope && !(f > 0.f)) ? d * sl : d);        // torch's PReLU gradient at 0 is the slope
  }
  *reinterpret_cast<half8_t*>(dx + i * 8) = o;
}

__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                             half_t* __restrict__ dx, int Ho, int Wo, int C8, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)(i % C8);
  long p = i / C8;
  const int wo = (int)(p % Wo);
  p /= Wo;
  const int ho = (int)(p % Ho);
  const long b = p / Ho;
  const long W = 2L * Wo, C = 8L * C8;
  const long o00 = ((b * 2 * Ho + 2 * ho) * W + 2 * wo) * C + c * 8;
  const half8_t a = *reinterpret_cast<const half8_t*>(x + o00), bq = *reinterpret_cast<const half8_t*>(x + o00 + C);
  const half8_t cq = *reinterpret_cast<const half8_t*>(x + o00 + W * C), d = *reinterpret_cast<const half8_t*>(x + o00 + W * C + C);
  const half8_t g = *reinterpret_cast<const half8_t*>(dy + i * 8);
  half8_t ga, gb, gc, gd;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int k = 0;                                            // first maximum in window scan order (a later one must be strictly greater)
    float m = (float)a[e];
    if ((float)bq[e] > m) m = (float)bq[e], k = 1;
    if ((float)cq[e] > m) m = (float)cq[e], k = 2;
    if ((float)d[e] > m) k = 3;
    const half_t z = (half_t)0.f;
    ga[e] = k == 0 ? g[e] : z;
    gb[e] = k == 1 ? g[e] : z;
    gc[e] = k == 2 ? g[e] : z;
    gd[e] = k == 3 ? g[e] : z;
  }
  *reinterpret_cast<half8_t*>(dx + o00) = ga;
  *reinterpret_cast<half8_t*>(dx + o00 + C) = gb;
  *reinterpret_cast<half8_t*>(dx + o00 + W * C) = gc;
  *reinterpret_cast<half8_t*>(dx + o00 + W * C + C) = gd;
}

// one block per (b, 64-channel slab) like the squeeze: sum over pixels of dpre * x, times sigmoid' / HW
__global__ __launch_bounds__(256) void se_gate_grad_kernel(const half_t* __restrict__ x, const half_t* __restrict__ se,
                                                           const half_t* __restrict__ residual, const float* __restrict__ slope,
                                                           const half_t* __restrict__ dy, half_t* __restrict__ dgl, int HW, int C) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, c = blockIdx.x * 64 + lane;
  const float sl = slope[0];
  float s = 0.f, g = 0.f;
  if (c < C) {
    g = af_sigmoid((float)se[(size_t)b * C + c]);
    for (int p = w; p < HW; p += 4) {
      const size_t o = ((size_t)b * HW + p) * C + c;
      const float xv = (float)x[o];
      const float pre = xv * g + (float)residual[o];
      const float d = (float)dy[o];
      s += (pre > 0.f ? d : d * sl) * xv;
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C)
    dgl[(size_t)b * C + c] = (half_t)((red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * g * (1.0f - g) / (float)HW);
}

__global__ __launch_bounds__(256) void se_residual_prelu_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ se,
                                                                    const half_t* __restrict__ residual, const float* __restrict__ slope,
                                                                    const half_t* __restrict__ dy, const half_t* __restrict__ dpool,
                                                                    half_t* __restrict__ dx, half_t* __restrict__ dres, long n8, int C8,
                                                                    long per_batch8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c0 = (int)(i % C8) * 8;
  const long b = i / per_batch8;
  const half8_t v = *reinterpret_cast<const half8_t*>(x + i * 8), r = *reinterpret_cast<const half8_t*>(residual + i * 8);
  const half8_t gy = *reinterpret_cast<const half8_t*>(dy + i * 8);
  const float sl = slope[0];
  half8_t ox, orr;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float g = 1.0f, dp = 0.f;
    if (se) g = af_sigmoid((float)se[b * C8 * 8 + c0 + e]);
    if (dpool) dp = (float)dpool[b * C8 * 8 + c0 + e];
    const float pre = (float)v[e] * g + (float)r[e];
    const float d = pre > 0.f ? (float)gy[e] : (float)gy[e] * sl;
    ox[e] = (half_t)(d * g + dp);
    orr[e] = (half_t)d;
  }
  *reinterpret_cast<half8_t*>(dx + i * 8) = ox;
  *reinterpret_cast<half8_t*>(dres + i * 8) = orr;
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
  const int vec = C % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)out,
                     HW, C, vec);
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

extern "C" int af_affine_prelu_bwd(const void* x, const void* scale, const void* shift, const void* slope, const void* dy, void* dx,
                                   int64_t rows, int C, void* stream) {
  AF_REQUIRE(dy && dx && rows > 0 && C > 0 && C % 8 == 0, "af_affine_prelu_bwd: C must be a positive multiple of 8");
  AF_REQUIRE((scale == nullptr) == (shift == nullptr), "af_affine_prelu_bwd: scale and shift go together");
  AF_REQUIRE(x || !slope, "af_affine_prelu_bwd: the PReLU gradient needs the forward input x");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = rows * (C / 8);
  hipLaunchKernelGGL(affine_prelu_bwd_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const float*)scale,
                     (const float*)shift, (const float*)slope, (const half_t*)dy, (half_t*)dx, n8, C / 8);
  return af_check_launch("af_affine_prelu_bwd");
}

extern "C" int af_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int B, int Ho, int Wo, int C, void* stream) {
  AF_REQUIRE(x && dy && dx && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "af_maxpool2x2_bwd: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = (long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)dy, (half_t*)dx,
                     Ho, Wo, C / 8, n8);
  return af_check_launch("af_maxpool2x2_bwd");
}

extern "C" int af_se_gate_grad(const void* x, const void* se_logits, const void* residual, const void* slope, const void* dy, void* dgl,
                               int B, int HW, int C, void* stream) {
  AF_REQUIRE(x && se_logits && residual && slope && dy && dgl && B > 0 && HW > 0 && C > 0, "af_se_gate_grad: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(se_gate_grad_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (const half_t*)se_logits, (const half_t*)residual, (const float*)slope, (const half_t*)dy, (half_t*)dgl, HW, C);
  return af_check_launch("af_se_gate_grad");
}

extern "C" int af_se_residual_prelu_bwd(const void* x, const void* se_logits, const void* residual, const void* slope, const void* dy,
                                        const void* dpool, void* dx, void* dres, int B, int HW, int C, void* stream) {
  AF_REQUIRE(x && residual && slope && dy && dx && dres && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "af_se_residual_prelu_bwd: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = (long)B * HW * (C / 8);
  hipLaunchKernelGGL(se_residual_prelu_bwd_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)se_logits,
                     (const half_t*)residual, (const float*)slope, (const half_t*)dy, (const half_t*)dpool, (half_t*)dx, (half_t*)dres, n8,
                     C / 8, (long)HW * (C / 8));
  return af_check_launch("af_se_residual_prelu_bwd");
}
