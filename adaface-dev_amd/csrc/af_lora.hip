// af_lora.hip -- element-wise / gather kernels of the TRAINABLE DoRA adapters on the U-Net's up_blocks.3 convolutions
// (reference adaface/diffusers_attn_lora_capture.py:541-591 attaches peft DoRA Conv2d adapters, rank 192; peft's
// DoraConv2dLayer.forward computes  y = base(x) + (s - 1) * conv(xd, W) + s * scaling * B(A(xd)),  xd = dropout(x)).
// The convolutions and their input gradients run on af_gemm; this file holds what is left:
//   af_dora_combine   y = y0 + u[c] * c2 + v[c] * lb                    (u = s - 1, v = s * scaling, per output channel)
//   af_mul_f16        xd = x * mask                                     (mask holds 0 or 1 / (1 - p))
//   af_im2col3x3      X9[m, (ky, kx, c)] = x[b, oy*stride + ky - 1, ox*stride + kx - 1, c] (zero halo): the explicit
//                     operand of the 3x3 WEIGHT gradient  dA = d_t^T . X9  (a GEMM reducing over pixels; only the two
//                     adapter-carrying ResBlocks need it, so the implicit-GEMM loader is not extended for it)
#include "af_common.h"

namespace {

__global__ __launch_bounds__(256) void dora_combine_kernel(const half_t* __restrict__ y0, const half_t* __restrict__ c2,
                                                           const half_t* __restrict__ lb, const float* __restrict__ u,
                                                           const float* __restrict__ v, half_t* __restrict__ out, long n8, int C8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c0 = (int)(i % C8) * 8;
  const half8_t a = *reinterpret_cast<const half8_t*>(y0 + i * 8), b = *reinterpret_cast<const half8_t*>(c2 + i * 8);
  const half8_t l = *reinterpret_cast<const half8_t*>(lb + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + u[c0 + e] * (float)b[e] + v[c0 + e] * (float)l[e]);
  *reinterpret_cast<half8_t*>(out + i * 8) = o;
}

__global__ __launch_bounds__(256) void mul_f16_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                      half_t* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const half8_t x = *reinterpret_cast<const half8_t*>(a + i * 8), y = *reinterpret_cast<const half8_t*>(b + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x[e] * (float)y[e]);
  *reinterpret_cast<half8_t*>(out + i * 8) = o;
}

// one thread per (output pixel, tap, 8-channel chunk): 16-byte gather, 16-byte store
__global__ __launch_bounds__(256) void im2col3x3_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, int B, int H, int W,
                                                        int C8, int Ho, int Wo, int stride, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ch = (int)(i % C8);
  long r = i / C8;
  const int tap = (int)(r % 9);
  r /= 9;
  const int ox = (int)(r % Wo);
  r /= Wo;
  const int oy = (int)(r % Ho);
  const int b = (int)(r / Ho);
  const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
  half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
    v = *reinterpret_cast<const half8_t*>(x + (((size_t)b * H + iy) * W + ix) * (C8 * 8) + ch * 8);
  *reinterpret_cast<half8_t*>(out + i * 8) = v;
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int af_dora_combine(const void* y0, const void* c2, const void* lb, const void* u, const void* v, void* out,
                               int64_t rows, int C, void* stream) {
  AF_REQUIRE(y0 && c2 && lb && u && v && out && rows > 0 && C > 0 && C % 8 == 0, "af_dora_combine: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n8 = rows * (C / 8);
  hipLaunchKernelGGL(dora_combine_kernel, g1(n8), dim3(256), 0, (hipStream_t)stream, (const half_t*)y0, (const half_t*)c2,
                     (const half_t*)lb, (const float*)u, (const float*)v, (half_t*)out, n8, C / 8);
  return af_check_launch("af_dora_combine");
}

extern "C" int af_mul_f16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  AF_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "af_mul_f16: n must be a positive multiple of 8");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(mul_f16_kernel, g1(n / 8), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, (half_t*)out,
                     (long)(n / 8));
  return af_check_launch("af_mul_f16");
}

extern "C" int af_im2col3x3(const void* x, void* out, int B, int H, int W, int C, int stride, void* stream) {
  AF_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (stride == 1 || stride == 2), "af_im2col3x3: bad argument");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const long n = (long)B * Ho * Wo * 9 * (C / 8);
  hipLaunchKernelGGL(im2col3x3_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)out, B, H, W, C / 8, Ho, Wo,
                     stride, n);
  return af_check_launch("af_im2col3x3");
}
