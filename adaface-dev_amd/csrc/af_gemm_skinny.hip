// af_gemm_skinny.hip -- tile 11 of af_gemm: 1x1 GEMMs that are too small to fill the chip with K loops of barriers.
//
// Where it is used: the CLIP encoders' linears (M = 22 .. 388 tokens, K = 768 / 3072), the attention projections of the 16x16 / 8x8
// U-Net levels (M = 512 .. 2048, weights read once), the adapters' rank-r projections, weight gradients with few tokens.  In the
// register-staged 64 x 64 kernel these run a K loop whose every step waits for its global loads behind a workgroup barrier
// (12 - 20 us for < 2 GFLOP); split-K shortens the loop but pays a second launch (~6 us) for the reduction.
//
// Design (gfx950): one workgroup = 4 waves = one 64 x 64 output tile; the FOUR WAVES SPLIT K between them.  Each wave loads its
// MFMA fragments straight from global memory into registers (16 bytes per lane: 16 rows x 32 k per v_mfma_f32_16x16x32_f16
// operand, rows of W and of the activations alike), no LDS staging, no barrier inside the K loop, two register buffers of KS
// k-steps each so that up to 2 * KS * 8 loads per lane are in flight.  The four partial 64 x 64 accumulators meet in LDS once
// (64 KB, conflict-free 16-byte rows), wave w sums and finishes the w-th 16-column strip with the standard epilogue
// (bias, row bias, SiLU / quick-GELU, residual, fp16 or fp32 store).  Summation order is fixed (wave 0 .. 3): deterministic.
#include "af_common.h"

namespace {

struct SkinnyDev {
  const half_t* a;
  const half_t* wt;
  const float* bias;
  const half_t* rowbias;
  const half_t* residual;
  void* out;
  int M, N, c1, kpad, lda;
  int rows_per_batch, ld_rowbias, act, ld_out, out_f32;
  int tiles_n;
};

template <int KS>
__global__ __launch_bounds__(256, 2) void af_gemm_skinny_kernel(SkinnyDev p) {
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  floatx4* red = reinterpret_cast<floatx4*>(af_smem);            // [wave][tn * 4 + tm][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x - tile_m * p.tiles_n;

  // fragment row pointers (k offset added per step); rows past M are clamped (their accumulators are never stored)
  const half_t* wp[4];
  const half_t* ap[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    wp[t] = p.wt + (size_t)(tile_n * 64 + t * 16 + l16) * p.kpad + kq * 8;
    const int m = min(tile_m * 64 + t * 16 + l16, p.M - 1);
    ap[t] = p.a + (size_t)m * p.lda + kq * 8;
  }
  const int nk = p.kpad >> 5;                                    // k-steps of 32
  const int per = (nk + 3) >> 2;
  const int s0 = wave * per, s1 = min(nk, s0 + per);

  floatx4 acc[4][4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};

  half8_t wb0[KS][4], xb0[KS][4], wb1[KS][4], xb1[KS][4];
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#define SK_LOAD(WB, XB, S)                                                                  \
  _Pragma("unroll") for (int u = 0; u < KS; ++u) {                                           \
    const int st = (S) + u;                                                                  \
    if (st < s1) {                                                                           \
      const int k = st << 5;                                                                 \
      const bool ka = k + kq * 8 < p.c1; /* c1 % 8 == 0: a chunk is all inside or all pad */ \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                        \
        WB[u][t] = *reinterpret_cast<const half8_t*>(wp[t] + k);                             \
        XB[u][t] = ka ? *reinterpret_cast<const half8_t*>(ap[t] + k) : zero8;                \
      }                                                                                      \
    }                                                                                        \
  }
#define SK_COMPUTE(WB, XB, S)                                                                             \
  _Pragma("unroll") for (int u = 0; u < KS; ++u) {                                                         \
    if ((S) + u < s1) {                                                                                    \
      _Pragma("unroll") for (int tn = 0; tn < 4; ++tn)                                                     \
      _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                                     \
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WB[u][tn], XB[u][tm], acc[tn][tm], 0, 0, 0);  \
    }                                                                                                      \
  }
  SK_LOAD(wb0, xb0, s0)
  for (int s = s0; s < s1; s += 2 * KS) {
    SK_LOAD(wb1, xb1, s + KS)
    SK_COMPUTE(wb0, xb0, s)
    SK_LOAD(wb0, xb0, s + 2 * KS)
    SK_COMPUTE(wb1, xb1, s + KS)
  }
#undef SK_LOAD
#undef SK_COMPUTE

  // ---- the four K-partials meet in LDS; wave w finishes output columns [16 w, 16 w + 16) of the tile
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) red[(wave * 16 + tn * 4 + tm) * 64 + lane] = acc[tn][tm];
  __syncthreads();
  const int n0 = tile_n * 64 + wave * 16 + 4 * kq;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = tile_m * 64 + tm * 16 + l16;
    floatx4 v = red[(0 * 16 + wave * 4 + tm) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[(w * 16 + wave * 4 + tm) * 64 + lane];
    if (m >= p.M || n0 >= p.N) continue;
    if (p.bias) v += *reinterpret_cast<const floatx4*>(p.bias + n0);
    if (p.rowbias) {
      const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
    }
    if (p.act == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = af_silu(v[i]);
    } else if (p.act == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = v[i] / (1.0f + __expf(-1.702f * v[i]));
    }
    if (p.residual) {
      const half4_t rv = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * p.N + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
    }
    if (p.out_f32) {
      *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ld_out + n0) = v;
    } else {
      const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ld_out + n0) = h;
    }
  }
}

}  // namespace

// 0 = launched, 1 = outside this kernel's scope (the caller falls back to the register-staged tiles)
int af_gemm_skinny_try_launch(const af_gemm_desc* d, hipStream_t stream) {
  if (d->taps != 1 || d->a2 != nullptr || d->c2 != 0 || d->act == AF_ACT_GEGLU || d->out_mode == AF_OUT_SPLIT_T) return 1;
  if (d->c1 % 8 != 0 || d->kpad % 32 != 0 || d->N % 4 != 0 || d->M <= 0) return 1;
  const int lda = d->lda1 ? d->lda1 : d->c1;
  const int ld_out = d->ld_out ? d->ld_out : d->N;
  if (lda % 8 != 0 || ld_out % 4 != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(d->a1) & 15) != 0) return 1;
  SkinnyDev p;
  p.a = (const half_t*)d->a1;
  p.wt = (const half_t*)d->wt;
  p.bias = (const float*)d->bias;
  p.rowbias = (const half_t*)d->rowbias;
  p.residual = (const half_t*)d->residual;
  p.out = d->out;
  p.M = d->M;
  p.N = d->N;
  p.c1 = d->c1;
  p.kpad = d->kpad;
  p.lda = lda;
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.ld_rowbias = d->ld_rowbias;
  p.act = d->act == AF_ACT_SILU ? 1 : (d->act == AF_ACT_QUICKGELU ? 3 : 0);
  p.ld_out = ld_out;
  p.out_f32 = d->out_mode == AF_OUT_F32;
  p.tiles_n = (d->N + 63) / 64;
  const int tiles_m = (d->M + 63) / 64;
  // the packed weight has Npad = ceil(N / 128) * 128 rows, so the last 64-column tile's rows exist
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(af_gemm_skinny_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_done = true;
  }
  hipLaunchKernelGGL(af_gemm_skinny_kernel<2>, dim3((unsigned)(tiles_m * p.tiles_n)), dim3(256), 65536, stream, p);
  return 0;
}
