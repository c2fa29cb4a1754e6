// af_bwd.hip -- backward-pass kernels that are not matmul-shaped (those reuse af_gemm with
// transposed / flipped packed weights): GroupNorm(+SiLU) / LayerNorm / GEGLU input gradients,
// the adjoint of nearest-x2 upsampling, token<->channel-major transposes for the attention
// backward, gradient accumulation, and the fused cautious-AdamW step.
//
// The U-Net base weights are frozen in the reference (ddpm.py:4131-4132), so the backward is
// activation-gradient only: no gamma/beta/weight gradients are produced here.
#include "af_common.h"

namespace {

constexpr int GB_NBLK = 32;
constexpr int GB_MAXG = 32;

struct GnBwdArgs {
  const half_t* x1;
  const half_t* x2;
  int c1, c2, C, CP;
  const float* gamma;
  const float* beta;
  const float* stats;  // [B][groups][2] (mean, rstd) from the forward
  const half_t* dy;    // [B][HW][C]
  const half_t* add;   // optional [B][HW][C], added to dx
  half_t* dx1;         // [B][HW][c1]
  half_t* dx2;         // [B][HW][c2] (c2 > 0)
  int B, HW, groups, cpg, silu;
  float* ws;  // [B][GB_NBLK][GB_MAXG][2]
  int ppb;
};

__device__ __forceinline__ half8_t gb_load(const GnBwdArgs& a, int b, int pix, int c0) {
  const size_t row = (size_t)b * a.HW + pix;
  if (c0 < a.c1) return *reinterpret_cast<const half8_t*>(a.x1 + row * a.c1 + c0);
  return *reinterpret_cast<const half8_t*>(a.x2 + row * a.c2 + (c0 - a.c1));
}

// d/du [u * sigmoid(u)]
__device__ __forceinline__ float silu_grad(float u) {
  const float s = af_sigmoid(u);
  return s * (1.0f + u * (1.0f - s));
}

// per-thread constants for its 8 channels: xhat = x*ka + kb ; u = xhat*gamma + beta
struct ChanConst {
  float ka[8], kb[8], g[8], bt[8];
};

template <int CT>
__device__ __forceinline__ void gb_consts(const GnBwdArgs& a, int b, int chunk0, ChanConst (&cc)[CT]) {
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = chunk0 + 256 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = min(ch * 8 + e, a.C - 1);
      const int g = c / a.cpg;
      const float mean = a.stats[((size_t)b * a.groups + g) * 2 + 0];
      const float rstd = a.stats[((size_t)b * a.groups + g) * 2 + 1];
      cc[j].ka[e] = rstd;
      cc[j].kb[e] = -mean * rstd;
      cc[j].g[e] = a.gamma[c];
      cc[j].bt[e] = a.beta[c];
    }
  }
}

// pass 1: partial sums per group of  g = dy*act'(u)*gamma  and  g*xhat
template <int CT>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  float* red = reinterpret_cast<float*>(af_smem);
  const int t = threadIdx.x, b = blockIdx.y, blk = blockIdx.x;
  const int slots = CT == 1 ? a.ppb : 1;
  const int slot = CT == 1 ? t / a.CP : 0;
  const int chunk0 = CT == 1 ? t - slot * a.CP : t;
  const bool active = CT == 1 ? (slot < slots) : true;
  const int per = (a.HW + GB_NBLK - 1) / GB_NBLK;
  const int p0 = blk * per, p1 = min(a.HW, p0 + per);
  ChanConst cc[CT];
  gb_consts<CT>(a, b, chunk0, cc);
  float s1[CT][8], s2[CT][8];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[j][e] = s2[j][e] = 0.f;
  if (active) {
    for (int pix = p0 + slot; pix < p1; pix += slots) {
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int ch = chunk0 + 256 * j;
        if (ch < a.CP) {
          const half8_t xv = gb_load(a, b, pix, ch * 8);
          const half8_t dv = *reinterpret_cast<const half8_t*>(a.dy + ((size_t)b * a.HW + pix) * a.C + ch * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (float)xv[e] * cc[j].ka[e] + cc[j].kb[e];
            float g = (float)dv[e] * cc[j].g[e];
            if (a.silu) g *= silu_grad(xh * cc[j].g[e] + cc[j].bt[e]);
            s1[j][e] += g;
            s2[j][e] += g * xh;
          }
        }
      }
    }
  }
  float* r1 = red;
  float* r2 = red + slots * a.C;
  if (active) {
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ch = chunk0 + 256 * j;
      if (ch < a.CP) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          r1[slot * a.C + ch * 8 + e] = s1[j][e];
          r2[slot * a.C + ch * 8 + e] = s2[j][e];
        }
      }
    }
  }
  __syncthreads();
  for (int c = t; c < a.C; c += 256) {
    float u = 0.f, v = 0.f;
    for (int sl = 0; sl < slots; ++sl) {
      u += r1[sl * a.C + c];
      v += r2[sl * a.C + c];
    }
    r1[c] = u;
    r2[c] = v;
  }
  __syncthreads();
  if (t < a.groups) {
    float u = 0.f, v = 0.f;
    for (int c = t * a.cpg; c < (t + 1) * a.cpg; ++c) {
      u += r1[c];
      v += r2[c];
    }
    float* w = a.ws + (((size_t)b * GB_NBLK + blk) * GB_MAXG + t) * 2;
    w[0] = u;
    w[1] = v;
  }
}

// pass 2: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) + add
template <int CT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdArgs a) {
  __shared__ float mm[GB_MAXG][2];
  const int t = threadIdx.x, b = blockIdx.y;
  if (t < a.groups) {
    float u = 0.f, v = 0.f;
    for (int k = 0; k < GB_NBLK; ++k) {
      const float* w = a.ws + (((size_t)b * GB_NBLK + k) * GB_MAXG + t) * 2;
      u += w[0];
      v += w[1];
    }
    const float inv_n = 1.0f / ((float)a.HW * (float)a.cpg);
    mm[t][0] = u * inv_n;
    mm[t][1] = v * inv_n;
  }
  __syncthreads();
  const int slots = CT == 1 ? a.ppb : 1;
  const int slot = CT == 1 ? t / a.CP : 0;
  const int chunk0 = CT == 1 ? t - slot * a.CP : t;
  if (CT == 1 && slot >= slots) return;
  ChanConst cc[CT];
  gb_consts<CT>(a, b, chunk0, cc);
  float m1[CT][8], m2[CT][8];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = chunk0 + 256 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = min(ch * 8 + e, a.C - 1) / a.cpg;
      m1[j][e] = mm[g][0];
      m2[j][e] = mm[g][1];
    }
  }
  const int per = (a.HW + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = min(a.HW, p0 + per);
  for (int pix = p0 + slot; pix < p1; pix += slots) {
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ch = chunk0 + 256 * j;
      if (ch < a.CP) {
        const size_t row = (size_t)b * a.HW + pix;
        const half8_t xv = gb_load(a, b, pix, ch * 8);
        const half8_t dv = *reinterpret_cast<const half8_t*>(a.dy + row * a.C + ch * 8);
        half8_t av = {0, 0, 0, 0, 0, 0, 0, 0};
        if (a.add) av = *reinterpret_cast<const half8_t*>(a.add + row * a.C + ch * 8);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (float)xv[e] * cc[j].ka[e] + cc[j].kb[e];
          float g = (float)dv[e] * cc[j].g[e];
          if (a.silu) g *= silu_grad(xh * cc[j].g[e] + cc[j].bt[e]);
          o[e] = (half_t)(cc[j].ka[e] * (g - m1[j][e] - xh * m2[j][e]) + (float)av[e]);
        }
        const int c0 = ch * 8;
        if (c0 < a.c1)
          *reinterpret_cast<half8_t*>(a.dx1 + row * a.c1 + c0) = o;
        else
          *reinterpret_cast<half8_t*>(a.dx2 + row * a.c2 + (c0 - a.c1)) = o;
      }
    }
  }
}

// LayerNorm input gradient, one wave per row
template <int CT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const half_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const half_t* __restrict__ dy, const half_t* __restrict__ add,
                                                            half_t* __restrict__ dx, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int CP = C >> 3;
  const size_t off = (size_t)row * C;
  half8_t xv[CT], dv[CT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    if (ch < CP) {
      xv[j] = *reinterpret_cast<const half8_t*>(x + off + ch * 8);
      dv[j] = *reinterpret_cast<const half8_t*>(dy + off + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)xv[j][e];
    }
  }
  const float mean = af_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j)
    if (lane + 64 * j < CP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)xv[j][e] - mean;
        q += d * d;
      }
    }
  const float rstd = rsqrtf(af_wave_sum(q) / (float)C + eps);
  float g[CT][8], xh[CT][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    if (ch < CP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[j][e] = ((float)xv[j][e] - mean) * rstd;
        g[j][e] = (float)dv[j][e] * gamma[ch * 8 + e];
        s1 += g[j][e];
        s2 += g[j][e] * xh[j][e];
      }
    }
  }
  const float m1 = af_wave_sum(s1) / (float)C, m2 = af_wave_sum(s2) / (float)C;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    if (ch < CP) {
      half8_t av = {0, 0, 0, 0, 0, 0, 0, 0};
      if (add) av = *reinterpret_cast<const half8_t*>(add + off + ch * 8);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)(rstd * (g[j][e] - m1 - xh[j][e] * m2) + (float)av[e]);
      *reinterpret_cast<half8_t*>(dx + off + ch * 8) = o;
    }
  }
}

// GEGLU on the interleaved pre-activation hp [M][2I]: 32-column groups [16 value | 16 gate]
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const half_t* __restrict__ hp, half_t* __restrict__ out, long M, int I) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per 8 output channels
  const int per_row = I >> 3;
  if (idx >= M * per_row) return;
  const long m = idx / per_row;
  const int c0 = (int)(idx - m * per_row) * 8;             // output channel of element 0
  const int col = (c0 >> 4) * 32 + (c0 & 15);             // its value column in hp
  const half8_t xv = *reinterpret_cast<const half8_t*>(hp + m * 2 * I + col);
  const half8_t gv = *reinterpret_cast<const half8_t*>(hp + m * 2 * I + col + 16);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)xv[e] * af_gelu_erf((float)gv[e]));
  *reinterpret_cast<half8_t*>(out + m * I + c0) = o;
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const half_t* __restrict__ hp, const half_t* __restrict__ dout,
                                                        half_t* __restrict__ dhp, long M, int I) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = I >> 3;
  if (idx >= M * per_row) return;
  const long m = idx / per_row;
  const int c0 = (int)(idx - m * per_row) * 8;
  const int col = (c0 >> 4) * 32 + (c0 & 15);
  const half8_t xv = *reinterpret_cast<const half8_t*>(hp + m * 2 * I + col);
  const half8_t gv = *reinterpret_cast<const half8_t*>(hp + m * 2 * I + col + 16);
  const half8_t dv = *reinterpret_cast<const half8_t*>(dout + m * I + c0);
  half8_t dxv, dgv;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g = (float)gv[e], d = (float)dv[e];
    const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
    dxv[e] = (half_t)(d * g * cdf);                        // d/dvalue = gelu(gate)
    dgv[e] = (half_t)(d * (float)xv[e] * (cdf + g * pdf)); // d/dgate  = value * gelu'(gate)
  }
  *reinterpret_cast<half8_t*>(dhp + m * 2 * I + col) = dxv;
  *reinterpret_cast<half8_t*>(dhp + m * 2 * I + col + 16) = dgv;
}

// adjoint of nearest x2 upsampling: y[b,h,w,:] = sum of the 2x2 block of x[b,2h..,2w..,:]
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int H,
                                                         int W, int C) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int cp = C >> 3;
  if (idx >= (long)B * H * W * cp) return;
  const int ch = (int)(idx % cp);
  const long pix = idx / cp;
  const int w = (int)(pix % W);
  const int h = (int)((pix / W) % H);
  const int b = (int)(pix / ((long)W * H));
  const half_t* base = x + (((size_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C + ch * 8;
  const half8_t v00 = *reinterpret_cast<const half8_t*>(base);
  const half8_t v01 = *reinterpret_cast<const half8_t*>(base + C);
  const half8_t v10 = *reinterpret_cast<const half8_t*>(base + (size_t)2 * W * C);
  const half8_t v11 = *reinterpret_cast<const half8_t*>(base + (size_t)2 * W * C + C);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v00[e] + (float)v01[e] + (float)v10[e] + (float)v11[e]);
  *reinterpret_cast<half8_t*>(y + pix * C + ch * 8) = o;
}

__global__ __launch_bounds__(256) void add_f16_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                      half_t* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const half8_t x = *reinterpret_cast<const half8_t*>(a + i * 8), y = *reinterpret_cast<const half8_t*>(b + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x[e] + (float)y[e]);
  *reinterpret_cast<half8_t*>(out + i * 8) = o;
}

__global__ __launch_bounds__(256) void axpy_f16_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, float alpha,
                                                       half_t* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const half8_t x = *reinterpret_cast<const half8_t*>(a + i * 8), y = *reinterpret_cast<const half8_t*>(b + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x[e] + alpha * (float)y[e]);
  *reinterpret_cast<half8_t*>(out + i * 8) = o;
}

// x [B][N][ldx] (C columns) -> y [B][C][ldy] (token index contiguous), 64x64 LDS tiles
__global__ __launch_bounds__(256) void transpose_tokens_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int N, int C,
                                                               int ldx, int ldy) {
  __shared__ half_t tile[64][66];
  const int b = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, c = c0 + tx;
    tile[r][tx] = (n < N && c < C) ? x[((size_t)b * N + n) * ldx + c] : (half_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, n = n0 + tx;
    if (c < C && n < ldy) y[((size_t)b * C + c) * ldy + n] = n < N ? tile[tx][r] : (half_t)0;
  }
}

// The same transpose with 16-byte global accesses on both sides (the scalar form above moves 2 bytes per lane: ~0.3 TB/s; the
// weight-gradient operands and the attention backward's Q^T / K^T / dO^T make ~400 such launches per training micro-batch).
// A 64-token x 64-channel tile: a lane loads 8 channels of one token (16 B), scatters them into the TRANSPOSED LDS tile
// [channel][token] (row stride 68 halves: 8-byte aligned rows, the 8 chunks of a wave land on 4 bank groups -> 2-way at worst),
// then reads 8 tokens of one channel (2 x ds_read_b64) and stores them as 16 B.  Needs C % 8 == 0, ldx % 8 == 0, ldy % 8 == 0 and
// 16-byte aligned bases; tokens >= N are zero-filled up to ldy like the scalar form.
constexpr int TT_ROW = 68;
__device__ __forceinline__ void transpose_tile_vec(const half_t* __restrict__ x, half_t* __restrict__ y, int N, int C, int ldx, int ldy,
                                                   int b, int n0, int c0, half_t* tile) {
  const int tid = threadIdx.x, ch = tid & 7, row = tid >> 3;           // load: token row (+32), channel chunk ch
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  half8_t v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + row + 32 * j, c = c0 + ch * 8;
    v[j] = (n < N && c < C) ? *reinterpret_cast<const half8_t*>(x + ((size_t)b * N + n) * ldx + c) : zero8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[(ch * 8 + e) * TT_ROW + row + 32 * j] = v[j][e];
  __syncthreads();
  // store: channel row (+32), token chunk ch
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + row + 32 * j, n = n0 + ch * 8;
    if (c < C && n < ldy) {
      const half4_t lo = *reinterpret_cast<const half4_t*>(tile + (row + 32 * j) * TT_ROW + ch * 8);
      const half4_t hi = *reinterpret_cast<const half4_t*>(tile + (row + 32 * j) * TT_ROW + ch * 8 + 4);
      const half8_t o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<half8_t*>(y + ((size_t)b * C + c) * ldy + n) = o;       // tokens >= N were loaded as zeros
    }
  }
}

__global__ __launch_bounds__(256) void transpose_tokens_vec_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int N, int C,
                                                                   int ldx, int ldy) {
  __shared__ __attribute__((aligned(16))) half_t tile[64 * TT_ROW];
  transpose_tile_vec(x, y, N, C, ldx, ldy, blockIdx.z, blockIdx.x * 64, blockIdx.y * 64, tile);
}

// up to three transposes of one batch in ONE launch (the attention backward's Q^T / dO^T / K^T, a weight gradient's two operands):
// blockIdx.x walks the tiles of the descriptors one after the other
struct TrDesc {
  const half_t* x;
  half_t* y;
  int N, C, ldx, ldy, tiles_n, tile_end;      // tile_end: first linear tile index past this descriptor
};
struct TrDescs {
  TrDesc d[3];
  int n;
};
__global__ __launch_bounds__(256) void transpose_tokens_multi_kernel(TrDescs ds) {
  __shared__ __attribute__((aligned(16))) half_t tile[64 * TT_ROW];
  int k = 0, first = 0;
  const int bid = blockIdx.x;
  if (ds.n > 1 && bid >= ds.d[0].tile_end) {
    k = 1;
    first = ds.d[0].tile_end;
    if (ds.n > 2 && bid >= ds.d[1].tile_end) {
      k = 2;
      first = ds.d[1].tile_end;
    }
  }
  const TrDesc d = ds.d[k];
  const int t = bid - first;
  const int tc = t / d.tiles_n, tn = t - tc * d.tiles_n;
  transpose_tile_vec(d.x, d.y, d.N, d.C, d.ldx, d.ldy, blockIdx.y, tn * 64, tc * 64, tile);
}

// ---- cautious AdamW (ldm/c_adamw.py:65-123) over a flat fp32 parameter buffer with per-tensor segments
struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  const long* seg_off;  // [nseg + 1]
  unsigned int* counts;  // [nseg] number of elements with m*g > 0
  float lr, beta1, beta2, eps, wd, step_size;
};

__global__ __launch_bounds__(256) void cadamw_moments_kernel(AdamArgs a) {
  const int seg = blockIdx.y;
  const long lo = a.seg_off[seg], hi = a.seg_off[seg + 1];
  unsigned int cnt = 0;
  for (long i = lo + (long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long)gridDim.x * 256) {
    const float g = a.g[i];
    const float m = a.m[i] * a.beta1 + g * (1.0f - a.beta1);
    const float v = a.v[i] * a.beta2 + g * g * (1.0f - a.beta2);
    a.m[i] = m;
    a.v[i] = v;
    cnt += (m * g > 0.f) ? 1u : 0u;
  }
  // integer count: deterministic regardless of arrival order
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(a.counts + seg, cnt);
}

__global__ __launch_bounds__(256) void cadamw_update_kernel(AdamArgs a) {
  const int seg = blockIdx.y;
  const long lo = a.seg_off[seg], hi = a.seg_off[seg + 1];
  const float mean = fmaxf((float)a.counts[seg] / (float)(hi - lo), 1e-3f);  // mask.mean().clamp_(min=1e-3)
  const float inv_mean = 1.0f / mean;
  for (long i = lo + (long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long)gridDim.x * 256) {
    float p = a.p[i];
    if (a.wd > 0.f) p += p * (-a.lr * a.wd);
    const float g = a.g[i], m = a.m[i];
    const float mask = (m * g > 0.f) ? inv_mean : 0.f;
    const float denom = sqrtf(a.v[i]) + a.eps;
    a.p[i] = p - a.step_size * (m * mask) / denom;
  }
}

// out[c] (+)= sum_r a[r,c] * b[r,c]; one block per 64 columns, 4 waves stride over rows, LDS fold
__global__ __launch_bounds__(256) void colsum_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                     float* __restrict__ out, int rows, int C, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C) {
    for (int r = w; r < rows; r += 4) {
      const float av = (float)a[(size_t)r * C + c];
      s += b ? av * (float)b[(size_t)r * C + c] : av;
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// tall matrices (rows >> columns: per-pixel gradients of the DoRA magnitudes): grid (C / 64, chunks), each workgroup folds its row
// range into partial[chunk][c]; colsum_reduce_kernel folds the chunks in a fixed order (deterministic, no atomics)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                             float* __restrict__ partial, int rows, int C, int rows_per_chunk) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < C) {
    for (int r = r0 + w; r < r1; r += 4) {
      const float av = (float)a[(size_t)r * C + c];
      s += b ? av * (float)b[(size_t)r * C + c] : av;
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// The same two reductions with 16-byte loads (C % 8 == 0, 16-byte aligned bases): a workgroup owns 64 columns = 8 chunks; its 256
// threads are 8 chunk lanes x 32 row lanes, every thread keeps four rows in flight, the 32 row lanes meet in LDS in a fixed order.
// The scalar forms above read 2 bytes per lane with ONE load in flight per thread: ~10 us for a [388, 768] bias gradient.
__device__ __forceinline__ void colsum_vec_body(const half_t* __restrict__ a, const half_t* __restrict__ b, int r0, int r1, int C, int c0,
                                                float (&s)[8]) {
  const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = c0 + ch * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (c >= C) return;
  for (int r = r0 + rl; r < r1; r += 128) {
    half8_t av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 32 * u;
      const bool ok = rr < r1;
      av[u] = ok ? *reinterpret_cast<const half8_t*>(a + (size_t)rr * C + c) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      if (b) bv[u] = ok ? *reinterpret_cast<const half8_t*>(b + (size_t)rr * C + c) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += b ? (float)av[u][e] * (float)bv[u][e] : (float)av[u][e];
  }
}

__global__ __launch_bounds__(256) void colsum_vec_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, float* __restrict__ out,
                                                         int rows, int C, int rows_per_chunk, int direct, int accumulate) {
  // direct: out = [C] totals (one chunk); otherwise out = partial[chunk][C]
  __shared__ float red[32][65];
  const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s[8];
  colsum_vec_body(a, b, r0, r1, C, c0, s);
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][ch * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    const int c = c0 + threadIdx.x;
    if (direct) out[c] = accumulate ? out[c] + t : t;
    else out[(size_t)blockIdx.y * C + c] = t;
  }
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int chunks, int C,
                                                            int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += partial[(size_t)k * C + c];
  out[c] = accumulate ? out[c] + s : s;
}

__global__ __launch_bounds__(256) void quickgelu_fwd_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const half8_t v = *reinterpret_cast<const half8_t*>(x + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = (float)v[e];
    o[e] = (half_t)(f * af_sigmoid(1.702f * f));
  }
  *reinterpret_cast<half8_t*>(y + i * 8) = o;
}

__global__ __launch_bounds__(256) void quickgelu_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                            half_t* __restrict__ dx, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const half8_t v = *reinterpret_cast<const half8_t*>(x + i * 8), g = *reinterpret_cast<const half8_t*>(dy + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = (float)v[e];
    const float sg = af_sigmoid(1.702f * f);
    o[e] = (half_t)((float)g[e] * sg * (1.0f + 1.702f * f * (1.0f - sg)));
  }
  *reinterpret_cast<half8_t*>(dx + i * 8) = o;
}

__global__ __launch_bounds__(256) void clamp_f32_kernel(float* __restrict__ a, float lo, float hi, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = fminf(fmaxf(a[i], lo), hi);       // NaN stays NaN (the overflow check runs before the clip)
}

__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ a, float s, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] *= s;
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }


// LayerNorm parameter gradients: dgamma[c] = sum_r dy[r,c] * xhat[r,c], dbeta[c] = sum_r dy[r,c], xhat = (x - mean_r) * rstd_r.
// Two small launches: (1) a wave per row writes (mean, rstd) -- exact two-pass variance on the row held in registers, as in
// layernorm_kernel; (2) a workgroup per 64 columns folds the rows with 16-byte loads, four rows in flight per thread (8 chunk lanes x
// 32 row lanes, meeting in LDS in a fixed order).  Replaces two fills, a LayerNorm pass for xhat and two column-sum launches per
// trainable LayerNorm.  (One launch with every workgroup recomputing all row statistics was tried first: 47 us for [388, 768] --
// 97 dependent row steps per wave -- against ~25 us for the five launches it replaced.)
constexpr int LNPG_MAX_ROWS = 2048;
template <int CT>
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const half_t* __restrict__ x, float* __restrict__ st, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int nch = C >> 3;
  const half_t* xr = x + (size_t)r * C;
  half8_t v[CT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int ch = lane + 64 * j;
    v[j] = ch < nch ? *reinterpret_cast<const half8_t*>(xr + ch * 8) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)v[j][e];
  }
  const float mean = af_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    if (lane + 64 * j < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = (float)v[j][e] - mean;
        q += dlt * dlt;
      }
    }
  }
  const float var = af_wave_sum(q) / (float)C;
  if (lane == 0) {
    st[2 * r] = mean;
    st[2 * r + 1] = rsqrtf(var + eps);
  }
}

__global__ __launch_bounds__(256) void ln_param_grads_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                             const float* __restrict__ st, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int rows, int C) {
  __shared__ float red[2 * 32 * 65];
  const int tid = threadIdx.x;
  const int ch = tid & 7, rl = tid >> 3;
  const int c = blockIdx.x * 64 + ch * 8;
  float sg[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sg[e] = sb[e] = 0.f;
  if (c < C) {
    for (int r = rl; r < rows; r += 128) {
      half8_t xv[4], dv[4];
      float mu[4], rs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 32 * u;
        const bool ok = rr < rows;
        const int rc = ok ? rr : 0;
        xv[u] = *reinterpret_cast<const half8_t*>(x + (size_t)rc * C + c);
        dv[u] = ok ? *reinterpret_cast<const half8_t*>(dy + (size_t)rc * C + c) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        mu[u] = st[2 * rc];
        rs[u] = st[2 * rc + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)dv[u][e];
          sg[e] += d * ((float)xv[u][e] - mu[u]) * rs[u];
          sb[e] += d;
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[rl * 65 + ch * 8 + e] = sg[e];
    red[32 * 65 + rl * 65 + ch * 8 + e] = sb[e];
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, col = tid & 63;
    const int cc = blockIdx.x * 64 + col;
    if (cc < C) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += red[which * 32 * 65 + k * 65 + col];
      (which ? dbeta : dgamma)[cc] = t;
    }
  }
}

}  // namespace


// [B, N, C (ldx)] -> [B, C, ldy] with the token index contiguous (tokens N .. ldy zero-filled); also used by af_attn_bwd.hip
static bool transpose_vec_ok(const half_t* x, const half_t* y, int C, int ldx, int ldy) {
  return C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}
void af_launch_transpose_tokens(const half_t* x, half_t* y, int B, int N, int C, int ldx, int ldy, hipStream_t stream) {
  dim3 grid((ldy + 63) / 64, (C + 63) / 64, B);
  if (transpose_vec_ok(x, y, C, ldx, ldy)) hipLaunchKernelGGL(transpose_tokens_vec_kernel, grid, dim3(256), 0, stream, x, y, N, C, ldx, ldy);
  else hipLaunchKernelGGL(transpose_tokens_kernel, grid, dim3(256), 0, stream, x, y, N, C, ldx, ldy);
}
// n <= 3 transposes of the same batch count: one launch when every one qualifies for the 16-byte form, separate launches otherwise
void af_launch_transpose_tokens_multi(const AfTransposeJob* jobs, int n, int B, hipStream_t stream) {
  bool vec = n >= 1 && n <= 3;
  for (int i = 0; vec && i < n; ++i) vec = transpose_vec_ok(jobs[i].x, jobs[i].y, jobs[i].C, jobs[i].ldx, jobs[i].ldy);
  if (!vec || n == 1) {
    for (int i = 0; i < n; ++i) af_launch_transpose_tokens(jobs[i].x, jobs[i].y, B, jobs[i].N, jobs[i].C, jobs[i].ldx, jobs[i].ldy, stream);
    return;
  }
  TrDescs ds;
  ds.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    TrDesc& d = ds.d[i];
    d.x = jobs[i].x;
    d.y = jobs[i].y;
    d.N = jobs[i].N;
    d.C = jobs[i].C;
    d.ldx = jobs[i].ldx;
    d.ldy = jobs[i].ldy;
    d.tiles_n = (jobs[i].ldy + 63) / 64;
    total += d.tiles_n * ((jobs[i].C + 63) / 64);
    d.tile_end = total;
  }
  hipLaunchKernelGGL(transpose_tokens_multi_kernel, dim3(total, B), dim3(256), 0, stream, ds);
}

extern "C" int af_groupnorm_bwd(const void* x1, const void* x2, int c1, int c2, const void* gamma, const void* beta,
                                const void* stats, const void* dy, const void* add, void* dx1, void* dx2, int B, int HW,
                                int groups, int silu, void* workspace, void* stream) {
  AF_REQUIRE(x1 && gamma && beta && stats && dy && dx1 && workspace, "af_groupnorm_bwd: null pointer");
  AF_REQUIRE(B > 0 && HW > 0 && c1 > 0 && c2 >= 0 && c1 % 8 == 0 && c2 % 8 == 0, "af_groupnorm_bwd: bad sizes");
  AF_REQUIRE(c2 == 0 || (x2 && dx2), "af_groupnorm_bwd: x2/dx2 required when c2 > 0");
  const int C = c1 + c2;
  AF_REQUIRE(groups > 0 && groups <= GB_MAXG && C % groups == 0, "af_groupnorm_bwd: groups must divide C and be <= 32");
  AF_SUPPORTED(C <= 4096, "af_groupnorm_bwd: C > 4096");
  GnBwdArgs a;
  a.x1 = (const half_t*)x1;
  a.x2 = (const half_t*)x2;
  a.c1 = c1;
  a.c2 = c2;
  a.C = C;
  a.CP = C / 8;
  a.gamma = (const float*)gamma;
  a.beta = (const float*)beta;
  a.stats = (const float*)stats;
  a.dy = (const half_t*)dy;
  a.add = (const half_t*)add;
  a.dx1 = (half_t*)dx1;
  a.dx2 = (half_t*)dx2;
  a.B = B;
  a.HW = HW;
  a.groups = groups;
  a.cpg = C / groups;
  a.silu = silu;
  a.ws = (float*)workspace;
  const int ct = (a.CP + 255) / 256;
  a.ppb = ct == 1 ? 256 / a.CP : 1;
  const int slots = ct == 1 ? a.ppb : 1;
  const size_t lds = (size_t)2 * slots * C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_GNORM, stream);
  dim3 gp(GB_NBLK, B), blk(256);
  int nb2 = (HW + slots - 1) / slots;
  nb2 = nb2 > 64 ? 64 : nb2;
  dim3 ga(nb2, B);
  if (ct == 1) {
    hipLaunchKernelGGL(gn_bwd_partial_kernel<1>, gp, blk, lds, s, a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, ga, blk, 0, s, a);
  } else {
    hipLaunchKernelGGL(gn_bwd_partial_kernel<2>, gp, blk, lds, s, a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<2>, ga, blk, 0, s, a);
  }
  return af_check_launch("af_groupnorm_bwd");
}

extern "C" int af_layernorm_bwd(const void* x, const void* gamma, const void* dy, const void* add, void* dx, int rows, int C,
                                float eps, void* stream) {
  AF_REQUIRE(x && gamma && dy && dx, "af_layernorm_bwd: null pointer");
  AF_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "af_layernorm_bwd: C must be a positive multiple of 8");
  AF_SUPPORTED(C <= 1536, "af_layernorm_bwd: C > 1536");
  const int ct = (C / 8 + 63) / 64;
  dim3 grid((rows + 3) / 4), blk(256);
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_LNORM, stream);
  const half_t* xx = (const half_t*)x;
  const float* g = (const float*)gamma;
  const half_t* d = (const half_t*)dy;
  const half_t* ad = (const half_t*)add;
  half_t* o = (half_t*)dx;
  switch (ct) {
    case 1: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, blk, 0, s, xx, g, d, ad, o, rows, C, eps); break;
    case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, blk, 0, s, xx, g, d, ad, o, rows, C, eps); break;
    default: hipLaunchKernelGGL(layernorm_bwd_kernel<3>, grid, blk, 0, s, xx, g, d, ad, o, rows, C, eps); break;
  }
  return af_check_launch("af_layernorm_bwd");
}

extern "C" int af_geglu_fwd(const void* hp, void* out, int64_t M, int inner, void* stream) {
  AF_REQUIRE(hp && out && M > 0 && inner > 0 && inner % 16 == 0, "af_geglu_fwd: inner must be a positive multiple of 16");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(geglu_fwd_kernel, g1((long)M * (inner / 8)), dim3(256), 0, (hipStream_t)stream, (const half_t*)hp,
                     (half_t*)out, (long)M, inner);
  return af_check_launch("af_geglu_fwd");
}

extern "C" int af_geglu_bwd(const void* hp, const void* dout, void* dhp, int64_t M, int inner, void* stream) {
  AF_REQUIRE(hp && dout && dhp && M > 0 && inner > 0 && inner % 16 == 0, "af_geglu_bwd: inner must be a positive multiple of 16");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(geglu_bwd_kernel, g1((long)M * (inner / 8)), dim3(256), 0, (hipStream_t)stream, (const half_t*)hp,
                     (const half_t*)dout, (half_t*)dhp, (long)M, inner);
  return af_check_launch("af_geglu_bwd");
}

extern "C" int af_sumpool2x2(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  AF_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "af_sumpool2x2: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(sumpool2x2_kernel, g1((long)B * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (half_t*)y, B, H, W, C);
  return af_check_launch("af_sumpool2x2");
}

extern "C" int af_add_f16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  AF_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "af_add_f16: n must be a positive multiple of 8");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(add_f16_kernel, g1(n / 8), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b,
                     (half_t*)out, (long)(n / 8));
  return af_check_launch("af_add_f16");
}

extern "C" int af_axpy_f16(const void* a, const void* b, float alpha, void* out, int64_t n, void* stream) {
  AF_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "af_axpy_f16: n must be a positive multiple of 8");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(axpy_f16_kernel, g1(n / 8), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, alpha,
                     (half_t*)out, (long)(n / 8));
  return af_check_launch("af_axpy_f16");
}

extern "C" int af_transpose_tokens(const void* x, void* y, int B, int N, int C, int ldx, int ldy, void* stream) {
  AF_REQUIRE(x && y && B > 0 && N > 0 && C > 0 && ldx >= C && ldy >= N, "af_transpose_tokens: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  af_launch_transpose_tokens((const half_t*)x, (half_t*)y, B, N, C, ldx, ldy, (hipStream_t)stream);
  return af_check_launch("af_transpose_tokens");
}

extern "C" int af_cadamw_step(void* p, const void* g, void* m, void* v, const void* seg_offsets, int nseg, void* counts,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step, int correct_bias,
                              void* stream) {
  AF_REQUIRE(p && g && m && v && seg_offsets && counts && nseg > 0 && step > 0, "af_cadamw_step: bad argument");
  AF_REQUIRE(nseg <= 65535, "af_cadamw_step: too many segments");
  AdamArgs a;
  a.p = (float*)p;
  a.g = (const float*)g;
  a.m = (float*)m;
  a.v = (float*)v;
  a.seg_off = (const long*)seg_offsets;
  a.counts = (unsigned int*)counts;
  a.lr = lr;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.wd = weight_decay;
  double ss = lr;
  if (correct_bias) ss = ss * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
  a.step_size = (float)ss;
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(unsigned int) * nseg, s);
  if (e != hipSuccess) return af_fail(AF_E_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
  dim3 grid(64, nseg), blk(256);
  hipLaunchKernelGGL(cadamw_moments_kernel, grid, blk, 0, s, a);
  hipLaunchKernelGGL(cadamw_update_kernel, grid, blk, 0, s, a);
  return af_check_launch("af_cadamw_step");
}

extern "C" int af_colsum(const void* a, const void* b, void* out, int rows, int C, int accumulate, void* stream) {
  AF_REQUIRE(a && out && rows > 0 && C > 0, "af_colsum: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const bool vec = C % 8 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(colsum_vec_kernel, dim3((C + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b,
                       (float*)out, rows, C, rows, 1, accumulate);
  else
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b,
                       (float*)out, rows, C, accumulate);
  return af_check_launch("af_colsum");
}

extern "C" int af_colsum_tall(const void* a, const void* b, void* partial, void* out, int rows, int C, int chunks, int accumulate,
                              void* stream) {
  AF_REQUIRE(a && partial && out && rows > 0 && C > 0 && chunks > 0 && chunks <= 65535, "af_colsum_tall: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const int rpc = (rows + chunks - 1) / chunks;
  const bool vec = C % 8 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(colsum_vec_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (const half_t*)a,
                       (const half_t*)b, (float*)partial, rows, C, rpc, 0, 0);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (const half_t*)a,
                       (const half_t*)b, (float*)partial, rows, C, rpc);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)partial, (float*)out,
                     chunks, C, accumulate);
  return af_check_launch("af_colsum_tall");
}

extern "C" int af_quickgelu_fwd(const void* x, void* y, int64_t n, void* stream) {
  AF_REQUIRE(x && y && n > 0 && n % 8 == 0, "af_quickgelu_fwd: n must be a positive multiple of 8");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(quickgelu_fwd_kernel, g1(n / 8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, (long)(n / 8));
  return af_check_launch("af_quickgelu_fwd");
}

extern "C" int af_quickgelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  AF_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, "af_quickgelu_bwd: n must be a positive multiple of 8");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(quickgelu_bwd_kernel, g1(n / 8), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)dy,
                     (half_t*)dx, (long)(n / 8));
  return af_check_launch("af_quickgelu_bwd");
}

extern "C" int af_clamp_f32(void* a, float lo, float hi, int64_t n, void* stream) {
  AF_REQUIRE(a && n > 0 && lo <= hi, "af_clamp_f32: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(clamp_f32_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, (float*)a, lo, hi, (long)n);
  return af_check_launch("af_clamp_f32");
}

extern "C" int af_scale_f32(void* a, float s, int64_t n, void* stream) {
  AF_REQUIRE(a && n > 0, "af_scale_f32: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  hipLaunchKernelGGL(scale_f32_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, (float*)a, s, (long)n);
  return af_check_launch("af_scale_f32");
}

extern "C" int af_layernorm_param_grads(const void* x, const void* dy, void* dgamma, void* dbeta, void* row_stats, int rows, int C, float eps,
                                        void* stream) {
  AF_REQUIRE(x && dy && dgamma && dbeta && row_stats, "af_layernorm_param_grads: null pointer");
  AF_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "af_layernorm_param_grads: C must be a positive multiple of 8");
  AF_SUPPORTED(rows <= LNPG_MAX_ROWS && C <= 1536, "af_layernorm_param_grads: more than 2048 rows or C > 1536 (use the column-sum path)");
  AfLaunchScope scope(AF_FAM_LNORM, stream);
  hipStream_t s = (hipStream_t)stream;
  const half_t* xx = (const half_t*)x;
  float* st = (float*)row_stats;
  const int ct = (C / 8 + 63) / 64;
  dim3 g1((rows + 3) / 4), blk(256);
  if (ct == 1) hipLaunchKernelGGL(ln_row_stats_kernel<1>, g1, blk, 0, s, xx, st, rows, C, eps);
  else if (ct == 2) hipLaunchKernelGGL(ln_row_stats_kernel<2>, g1, blk, 0, s, xx, st, rows, C, eps);
  else hipLaunchKernelGGL(ln_row_stats_kernel<3>, g1, blk, 0, s, xx, st, rows, C, eps);
  hipLaunchKernelGGL(ln_param_grads_kernel, dim3((C + 63) / 64), blk, 0, s, xx, (const half_t*)dy, (const float*)st, (float*)dgamma,
                     (float*)dbeta, rows, C);
  return af_check_launch("af_layernorm_param_grads");
}

extern "C" int af_transpose_tokens_pair(const void* x1, void* y1, int C1, int ldx1, const void* x2, void* y2, int C2, int ldx2, int B, int N,
                                        int ldy, void* stream) {
  AF_REQUIRE(x1 && y1 && x2 && y2 && B > 0 && N > 0 && C1 > 0 && C2 > 0 && ldx1 >= C1 && ldx2 >= C2 && ldy >= N,
             "af_transpose_tokens_pair: bad argument");
  AfLaunchScope scope(AF_FAM_ELEM, stream);
  const AfTransposeJob jobs[2] = {{(const half_t*)x1, (half_t*)y1, N, C1, ldx1, ldy}, {(const half_t*)x2, (half_t*)y2, N, C2, ldx2, ldy}};
  af_launch_transpose_tokens_multi(jobs, 2, B, (hipStream_t)stream);
  return af_check_launch("af_transpose_tokens_pair");
}
