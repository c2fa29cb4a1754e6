// af_gemm.hip -- one MFMA kernel template for every matmul-shaped op of the U-Net:
// 3x3 convolution as implicit GEMM over NHWC (stride 1/2, fused nearest-x2 upsample, fused
// channel concat of two sources), 1x1 convolution and nn.Linear.
//
//   out[m, n] = epilogue( sum_k A[m, k] * Wt[n, k] )
//
// Layout / mapping (gfx950, wave64):
//   * both operands are K-contiguous, so A (activations) and Wt (weights) fragments are
//     16-byte rows for v_mfma_f32_16x16x32_f16;
//   * the MFMA is issued "swapped" (Wt as the A operand, activations as the B operand) so the
//     accumulator holds 4 CONSECUTIVE output channels per lane -> 8-byte epilogue stores and
//     8-byte bias / residual loads along the contiguous dimension of the NHWC output;
//   * block tile BM x BN x 64, 4 waves as 2(M) x 2(N); LDS tiles are [rows][64] fp16 with the
//     16-byte chunk index XOR-swizzled by (row & 7) so ds_read_b128 fragment reads are
//     conflict-free; global->register->LDS staging, double-buffered, one barrier per K step
//     (the next tile's global loads are issued before the MFMAs of the current one);
//   * workgroup ids are remapped so each XCD (private L2) walks a contiguous range of tiles.
#include <stdlib.h>

#include "af_common.h"
#include <stdlib.h>

// Which operand an XCD's contiguous tile range should share in its private 4 MB L2: every XCD streams the operand it does NOT
// share in full, so share the bigger one.  Weights (N x K) beat activations (M x Cin) on the 16x16 / 8x8 levels, where the
// default M-major mapping made all 8 XCDs fetch the whole 29.5 MB filter of a 1280 -> 1280 conv (FETCH_SIZE, profiles/r01g).
int af_gemm_n_major(int M, int N, int K, int cin) {
  static const int force = getenv("AF_GEMM_NMAJOR") ? atoi(getenv("AF_GEMM_NMAJOR")) : -1;
  if (force >= 0) return force;
  return (long)N * K > (long)M * cin ? 1 : 0;
}

namespace {

struct GemmDev {
  const half_t* a1;
  const half_t* a2;
  const half_t* wt;
  const float* bias;
  const half_t* rowbias;
  const half_t* residual;
  half_t* out;
  half_t* out2;
  int M, N, K, kpad;
  int c1, c2, lda1, lda2;
  int H, W, Ho, Wo, HoWo, Heff, Weff, stride, upsample;
  int rows_per_batch, ld_rowbias, act_silu, ld_out, split_col, ld_out2;  // act_silu: 0 none, 1 SiLU, 3 quick-GELU
  int tiles_n, tiles_m, n_major;
  int tap_shift;   // 0: padding 1 all round; 1: taps shifted by +1 (padding (0,1,0,1))
  int splits, kt_per_split;  // split-K: blockIdx.y owns K steps [y*kt_per_split, ...)
  float* ws;                 // fp32 partials [splits][M][N] when splits > 1
  int* counters;             // in-kernel split-K reduction: per-tile arrival counters (zero between launches); nullptr = reduce pass
  int out_f32;               // AF_OUT_F32: the reduce pass stores fp32
  int wpf, wpf_coop;         // weight-tile L2 prefetch at kernel start (af_common.h): instructions per wave (0 = off), sharing workgroups
};

constexpr int BK = 64;
enum { EPI_STD = 0, EPI_GEGLU = 1, EPI_SPLIT_T = 2 };

// af_gemm_desc.defer_reduce of the call in progress (host side, per thread): where a launcher would run the separate reduce pass it stores the
// slab count there instead
thread_local int32_t* g_defer_reduce = nullptr;

// FAST (3x3 only): both channel counts are multiples of 64 and there is no upsampling, so one 64-wide K step lies
// inside ONE tap and ONE source for the whole workgroup: tap, channel offset and source are scalar (SGPR) values,
// and a thread only needs a precomputed per-row centre-pixel offset and a 9-bit mask of in-bounds taps.  This
// replaces ~170 VALU/SALU instructions of im2col address generation per K step (more than the 32 MFMAs cost)
// with ~4 VALU per row.
template <int BM, int BN, int TAPS, int EPI, bool FAST = false>
__global__ __launch_bounds__(256, 2) void af_gemm_kernel(GemmDev p) {
  constexpr int WM = BM / 2, WN = BN / 2;    // per-wave tile
  constexpr int TM = WM / 16, TN = WN / 16;  // 16x16 MFMA tiles per wave
  constexpr int AI = BM / 32, WI = BN / 32;  // 16-byte chunks per thread per K step
  constexpr int STAGE = (BM + BN) * BK;      // halves per LDS buffer
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  half_t* lds = reinterpret_cast<half_t*>(af_smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware, bijective tile remap (blocks b and b+8 share an XCD / L2).
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.n_major) {           // weights outweigh activations: an XCD's contiguous tile range shares W columns, not A rows
      tile_n = lid / p.tiles_m;
      tile_m = lid - tile_n * p.tiles_m;
    } else {
      tile_m = lid / p.tiles_n;
      tile_n = lid - tile_m * p.tiles_n;
    }
  }

  // ---- loader state: thread owns chunk column cc (8 halves) of rows rb + 32*i
  const int cc = tid & 7, rb = tid >> 3;
  const int Cin = p.c1 + p.c2;
  int a_iy0[AI], a_ix0[AI], a_base[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = tile_m * BM + rb + 32 * i;
    if (TAPS == 9) {
      if (m < p.M) {
        const int b = m / p.HoWo;
        const int rem = m - b * p.HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_iy0[i] = oy * p.stride - 1 + p.tap_shift;
        a_ix0[i] = ox * p.stride - 1 + p.tap_shift;
        a_base[i] = b * p.H * p.W;
      } else {
        a_iy0[i] = -(1 << 20);
        a_ix0[i] = 0;
        a_base[i] = 0;
      }
    } else {
      a_base[i] = (m < p.M) ? m : -1;
      a_iy0[i] = a_ix0[i] = 0;
    }
  }
  // FAST loader state: a_base[i] = linear index of the centre input pixel (or -1), a_mask[i] = in-bounds taps
  unsigned a_mask[AI];
  if (FAST) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      unsigned mk = 0;
      const int cy = a_iy0[i] + 1, cx = a_ix0[i] + 1;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int iy = cy + t9 / 3 - 1, ix = cx + t9 % 3 - 1;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t9;
      }
      a_mask[i] = a_iy0[i] < -1000 ? 0u : mk;
      a_base[i] = a_base[i] + cy * p.W + cx;
    }
  }
  const int nk_total = p.kpad / BK;
  const int kt_begin = blockIdx.y * p.kt_per_split;
  const int kt_end = min(nk_total, kt_begin + p.kt_per_split);
  int tap = 0, c = kt_begin * BK + cc * 8;  // TAPS==9: (tap, channel) of this thread's chunk in the current K step
  if (TAPS == 9) {
    tap = c / Cin;
    c -= tap * Cin;
  }

  half8_t ra[AI], rw[WI];
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tile = [&](int kt) {
    const half_t* wp = p.wt + (size_t)(tile_n * BN + rb) * p.kpad + kt * BK + cc * 8;
#pragma unroll
    for (int i = 0; i < WI; ++i) rw[i] = *reinterpret_cast<const half8_t*>(wp + (size_t)i * 32 * p.kpad);
    if (TAPS == 9 && FAST) {
      // workgroup-uniform: tap and channel offset of this K step (k = kt*64 is a multiple of 64 | Cin)
      const int k0 = kt * BK;
      const int tp = k0 / Cin;
      const int c0 = k0 - tp * Cin;
      const bool first = c0 < p.c1;
      const half_t* src = first ? p.a1 : p.a2;
      const int cs = first ? p.c1 : p.c2;
      const int coff = (first ? c0 : c0 - p.c1) + cc * 8;
      const int dpix = (tp / 3 - 1) * p.W + (tp % 3 - 1);
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool ok = (a_mask[i] >> tp) & 1u;   // tp >= 9 (K padding) -> no bit set
        ra[i] = ok ? *reinterpret_cast<const half8_t*>(src + (size_t)(a_base[i] + dpix) * cs + coff) : zero8;
      }
    } else if (TAPS == 9) {
      const bool kval = tap < 9;
      const int ky = tap / 3, kx = tap - ky * 3;
      const half_t* src;
      int cs, coff;
      if (c < p.c1) {
        src = p.a1;
        cs = p.c1;
        coff = c;
      } else {
        src = p.a2;
        cs = p.c2;
        coff = c - p.c1;
      }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        bool ok = kval && (unsigned)iy < (unsigned)p.Heff && (unsigned)ix < (unsigned)p.Weff;
        if (p.upsample == 2) ok = ok && !((iy | ix) & 1);  // zero-insertion: only even positions carry data
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        const size_t off = ((size_t)(a_base[i] + sy * p.W + sx)) * cs + coff;
        ra[i] = ok ? *reinterpret_cast<const half8_t*>(src + off) : zero8;
      }
      c += BK;
      while (c >= Cin) {
        c -= Cin;
        ++tap;
      }
    } else {
      const int k = kt * BK + cc * 8;
      const bool kval = k < p.K;
      const half_t* src;
      int ld, koff;
      if (k < p.c1) {
        src = p.a1;
        ld = p.lda1;
        koff = k;
      } else {
        src = p.a2;
        ld = p.lda2;
        koff = k - p.c1;
      }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool ok = kval && a_base[i] >= 0;
        ra[i] = ok ? *reinterpret_cast<const half8_t*>(src + (size_t)a_base[i] * ld + koff) : zero8;
      }
    }
  };

  const int st_off = ((cc ^ (rb & 7)) * 8);  // swizzled chunk, identical for every row this thread writes
  auto store_tile = [&](int buf) {
    half_t* As = lds + buf * STAGE;
    half_t* Ws = As + BM * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<half8_t*>(As + (rb + 32 * i) * BK + st_off) = ra[i];
#pragma unroll
    for (int i = 0; i < WI; ++i) *reinterpret_cast<half8_t*>(Ws + (rb + 32 * i) * BK + st_off) = rw[i];
  };

  floatx4 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const half_t* As = lds + buf * STAGE;
    const half_t* Ws = As + BM * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int sw = ((ks * 4 + fq) ^ (fr & 7)) * 8;
      half8_t wf[TN], xf[TM];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        wf[tn] = *reinterpret_cast<const half8_t*>(Ws + (wn * WN + tn * 16 + fr) * BK + sw);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        xf[tm] = *reinterpret_cast<const half8_t*>(As + (wm * WM + tm * 16 + fr) * BK + sw);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
    }
  };

  // (the 64 x 64 tile only: the 128 x 128 tile's staging buffers fill the 64 KB a launch gets without opting in, and the dump area must not
  // alias them -- another wave's ds_write may precede this wave's late-landing prefetch)
  if (p.wpf > 0 && BM == 64)
    af_prefetch_weight_tile(p.wt, p.kpad, (p.N + 127) / 128 * 128, tile_n * BN, BN, kt_begin, kt_end - kt_begin, p.wpf_coop, tile_m % p.wpf_coop,
                            p.wpf, 4, tid >> 6, tid & 63, af_smem + (size_t)2 * (BM + BN) * BK * sizeof(half_t));
  load_tile(kt_begin);
  store_tile(0);
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    const int buf = (kt - kt_begin) & 1;
    if (more) load_tile(kt + 1);
    compute(buf);
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- split-K: raw fp32 partial tile to the workspace; af_splitk_reduce applies the epilogue
  if (EPI == EPI_STD && p.splits > 1) {
    float* wsp = p.ws + (size_t)blockIdx.y * p.M * p.N;
    if (p.counters != nullptr) {
      // slabs handed to another workgroup inside this launch: WRITE-THROUGH (sc1) 16-byte stores, so that no release fence (an L2
      // write-back of every dirty line, ~20 us with every workgroup's slab dirty) is needed before the arrival counter
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(wsp, 0, p.M * p.N * 4, 0x00020000);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = tile_m * BM + wm * WM + tm * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int n0 = tile_n * BN + wn * WN + tn * 16 + 4 * fq;
          if (n0 < p.N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, acc[tn][tm]), rsrc, (m * p.N + n0) * 4, 0, 16);
        }
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = tile_m * BM + wm * WM + tm * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int n0 = tile_n * BN + wn * WN + tn * 16 + 4 * fq;
          if (n0 < p.N) *reinterpret_cast<floatx4*>(wsp + (size_t)m * p.N + n0) = acc[tn][tm];
        }
      }
    }
    if (p.counters == nullptr) return;               // af_splitk_reduce_kernel follows
    // in-kernel reduction by the last-arriving K-slice of this tile: same protocol and same summation order as af_gemm3.hip's
    // gemm3_epilogue (agent-scope release / acquire around a relaxed arrival counter; slabs summed in slice order)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* last_flag = reinterpret_cast<int*>(af_smem);
    if (threadIdx.x == 0) {
      int* cnt = p.counters + tile_m * p.tiles_n + tile_n;
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.splits - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *last_flag = last;
    }
    __syncthreads();
    if (!*last_flag) return;
    // The slabs were written through to memory, so every load is a full round trip: issue them in big batches.  Slice 0 lands
    // directly in the (now dead) accumulators, all TN x TM fragments in flight at once; every further slice is added from a
    // half-tile of temporaries.  Summation order = slice order, as in af_splitk_reduce_kernel (0 + s0 == s0 exactly).
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = tile_m * BM + wm * WM + tm * 16 + fr;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n0 = tile_n * BN + wn * WN + tn * 16 + 4 * fq;
        acc[tn][tm] = (m < p.M && n0 < p.N) ? *reinterpret_cast<const floatx4*>(p.ws + (size_t)m * p.N + n0) : floatx4{0.f, 0.f, 0.f, 0.f};
      }
    }
    for (int sp = 1; sp < p.splits; ++sp) {
      const float* slab = p.ws + (size_t)sp * p.M * p.N;
#pragma unroll
      for (int th = 0; th < TM; th += 2) {
        floatx4 part[TN][2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int tm = th + t2;
          const int m = tile_m * BM + wm * WM + tm * 16 + fr;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const int n0 = tile_n * BN + wn * WN + tn * 16 + 4 * fq;
            part[tn][t2] = (m < p.M && n0 < p.N) ? *reinterpret_cast<const floatx4*>(slab + (size_t)m * p.N + n0) : floatx4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tn][th + t2] += part[tn][t2];
      }
    }
  }

  // ---- epilogue: lane holds rows n0..n0+3 (consecutive output channels) of column m
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = tile_m * BM + wm * WM + tm * 16 + fr;
    if (m >= p.M) continue;
    const int bidx = (p.rowbias != nullptr || EPI == EPI_SPLIT_T) ? m / p.rows_per_batch : 0;
#pragma unroll
    for (int tn = 0; tn < TN; tn += (EPI == EPI_GEGLU ? 2 : 1)) {
      const int nt = tile_n * BN + wn * WN + tn * 16;  // first row of this 16-row MFMA tile
      const int n0 = nt + 4 * fq;
      float v[4];
      if (EPI == EPI_GEGLU) {
        // Wt rows are interleaved [16 value rows | 16 gate rows]; output column = nt/2 + 4*fq + i
        const int no = (nt >> 1) + 4 * fq;
        if (no >= (p.N >> 1)) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float xv = acc[tn][tm][i], gv = acc[tn + 1][tm][i];
          if (p.bias) {
            xv += p.bias[n0 + i];
            gv += p.bias[n0 + 16 + i];
          }
          v[i] = xv * af_gelu_erf(gv);
        }
        half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ld_out + no) = h;
      } else {
        if (n0 >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[tn][tm][i];
        if (p.bias) {
          const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + n0);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += bv[i];
        }
        if (p.rowbias) {
          const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)bidx * p.ld_rowbias + n0);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
        }
        if (p.act_silu == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = af_silu(v[i]);
        } else if (p.act_silu == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = v[i] * af_sigmoid(1.702f * v[i]);  // x * sigmoid(1.702 x)
        }
        if (EPI == EPI_SPLIT_T && n0 >= p.split_col) {
          const int tok = m - bidx * p.rows_per_batch;
          half_t* o2 = p.out2 + ((size_t)bidx * (p.N - p.split_col) + (n0 - p.split_col)) * p.ld_out2 + tok;
#pragma unroll
          for (int i = 0; i < 4; ++i) o2[(size_t)i * p.ld_out2] = (half_t)v[i];
          if (tok == p.rows_per_batch - 1) {          // the row pad tok + 1 .. ld_out2 - 1 is part of the output: zero
            for (int t = 1; tok + t < p.ld_out2; ++t)
#pragma unroll
              for (int i = 0; i < 4; ++i) o2[(size_t)i * p.ld_out2 + t] = (half_t)0.f;
          }
        } else {
          if (p.residual) {
            const half4_t rv = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * p.N + n0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
          }
          if (EPI == EPI_STD && p.out_f32) {           // AF_OUT_F32: the accumulator itself (weight gradients)
            *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ld_out + n0) = floatx4{v[0], v[1], v[2], v[3]};
          } else {
            half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ld_out + n0) = h;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile 18 (round 5): SMALL plain GEMMs with their fragments straight from global memory -- no LDS, no barrier, no split-K.
// The training legs issue thousands of GEMMs with a few hundred rows (the CLIP encoders' linears at 388 ... 1552 tokens and their dgrad /
// wgrad, rank-192 adapter projections, batch-1 low-resolution projections): on the register-staged 64 x 64 tile each is a chain of
// K / 64 dependent stages (global -> registers -> LDS -> barrier -> fragments -> MFMA), 9 - 11 us whatever the shape, and the tuned table
// answers with split-K, i.e. a second 5 us launch (profiles/r04s_gemm_census_stage2.txt: 4,513 such GEMMs + 3,890 reduces per Stage-2 micro-batch).
// Both operands are K-contiguous, so an MFMA fragment IS a 16-byte load per lane (row = lane & 15, 8 k = chunk lane >> 4): a wave owns
// 16 rows x 32 columns (one A fragment, two W fragments, two MFMAs per 32-deep step) and keeps D steps of fragments in flight ahead of the
// MFMAs -- a branch-free loop pinned with sched_barrier, as xattn_colmix_mfma_kernel: the launch is one memory round trip plus K / 32 MFMA
// pairs, not K / 64 round trips.  Workgroup = 32 x 64 outputs (4 waves).  Out-of-range rows read a clamped (valid) row and are dropped in
// the epilogue; K is padded by the packed weight's zero columns, against which the A fragment is zero too (clamped load, select at use).
// Scope: taps 1, one source, standard epilogue (bias, per-batch row bias, SiLU / quick-GELU, residual, fp16 or fp32 output), K % 8 == 0,
// 16-byte aligned rows.  Operand-swapped like af_gemm_kernel: a lane owns 4 consecutive output channels of one row.
template <int D>
__global__ __launch_bounds__(256) void af_gemm_direct_kernel(GemmDev p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x - tile_m * p.tiles_n;
  const int m = tile_m * 32 + wm * 16 + fr;
  const int nt = tile_n * 64 + wn * 32;                      // this wave's first output channel
  const int nk = p.kpad >> 5;                                // 32-deep steps (kpad is a multiple of 64)
  const int kc_max = (p.K >> 3) - 1;                         // last valid 8-element chunk of an A row
  const half_t* arow = p.a1 + (size_t)min(m, p.M - 1) * p.lda1;
  const half_t* w0 = p.wt + (size_t)(nt + fr) * p.kpad + 8 * fq;      // rows < npad = roundup(N, 128): always in the packed weight
  const half_t* w1 = w0 + (size_t)16 * p.kpad;
  floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  half8_t xa[D], wa[D], wb[D];
  auto load = [&](int kt, int sl) {
    const int kw = min(kt, nk - 1);                          // steps past the end re-read the last one (their MFMAs do not run)
    const int kc = kw * 4 + fq;
    const half8_t xv = *reinterpret_cast<const half8_t*>(arow + 8 * min(kc, kc_max));
    // chunks past K (the zero padding of the packed weight): the load is clamped to a valid address and the FRAGMENT zeroed -- against zero weights
    // a re-read chunk holding inf / NaN would give NaN (0 x inf) where the staged tiles feed zeros (a select on a loaded value, not on the load)
    xa[sl] = kc > kc_max ? half8_t{0, 0, 0, 0, 0, 0, 0, 0} : xv;
    wa[sl] = *reinterpret_cast<const half8_t*>(w0 + kw * 32);
    wb[sl] = *reinterpret_cast<const half8_t*>(w1 + kw * 32);
  };
#pragma unroll
  for (int sl = 0; sl < D; ++sl) load(sl, sl);
  for (int kt = 0; kt < nk; kt += D) {                       // nk % D == 0 (host)
#pragma unroll
    for (int sl = 0; sl < D; ++sl) {
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[sl], xa[sl], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[sl], xa[sl], acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load(kt + D + sl, sl);
    }
  }
  if (m >= p.M) return;
  const int bidx = p.rowbias != nullptr ? m / p.rows_per_batch : 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n0 = nt + t * 16 + 4 * fq;
    if (n0 >= p.N) continue;
    const floatx4 a = t ? acc1 : acc0;
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.bias) {
      const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += bv[i];
    }
    if (p.rowbias) {
      const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)bidx * p.ld_rowbias + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
    }
    if (p.act_silu == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = af_silu(v[i]);
    } else if (p.act_silu == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = v[i] * af_sigmoid(1.702f * v[i]);
    }
    if (p.residual) {
      const half4_t rv = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * p.N + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
    }
    if (p.out_f32) {
      *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ld_out + n0) = floatx4{v[0], v[1], v[2], v[3]};
    } else {
      const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ld_out + n0) = h;
    }
  }
}

// tile 18's scope (af_gemm falls back to the register-staged 64 x 64 tile outside it)
static bool gemm_direct_eligible(const af_gemm_desc* d, const GemmDev& p) {
  return d->taps == 1 && d->a2 == nullptr && d->c2 == 0 && d->act != AF_ACT_GEGLU && d->out_mode != AF_OUT_SPLIT_T && d->ln_colsum == nullptr &&
         d->gn_partials == nullptr && d->K % 8 == 0 && d->K >= 8 && p.lda1 % 8 == 0 && d->N % 4 == 0 && p.ld_out % 4 == 0 &&
         (((uintptr_t)d->a1 | (uintptr_t)d->wt) & 15) == 0 && d->kpad % 64 == 0 &&
         // the epilogue's vector accesses: floatx4 bias loads and (fp32 output) stores, half4 row-bias / residual loads and fp16 stores
         ((uintptr_t)d->bias & 15) == 0 && (((uintptr_t)d->rowbias | (uintptr_t)d->residual) & 7) == 0 && d->ld_rowbias % 4 == 0 &&
         ((uintptr_t)d->out & (d->out_mode == AF_OUT_F32 ? 15 : 7)) == 0;
}

static int launch_direct(const GemmDev& p0, hipStream_t stream) {
  GemmDev p = p0;
  p.tiles_m = (p.M + 31) / 32;
  p.tiles_n = (p.N + 63) / 64;
  p.splits = 1;
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  if ((p.kpad >> 5) % 4 == 0) hipLaunchKernelGGL(af_gemm_direct_kernel<4>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(af_gemm_direct_kernel<2>, grid, block, 0, stream, p);
  return af_check_launch("af_gemm(tile 18)");
}

// split-K second pass: sum the fp32 partials and apply the standard epilogue (4 channels per thread)
__global__ __launch_bounds__(256) void af_splitk_reduce_kernel(GemmDev p) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N >> 2;
  if (i4 >= (long)p.M * n4) return;
  const int m = (int)(i4 / n4);
  const int n0 = (int)(i4 - (long)m * n4) * 4;
  floatx4 v = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < p.splits; ++sp) v += *reinterpret_cast<const floatx4*>(p.ws + ((size_t)sp * p.M + m) * p.N + n0);
  if (p.bias) v += *reinterpret_cast<const floatx4*>(p.bias + n0);
  if (p.rowbias) {
    const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias + n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
  }
  if (p.act_silu == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = af_silu(v[i]);
  } else if (p.act_silu == 3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] * af_sigmoid(1.702f * v[i]);
  }
  if (p.residual) {
    const half4_t rv = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * p.N + n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
  }
  if (p.out_f32) {
    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ld_out + n0) = v;
    return;
  }
  const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ld_out + n0) = h;
}

template <int BM, int BN, int TAPS, int EPI, bool FAST = false>
int launch(const GemmDev& p0, hipStream_t stream) {
  GemmDev p = p0;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.tiles_m = tiles_m;
  p.n_major = af_gemm_n_major(p.M, p.N, p.K, TAPS == 9 ? p.c1 + p.c2 : p.K);
  {
    static const int wpf_env = getenv("AF_GEMM3_WPREFETCH") ? atoi(getenv("AF_GEMM3_WPREFETCH")) : 4;
    static const int coop_env = getenv("AF_GEMM3_WPF_COOP") ? atoi(getenv("AF_GEMM3_WPF_COOP")) : 32;
    p.wpf = wpf_env > AF_WPF_MAX ? AF_WPF_MAX : wpf_env;
    p.wpf_coop = tiles_m < coop_env ? tiles_m : coop_env;
  }
  const int nk = p.kpad / BK;
  if (p.splits > nk) p.splits = nk;
  p.kt_per_split = (nk + p.splits - 1) / p.splits;
  p.splits = (nk + p.kt_per_split - 1) / p.kt_per_split;  // no empty split
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(half_t) + (BM == 64 ? AF_WPF_DUMP_BYTES : 0);   // staging buffers (+ the weight prefetch's dump area)
  // in-kernel reduction by the last-arriving slice: up to 16 slices here (round 5: the training legs' small outputs, where the separate reduce pass
  // is a 5 us launch over a few KB -- ops.SPLITK_FUSED_BYTES); the caller decides WHEN (af_gemm_desc.splitk_fused), this is only what the code supports
  if (p.counters && (EPI != EPI_STD || p.splits <= 1 || p.splits > 16 || tiles_m * p.tiles_n > AF_SPLITK_MAX_TILES)) p.counters = nullptr;
  dim3 grid(tiles_m * p.tiles_n, p.splits), block(256);
  hipLaunchKernelGGL((af_gemm_kernel<BM, BN, TAPS, EPI, FAST>), grid, block, lds, stream, p);
  if (EPI == EPI_STD && p.splits > 1 && p.counters == nullptr) {
    if (g_defer_reduce != nullptr) {
      *g_defer_reduce = p.splits;                    // the consumer reduces the slabs (af_groupnorm_splitk / af_splitk_reduce)
    } else {
      const long n = (long)p.M * (p.N >> 2);
      hipLaunchKernelGGL(af_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
    }
  }
  return af_check_launch("af_gemm");
}

template <int TAPS, int EPI>
int launch_tile(const GemmDev& p, int tile, hipStream_t stream) {
  static const bool no_fast = getenv("AF_NO_FASTCONV") != nullptr;   // A/B switch for profiling
  if (!no_fast && TAPS == 9 && EPI == EPI_STD && !p.upsample && p.c1 % BK == 0 && p.c2 % BK == 0) {
    if (tile == 1) return launch<128, 128, TAPS, EPI, true>(p, stream);
    return launch<64, 64, TAPS, EPI, true>(p, stream);
  }
  if (tile == 1) return launch<128, 128, TAPS, EPI>(p, stream);
  return launch<64, 64, TAPS, EPI>(p, stream);
}

}  // namespace

extern "C" int af_splitk_reduce(const void* slabs, int splits, const void* bias, const void* rowbias, int ld_rowbias, int rows_per_batch, const void* residual,
                                void* out, int M, int N, void* stream) {
  AF_REQUIRE(slabs && out && splits >= 1 && M > 0 && N > 0 && N % 4 == 0, "af_splitk_reduce: bad arguments");
  AF_REQUIRE(rowbias == nullptr || (ld_rowbias >= N && rows_per_batch > 0), "af_splitk_reduce: rowbias needs ld_rowbias >= N and rows_per_batch > 0");
  AF_REQUIRE((((uintptr_t)slabs | (uintptr_t)bias) & 15) == 0 && (((uintptr_t)rowbias | (uintptr_t)residual | (uintptr_t)out) & 7) == 0 && ld_rowbias % 4 == 0,
             "af_splitk_reduce: slabs / bias must be 16-byte aligned, rowbias / residual / out 8-byte aligned");
  GemmDev p{};
  p.ws = const_cast<float*>((const float*)slabs);
  p.splits = splits;
  p.bias = (const float*)bias;
  p.rowbias = (const half_t*)rowbias;
  p.ld_rowbias = ld_rowbias;
  p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
  p.residual = (const half_t*)residual;
  p.out = (half_t*)out;
  p.M = M;
  p.N = N;
  p.ld_out = N;
  AfLaunchScope scope(AF_FAM_GEMM, stream);
  const long n = (long)M * (N >> 2);
  hipLaunchKernelGGL(af_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return af_check_launch("af_splitk_reduce");
}

// af_gemm3.hip
int af_gemm3_try_launch(const af_gemm_desc* d, int splits, int wide, hipStream_t stream);
int af_gemm3_effective_splits(const af_gemm_desc* d, int splits, int wide);

extern "C" int af_gemm_gn_stats_ok(int tile, int splits, int taps, int act, int out_mode, int N, int cpg, int rows_per_batch) {
  if (tile == 19) tile = 14;                         // the round-5 loop of the halo-resident kernel: same tile, same epilogue
  const int bn = tile == 7 ? 320 : ((tile == 11 || tile == 13) ? 160 : (tile == 14 ? (N % 160 == 0 ? 160 : 128) : 0));   // (tile 14: 256 x 160, or 256 x 128 where N is no 160-multiple)
  if (bn == 0 || splits > 1 || (taps != 1 && taps != 9) || act == AF_ACT_GEGLU || out_mode != AF_OUT_NORMAL) return 0;
  if (tile == 14 && (taps != 9 || rows_per_batch % 256 != 0)) return 0;      // the halo-resident kernel's tile is 256 rows of one image
  if (cpg <= 0 || cpg % 2 != 0 || bn % cpg != 0 || N % cpg != 0 || N / cpg > 32 || N % bn != 0 || N % 8 != 0) return 0;
  if (rows_per_batch <= 0 || rows_per_batch % 128 != 0 || rows_per_batch / 128 > 128) return 0;
  return 1;
}

extern "C" int af_gemm(const af_gemm_desc* d, void* stream) {
  AF_REQUIRE(d != nullptr, "af_gemm: null descriptor");
  AF_REQUIRE(d->a1 && d->wt && d->out, "af_gemm: a1/wt/out must be non-null");
  AF_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "af_gemm: M, N, K must be positive");
  AF_REQUIRE(d->taps == 1 || d->taps == 9, "af_gemm: taps must be 1 or 9");
  AF_REQUIRE(d->kpad % BK == 0 && d->kpad >= d->K, "af_gemm: kpad must be a multiple of 64 and >= K");
  AF_REQUIRE(d->N % 4 == 0, "af_gemm: N must be a multiple of 4");
  AF_REQUIRE(d->c1 > 0 && d->c1 % 8 == 0 && d->c2 >= 0 && d->c2 % 8 == 0, "af_gemm: c1/c2 must be multiples of 8");
  AF_REQUIRE(d->c2 == 0 || d->a2 != nullptr, "af_gemm: a2 is null but c2 > 0");
  const bool ktail = d->c3 > 0 || d->c4 > 0;         // K-concatenated 1x1 tail (the ResBlock's shortcut inside its second convolution)
  if (ktail) {
    AF_REQUIRE(d->taps == 9 && d->c3 > 0 && d->a3 != nullptr && d->c4 >= 0 && (d->c4 == 0 || d->a4 != nullptr), "af_gemm: the K tail needs taps == 9 and a3 (a4 when c4 > 0)");
    AF_REQUIRE(d->c3 % 64 == 0 && d->c4 % 64 == 0, "af_gemm: c3 / c4 must be multiples of 64");
    AF_REQUIRE((d->lda3 == 0 || (d->lda3 >= d->c3 && d->lda3 % 8 == 0)) && (d->lda4 == 0 || (d->lda4 >= d->c4 && d->lda4 % 8 == 0)), "af_gemm: bad lda3 / lda4");
    AF_REQUIRE((d->stride == 0 || d->stride == 1) && d->upsample == 0 && d->tap_shift == 0, "af_gemm: the K tail needs stride 1, no upsample, no tap_shift");
    AF_SUPPORTED((d->tile >= 7 && d->tile <= 15), "af_gemm: the K tail runs on the whole-line tiles 7 .. 13, 15 and the halo-resident tile 14 only");
  }
  AF_REQUIRE(d->K == d->taps * (d->c1 + d->c2) + d->c3 + d->c4, "af_gemm: K != taps*(c1+c2) (+ c3 + c4)");
  g_defer_reduce = nullptr;
  if (d->defer_reduce != nullptr) {
    AF_REQUIRE(d->act == AF_ACT_NONE && d->out_mode == AF_OUT_NORMAL && (d->ld_out == 0 || d->ld_out == d->N) && !d->splitk_fused,
               "af_gemm: defer_reduce needs the standard epilogue without activation, a dense fp16 output and the two-launch split-K form");
    *d->defer_reduce = 0;
    g_defer_reduce = d->defer_reduce;
  }
  struct DeferScope { ~DeferScope() { g_defer_reduce = nullptr; } } defer_scope;
  const bool gnp = d->gn_partials != nullptr;
  if (gnp) {
    const int rpb = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
    AF_SUPPORTED(af_gemm_gn_stats_ok(d->tile, d->splits, d->taps, d->act, d->out_mode, d->N, d->gn_cpg, rpb) == 1,
                 "af_gemm: gn_partials needs a whole-line tile whose width is a multiple of gn_cpg, the standard fp16 epilogue, whole 128-row tiles per "
                 "batch item and no split-K (af_gemm_gn_stats_ok)");
    AF_REQUIRE(d->M % rpb == 0 && (d->ld_out == 0 || d->ld_out == d->N) && ((uintptr_t)d->out & 15) == 0, "af_gemm: gn_partials needs M = B * rows_per_batch and a dense, 16-byte aligned output");
  }
  GemmDev p;
  p.a1 = (const half_t*)d->a1;
  p.a2 = (const half_t*)d->a2;
  p.wt = (const half_t*)d->wt;
  p.bias = (const float*)d->bias;
  p.rowbias = (const half_t*)d->rowbias;
  p.residual = (const half_t*)d->residual;
  p.out = (half_t*)d->out;
  p.out2 = (half_t*)d->out2;
  p.M = d->M;
  p.N = d->N;
  p.K = d->K;
  p.kpad = d->kpad;
  p.c1 = d->c1;
  p.c2 = d->c2;
  p.lda1 = d->lda1 ? d->lda1 : d->c1;
  p.lda2 = d->lda2 ? d->lda2 : d->c2;
  p.stride = d->stride ? d->stride : 1;
  p.upsample = d->upsample;
  p.H = d->H;
  p.W = d->W;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.HoWo = d->Ho * d->Wo;
  p.Heff = p.upsample ? 2 * d->H : d->H;
  p.Weff = p.upsample ? 2 * d->W : d->W;
  if (p.upsample == 2 && d->taps == 9) {  // zero-inserted image may be declared one short (odd forward input size)
    p.Heff = d->Ho;
    p.Weff = d->Wo;
  }
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.ld_rowbias = d->ld_rowbias;
  p.act_silu = d->act == AF_ACT_SILU ? 1 : (d->act == AF_ACT_QUICKGELU ? 3 : 0);
  p.split_col = d->split_col;
  p.ld_out2 = d->ld_out2;
  p.tiles_n = 0;
  p.tap_shift = d->taps == 9 ? d->tap_shift : 0;
  p.splits = d->splits > 1 ? d->splits : 1;
  p.kt_per_split = 0;
  p.ws = (float*)d->workspace;
  p.counters = (d->splitk_fused && d->workspace && d->workspace_bytes > AF_SPLITK_COUNTER_BYTES)
                   ? reinterpret_cast<int*>(static_cast<char*>(d->workspace) + d->workspace_bytes - AF_SPLITK_COUNTER_BYTES) : nullptr;
  if (d->taps == 9) {
    AF_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "af_gemm: conv geometry missing");
    AF_REQUIRE(p.stride == 1 || p.stride == 2, "af_gemm: stride must be 1 or 2");
    AF_REQUIRE(p.upsample >= 0 && p.upsample <= 2, "af_gemm: upsample must be 0, 1 (nearest) or 2 (zero-insert)");
    AF_REQUIRE(!(p.upsample && p.stride != 1), "af_gemm: upsample requires stride 1");
    AF_REQUIRE(d->M == d->B * d->Ho * d->Wo, "af_gemm: M != B*Ho*Wo");
    if (p.upsample == 2)
      AF_REQUIRE((d->Ho == 2 * d->H || d->Ho == 2 * d->H - 1) && (d->Wo == 2 * d->W || d->Wo == 2 * d->W - 1),
                 "af_gemm: zero-insert mode needs Ho in {2H, 2H-1}, Wo in {2W, 2W-1}");
    AF_REQUIRE(d->Ho == (p.Heff + 2 - 3) / p.stride + 1 && d->Wo == (p.Weff + 2 - 3) / p.stride + 1,
               "af_gemm: Ho/Wo inconsistent with H/W/stride/upsample");
  }
  if (d->rowbias) AF_REQUIRE(d->ld_rowbias >= d->N, "af_gemm: ld_rowbias < N");
  const bool geglu = d->act == AF_ACT_GEGLU;
  if (geglu) {
    AF_REQUIRE(d->N % 32 == 0, "af_gemm: GEGLU needs N % 32 == 0 (interleaved 16-row groups)");
    AF_REQUIRE(d->out_mode == AF_OUT_NORMAL && !d->rowbias && !d->residual, "af_gemm: GEGLU epilogue is exclusive");
    p.ld_out = d->ld_out ? d->ld_out : d->N / 2;
  } else {
    p.ld_out = d->ld_out ? d->ld_out : (d->out_mode == AF_OUT_SPLIT_T ? d->split_col : d->N);
  }
  if (d->out_mode == AF_OUT_SPLIT_T) {
    AF_REQUIRE(d->out2 != nullptr, "af_gemm: AF_OUT_SPLIT_T needs out2");
    AF_REQUIRE(d->split_col % 16 == 0 && d->split_col >= 0 && d->split_col < d->N, "af_gemm: bad split_col");
    AF_REQUIRE(d->rows_per_batch > 0 && d->ld_out2 >= d->rows_per_batch, "af_gemm: bad ld_out2/rows_per_batch");
    AF_REQUIRE(!d->residual, "af_gemm: residual unsupported with AF_OUT_SPLIT_T");
  }
  AF_REQUIRE(p.ld_out % 4 == 0, "af_gemm: ld_out must be a multiple of 4");
  p.out_f32 = d->out_mode == AF_OUT_F32;
  if (p.out_f32)
    AF_REQUIRE(!geglu && !d->splitk_fused && d->N % 4 == 0,
               "af_gemm: AF_OUT_F32 needs the standard epilogue, N % 4 == 0 and (with split-K) the two-launch form");
  if (p.splits > 1) {
    AF_REQUIRE(!geglu && d->out_mode != AF_OUT_SPLIT_T, "af_gemm: split-K only with the standard epilogue");
    AF_REQUIRE(d->workspace != nullptr && d->workspace_bytes - (d->splitk_fused ? AF_SPLITK_COUNTER_BYTES : 0) >= (int64_t)p.splits * d->M * d->N * 4,
               "af_gemm: split-K needs workspace >= splits*M*N*4 bytes (+ AF_SPLITK_COUNTER_BYTES of zeroed counters with splitk_fused)");
  }

  int tile = d->tile;
  if (tile == 0) {
    const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
    tile = (t128 >= 192 && d->N >= 96) ? 1 : 2;
  }
  AF_REQUIRE(tile >= 1 && tile <= 19, "af_gemm: tile must be 0 .. 19");

  AfLaunchScope scope(AF_FAM_GEMM, stream);
  hipStream_t s = (hipStream_t)stream;
  AF_REQUIRE(d->tap_shift == 0 || (d->tap_shift == 1 && d->taps == 9 && !p.upsample), "af_gemm: tap_shift is 0 or 1 (3x3, no upsample)");
  if (tile == 18) {                                  // small GEMMs straight from global memory; outside its scope: the 64 x 64 tile
    if (gemm_direct_eligible(d, p)) return launch_direct(p, s);
    tile = 2;
  }
  if (tile >= 3 && d->tap_shift) tile = 1;          // the ring kernel keeps the symmetric-padding loader only
  if (tile >= 3) {
    const int eff = af_gemm3_effective_splits(d, p.splits, tile - 3);
    // AF_OUT_F32: the LDS-DMA kernels only reach fp32 through the split-K reduce pass; unsplit, the register-staged kernel stores it
    const int rc3 = (p.out_f32 && eff < 2) ? 1 : af_gemm3_try_launch(d, p.splits, tile - 3, s);
    if (rc3 == 0) return af_check_launch("af_gemm(tile 3)");
    if (rc3 == 2) {
      if (g_defer_reduce != nullptr) {
        *g_defer_reduce = eff;
        return af_check_launch("af_gemm(tile 3, split-K, reduce deferred)");
      }
      p.splits = eff;
      p.ld_out = d->ld_out ? d->ld_out : d->N;
      const long n = (long)p.M * (p.N >> 2);
      hipLaunchKernelGGL(af_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
      return af_check_launch("af_gemm(tile 3, split-K)");
    }
    AF_SUPPORTED(!ktail, "af_gemm: the K tail is outside the chosen tile's scope");
    AF_SUPPORTED(!gnp, "af_gemm: gn_partials is outside the chosen tile's scope");
    tile = 1;  // outside the pipelined kernel's scope
  }
  AF_SUPPORTED(d->ln_colsum == nullptr, "af_gemm: a folded LayerNorm (ln_colsum) needs a whole-line tile (7 .. 13, 16, 17) whose scope covers the shape, "
                                        "taps == 1, c2 == 0 and no split-K");
  if (geglu) return d->taps == 9 ? af_fail(AF_E_UNSUPPORTED, "af_gemm: GEGLU on a 3x3 conv")
                                 : launch_tile<1, EPI_GEGLU>(p, tile, s);
  if (d->out_mode == AF_OUT_SPLIT_T)
    return d->taps == 9 ? af_fail(AF_E_UNSUPPORTED, "af_gemm: SPLIT_T on a 3x3 conv") : launch_tile<1, EPI_SPLIT_T>(p, tile, s);
  return d->taps == 9 ? launch_tile<9, EPI_STD>(p, tile, s) : launch_tile<1, EPI_STD>(p, tile, s);
}
