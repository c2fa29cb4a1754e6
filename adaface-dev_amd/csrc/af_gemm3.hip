// af_gemm3.hip -- pipelined variant of the GEMM / implicit-conv kernel ("tile 3"):
// operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write) into a
// 4-slot ring of 128 x 32 (A) + 128 x 32 (W) fp16 tiles, three K-steps ahead of the MFMAs; a wave waits
// only for its own oldest stage with a COUNTED s_waitcnt vmcnt(8) and one raw s_barrier per K-step publishes
// it to the other waves.  The rocprofv3 PMC profile of the register-staged kernel (af_gemm.hip) showed the
// MFMA pipe ~33 % busy with both resident waves parked on the per-K-step chain
// {global-load latency -> vmcnt(0) -> 8 x ds_write_b128 -> barrier -> ds_read latency}; this structure takes
// the load latency and the LDS writes off that chain.
//
// LDS image: rows of 32 halves (64 B).  One LDS-DMA wave-instruction writes 1 KiB = 16 rows x 4 chunks of 16 B,
// lane-linear (lane -> row lane>>2, physical chunk lane&3), so the XOR swizzle that makes the MFMA fragment
// reads (ds_read_b128, 16 rows x one chunk) conflict-free is applied on the SOURCE side: the lane that fills
// physical chunk p of row r fetches logical chunk p ^ PI[(r>>2)&3], PI = {0,2,3,1}; the reader applies the same
// involution.  Halo / out-of-range lanes fetch from a 16-byte zero page.
//
// Scope: standard epilogue (bias, per-batch row bias, SiLU / quick-GELU, residual, split-K partials); plain rows or
// 3x3 taps with channel counts that are multiples of 32 and no upsampling.  Everything else stays on af_gemm.hip.
#include <stdlib.h>

#include <algorithm>

#include "af_common.h"

int af_gemm_n_major(int M, int N, int K, int cin);   // af_gemm.hip

namespace {

struct Gemm3Dev {
  const half_t* a1;
  const half_t* a2;
  const half_t* wt;
  const half_t* zeros;
  const float* bias;
  const half_t* rowbias;
  const half_t* residual;
  half_t* out;
  half_t* out2;
  int split_col, ld_out2;
  int M, N, K, kpad, npad;
  int c1, c2, lda1, lda2;
  const half_t* a3;      // K-concatenated 1x1 tail of a 3x3 convolution (af_gemm_desc.a3 / a4): plain rows behind the nine tap blocks
  const half_t* a4;
  int c3, c4, lda3, lda4;
  int H, W, Ho, Wo, HoWo, stride;
  int upsample;   // 1: nearest x2 folded into the 3x3 gather (whole-line kernel only): Ho = 2H, Wo = 2W
  int rows_per_batch, ld_rowbias, act, ld_out;
  int tiles_n, tiles_m, n_major, splits, kt_per_split;
  float* ws;
  int* counters;  // in-kernel split-K reduction: one arrival counter per output tile (zero between launches); nullptr = separate reduce pass
  int ablate;  // profiling only (AF_GEMM3_ABLATE): 1 = no DMA after the prologue, 2 = fragments read once, 4 = no barrier,
               // 8 = no main loop, 16 = no epilogue, 32 = direct (un-staged) epilogue stores
  int stage_ok;  // output rows can be written as 16-byte chunks (N, ld_out multiples of 8, 16-byte aligned base)
  const float* ln_cs;   // LayerNorm folded into the GEMM (af_gemm_desc.ln_colsum): column sums of the packed weight, or nullptr
  float ln_eps;
  float* gn_ws;         // af_gemm_desc.gn_partials: GroupNorm partial statistics of the output tile from the staged standard epilogue, or nullptr
  int gn_cpg;
  int wpf;              // whole-line kernel: weight-tile L2 prefetch at kernel start, wave instructions per wave (0 = off; AF_GEMM3_WPREFETCH)
  int wpf_coop;         // workgroups that share one weight tile and split its prefetch (min(tiles_m, 32))
  int patch_tx;         // halo-resident kernel, PATCH form (images wider than 64 pixels): 16 x 16-pixel tiles, patch_tx of them per image row, patch_tpi per image
  int patch_tpi;
};

constexpr int BK3 = 32;
constexpr int NST_DEFAULT = 4;         // ring slots (template parameter NST of the kernel; 5 where 160 KB of LDS allow it)


// PI = {0, 2, 3, 1} packed two bits per entry: 0b01'11'10'00 = 0x78, PI[g] = (0x78 >> 2g) & 3

__device__ __forceinline__ void glds16(const half_t* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// The same instruction as inline asm (halo-resident kernel): for the builtin, hipcc retires every pending ds_read (s_waitcnt lgkmcnt(0)) in front of an
// LDS-DMA issue it cannot prove disjoint from them -- the ping-pong loop issues its pieces right behind its fragment reads, into a slot / buffer that
// nobody reads (barrier-ordered, see there).  lds_dst must be wave-uniform.  M0 is named as clobbered: the compiler re-materialises it for its own uses.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_nowait(const half_t* src, char* lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"((unsigned)(size_t)lds_dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// LDS-DMA with a SCALAR base pointer and a 32-bit per-lane byte offset (round 6): what is per-lane about a piece's source is computed once, the
// per-stage part (K offset, tap) is scalar arithmetic.  lds_dst wave-uniform.
__device__ __forceinline__ void glds16_sbase(const half_t* sbase, unsigned voff, char* lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"((unsigned)(size_t)lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}
// the same with the lanes whose `pix` is negative switched off (no LDS write for them)
__device__ __forceinline__ void glds16_sbase_masked(const half_t* sbase, unsigned voff, int pix, char* lds_dst) {
  unsigned long long keep;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile(
      "v_cmp_lt_i32 vcc, -1, %2\n\t"
      "s_and_saveexec_b64 %0, vcc\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(keep)
      : "v"(voff), "v"(pix), "s"(sbase), "s"((unsigned)(size_t)lds_dst)
      : "memory", "m0", "vcc");
#pragma clang diagnostic pop
}

// Tile shape: NWM x NWN waves, each wave 64 (M) x 16*TN (N).  Two instantiations:
//   <2, 2, 4>: 128 x 128, 4 waves, 2 workgroups per CU             (64 FLOP per operand byte)
//   <2, 4, 5>: 128 x 320, 8 waves, 1 workgroup per CU              (91 FLOP per operand byte)
// The ablation in profiles/r01d_gemm_ablation.txt shows the 128 x 128 kernel is bound by L2 -> CU operand
// traffic (no-DMA build: 555 -> 900 TFLOP/s; no-LDS-read build: unchanged), and every channel count of SD-1.5 is a
// multiple of 320, so the wide tile raises the intensity AND tiles N exactly (M = 32768, N = 320 -> 256 tiles).
// Every wave issues exactly 4 LDS-DMA pieces per stage in both shapes (2 A + 2 W, or 1 A + 3 W with the W pieces
// padded from 20 to 24; padding pieces fetch the zero page), so the counted waits are the same constants.
enum { E3_STD = 0, E3_GEGLU = 1, E3_SPLIT_T = 2 };
#ifndef AF_GEMM3W_DIET
#define AF_GEMM3W_DIET 1
#endif

// Epilogue shared by the ring kernel and the whole-line kernel: split-K partial tiles, or bias / row bias / activation / residual
// (identical arithmetic to af_gemm.hip's standard epilogue), GEGLU, transposed-V split.  LDSB = bytes of LDS the main loop owned.
template <int EPI, int NWM, int NWN, int TN, int LDSB, bool PATCH = false>
__device__ __forceinline__ void gemm3_epilogue(const Gemm3Dev& p, floatx4 (&acc)[TN][4], char* af_smem, int tile_m, int tile_n, int wm, int wn,
                                               int fr, int fq, int tid, const float* lnst = nullptr) {
  constexpr int TM = 4, NW = NWM * NWN, BM = NWM * 64, BN = NWN * TN * 16;
  // PATCH (halo-resident kernel on images wider than 64 pixels): the tile's 256 rows are a 16 x 16-pixel patch of one image, local row q = pixel
  // (q >> 4, q & 15) of the patch; staged standard epilogue without split-K only (the host guarantees it)
  int patch_m0 = 0;
  if constexpr (PATCH) {
    const int im = tile_m / p.patch_tpi, t = tile_m - im * p.patch_tpi;
    const int py = t / p.patch_tx, px = t - py * p.patch_tx;
    patch_m0 = (im * p.Ho + py * 16) * p.Wo + px * 16;
  }
  auto tile_row = [&](int row) { return PATCH ? patch_m0 + (row >> 4) * p.Wo + (row & 15) : tile_m * BM + row; };
  if (lnst != nullptr) {
    // LayerNorm folded into this GEMM: acc = x . (gamma W)^T of the UN-normalised rows; LN(x) W^T = rstd * (acc - mean * colsum) (+ b + W beta,
    // which is the packed bias).  lnst[row] = (mean, rstd) of the tile's rows, written by the main loop's statistics waves.
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = wm * 64 + tm * 16 + fr;
      const float mean = lnst[2 * row], rstd = lnst[2 * row + 1];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const floatx4 cs = *reinterpret_cast<const floatx4*>(lnst + 2 * BM + wn * TN * 16 + tn * 16 + 4 * fq);   // the tile's column sums, staged in LDS
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[tn][tm][e] = rstd * (acc[tn][tm][e] - mean * cs[e]);
      }
    }
  }
  // ---- epilogue (identical arithmetic to af_gemm.hip's standard epilogue)
  if (p.splits > 1) {
    float* wsp = p.ws + (size_t)blockIdx.y * p.M * p.N;
    if (p.counters != nullptr) {
      // slabs handed to another workgroup inside this launch: WRITE-THROUGH (sc1) 16-byte stores, so that no release fence (an L2
      // write-back of every dirty line, ~20 us with every workgroup's slab dirty) is needed before the arrival counter
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(wsp, 0, p.M * p.N * 4, 0x00020000);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = tile_m * BM + wm * 64 + tm * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int n0 = tile_n * BN + wn * TN * 16 + tn * 16 + 4 * fq;
          if (n0 < p.N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, acc[tn][tm]), rsrc, (m * p.N + n0) * 4, 0, 16);
        }
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = tile_m * BM + wm * 64 + tm * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int n0 = tile_n * BN + wn * TN * 16 + tn * 16 + 4 * fq;
          if (n0 < p.N) *reinterpret_cast<floatx4*>(wsp + (size_t)m * p.N + n0) = acc[tn][tm];
        }
      }
    }
    if (TN > 5 || p.counters == nullptr) return;      // the caller runs af_splitk_reduce_kernel (TN > 5: tiles that never split)
    // ---- in-kernel reduction (no second launch): every K-slice publishes its slab, the LAST slice to arrive at the tile's counter
    // sums all slabs in slice order (the same order as the reduce kernel: bit-identical results, independent of arrival order)
    // and runs the epilogue.  Hand-off = the counter form of the agent-scope release / acquire protocol (cdna_hip_programming.md,
    // Guideline 16 recipe R1 / "Projection GEMM" item 2): write-through (sc1) slab stores -> every storing wave drains its stores ->
    // workgroup barrier -> one lane: relaxed agent fetch_add; the last arriver: agent acquire, drained, barrier, plain loads.
    // Placement independent; the counter is restored to zero by the last arriver (launch boundaries order it for the next launch).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* last_flag = reinterpret_cast<int*>(af_smem);  // the staging ring is idle; ONE LDS object in the kernel (no second __shared__)
    if (tid == 0) {
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
      const int m = tile_m * BM + wm * 64 + tm * 16 + fr;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n0 = tile_n * BN + wn * TN * 16 + tn * 16 + 4 * fq;
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
          const int m = tile_m * BM + wm * 64 + tm * 16 + fr;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const int n0 = tile_n * BN + wn * TN * 16 + tn * 16 + 4 * fq;
            part[tn][t2] = (m < p.M && n0 < p.N) ? *reinterpret_cast<const floatx4*>(slab + (size_t)m * p.N + n0) : floatx4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tn][th + t2] += part[tn][t2];
      }
    }
    // fall through: acc now holds the full sums, the ordinary epilogue follows
  }
  // Staged epilogue: the tile is assembled in LDS (the ring is idle now) and written out as whole rows, 16 bytes per lane, instead
  // of 8-byte stores that touch 16 rows x 32 bytes per wave instruction.  Measured on M32768 N2560 K320 the direct form spends
  // 66 of 141 us in the epilogue (profiles/r01r_gemm_epilogue.txt).
  constexpr int BNO = EPI == E3_GEGLU ? BN / 2 : BN;          // output columns of the tile
  constexpr int TS = BNO + 8;                                  // staging row stride (halves): 16 bytes of padding
  constexpr bool kCanStage = (EPI == E3_STD || EPI == E3_GEGLU) && (size_t)BM * TS * 2 <= (size_t)LDSB;
  if (kCanStage && (PATCH || (p.stage_ok && !(p.ablate & 32)))) {
    __syncthreads();                                           // every wave is done reading the last ring slot
    half_t* T = reinterpret_cast<half_t*>(af_smem);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = wm * 64 + tm * 16 + fr;
      const int m = tile_row(row);
      const bool mok = m < p.M;
      const int bidx = (p.rowbias && mok) ? m / p.rows_per_batch : 0;
      if (EPI == E3_GEGLU) {
#pragma unroll
        for (int tn = 0; tn + 1 < TN; tn += 2) {
          const int nt = tile_n * BN + wn * TN * 16 + tn * 16;
          const int n0 = nt + 4 * fq;
          const int col = ((wn * TN * 16 + tn * 16) >> 1) + 4 * fq;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (mok && (nt >> 1) + 4 * fq < (p.N >> 1)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float xv = acc[tn][tm][e], gv = acc[tn + 1][tm][e];
              if (p.bias) {
                xv += p.bias[n0 + e];
                gv += p.bias[n0 + 16 + e];
              }
              v[e] = xv * af_gelu_erf(gv);
            }
          }
          const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<half4_t*>(T + row * TS + col) = h;
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = wn * TN * 16 + tn * 16 + 4 * fq;
          const int n0 = tile_n * BN + col;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (mok && n0 < p.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[tn][tm][e];
            if (p.bias) {
              const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + n0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bv[e];
            }
            if (p.rowbias) {
              const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)bidx * p.ld_rowbias + n0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            if (p.act == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = af_silu(v[e]);
            } else if (p.act == 3) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] * af_sigmoid(1.702f * v[e]);
            }
            if (p.residual) {
              const half4_t rv = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * p.N + n0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
          }
          const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<half4_t*>(T + row * TS + col) = h;
        }
      }
    }
    __syncthreads();
    constexpr bool kGnFits = EPI == E3_STD && (BM == 128 || BM == 256) && (size_t)BM * TS * 2 + 2 * ((64 * NW) / 32) * 32 * 4 <= (size_t)LDSB;
    if constexpr (kGnFits) if (p.gn_ws != nullptr) {
      // GroupNorm partial statistics of THIS tile from the staged fp16 values (what the consumer's normalisation will read): thread (group g,
      // row slice rs) sums its rows x cpg / 2 channel pairs, the slices meet in LDS behind the tile, one partial per (128-row block, group) leaves
      // for af_groupnorm's partial workspace [B][128][32][2] at the block's index inside its image.  A partial is (sum, M2 about the block's own
      // mean) -- af_common.h, GroupNorm partial statistics -- from sums shifted by the group's first element of the block (the same pivot in every
      // slice, so the slices add up).  A 256-row tile (the halo-resident convolution) leaves its two 128-row blocks separately.  The tile's width
      // is a multiple of the group width and M a multiple of 128 (validated by the host).
      constexpr int NSL = (64 * NW) / 32;                        // row slices: 32 threads (groups) per slice
      constexpr int RPS = BM / NSL;
      constexpr int NBK = BM / 128, SPB = NSL / NBK;             // 128-row blocks per tile, slices per block
      float* red = reinterpret_cast<float*>(af_smem + (size_t)BM * TS * 2);
      const int g = tid & 31, rs = tid >> 5;
      const int ng = BNO / p.gn_cpg;
      float s = 0.f, q = 0.f;
      if (g < ng) {
        const int hp = p.gn_cpg >> 1;
        const half_t pivot = T[(rs / SPB) * 128 * TS + g * p.gn_cpg];
        const half_t nhp = pivot * (half_t)-0.5f;
        const half2_t np2 = {nhp, nhp}, one2 = {(half_t)1.0f, (half_t)1.0f};
        for (int r = rs * RPS; r < (rs + 1) * RPS; ++r) {
          const half2_t* tp = reinterpret_cast<const half2_t*>(T + r * TS + g * p.gn_cpg);
          for (int j = 0; j < hp; ++j) {
            const half2_t d2 = gn_half_diff(tp[j], np2);            // (x - pivot) / 2 in packed fp16 + dot2 (af_common.h; af_norm.hip, gn_partial_kernel)
            s = __builtin_amdgcn_fdot2(d2, one2, s, false);
            q = __builtin_amdgcn_fdot2(d2, d2, q, false);
          }
        }
        s *= 2.f;
        q *= 4.f;
      }
      red[rs * 32 + g] = s;
      red[NSL * 32 + rs * 32 + g] = q;
      __syncthreads();
      if (tid < 32 * NBK && (tid & 31) < ng) {
        const int gg = tid & 31, bl = tid >> 5;
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int k = 0; k < SPB; ++k) {
          ss += red[(bl * SPB + k) * 32 + gg];
          qq += red[NSL * 32 + (bl * SPB + k) * 32 + gg];
        }
        const GnAcc acc = gn_acc_from_shifted(128.f * (float)p.gn_cpg, (float)T[bl * 128 * TS + gg * p.gn_cpg], ss, qq);
        const int m0 = tile_m * BM;
        const int bt = m0 / p.rows_per_batch, blk = (m0 - bt * p.rows_per_batch) / 128 + bl;
        const int g0 = (tile_n * BNO) / p.gn_cpg;
        float* w = p.gn_ws + (((size_t)bt * 128 + blk) * 32 + g0 + gg) * 2;
        w[0] = acc.s;
        w[1] = acc.m2;
      }
    }
    constexpr int CPR = BNO / 8;                                // 16-byte chunks per output row
    const int ncols = EPI == E3_GEGLU ? (p.N >> 1) : p.N;
    for (int c = tid; c < BM * CPR; c += 64 * NW) {
      const int row = c / CPR, cc = c - row * CPR;
      const int m = tile_row(row), n = tile_n * BNO + cc * 8;
      if (m < p.M && n < ncols)
        *reinterpret_cast<half8_t*>(p.out + (size_t)m * p.ld_out + n) = *reinterpret_cast<const half8_t*>(T + row * TS + cc * 8);
    }
    return;
  }
  if constexpr (EPI == E3_SPLIT_T) {
    // Staged forms of the transposed-V split (round 4).  The direct form below writes q | k with 8-byte stores that touch 16 rows x 32 bytes per wave
    // instruction and V^T with 2-BYTE stores (a lane's four channels are four different V^T rows) -- at M 32768, N 960, K 320 (the C = 320 q | k | v
    // projection, 63 MB of output) that epilogue is most of the kernel.  When a tile lies on one side of split_col it is assembled in the idle ring
    // instead: q / k tiles as rows (the standard staged form), V tiles TRANSPOSED ([channel][token]) and written as 16-byte runs of 8 tokens along
    // each V^T row.  Needs a tile's rows to be tokens of ONE batch item (rows_per_batch % BM == 0) and aligned outputs; anything else stays direct.
    constexpr int TS = BN + 8;                                  // row staging stride (halves)
    constexpr int TT = BM + 8;                                  // transposed staging stride: BM tokens + 16 bytes
    constexpr bool kFits = (size_t)BM * TS * 2 <= (size_t)LDSB && (size_t)BN * TT * 2 <= (size_t)LDSB;
    const int tile_lo = tile_n * BN;
    const bool rows_side = tile_lo + BN <= p.split_col;
    const bool vt_side = tile_lo >= p.split_col;
    const int nv = p.N - p.split_col;
    const bool ok_rows = rows_side && p.split_col % 8 == 0 && p.ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
    const bool ok_vt = vt_side && p.rows_per_batch % BM == 0 && p.ld_out2 % 8 == 0 && (reinterpret_cast<uintptr_t>(p.out2) & 15) == 0 &&
                       p.ld_out2 >= p.rows_per_batch;
    if (kFits && (ok_rows || ok_vt) && !(p.ablate & 32)) {
      __syncthreads();                                          // every wave is done reading the last ring slot
      half_t* T = reinterpret_cast<half_t*>(af_smem);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = wm * 64 + tm * 16 + fr;
        const int m = tile_m * BM + row;
        const bool mok = m < p.M;
        const int bidx = (p.rowbias && mok) ? m / p.rows_per_batch : 0;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = wn * TN * 16 + tn * 16 + 4 * fq;
          const int n0 = tile_lo + col;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (mok && n0 < p.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[tn][tm][e];
            if (p.bias) {
              const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + n0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bv[e];
            }
            if (p.rowbias) {
              const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)bidx * p.ld_rowbias + n0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            if (p.act == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = af_silu(v[e]);
            } else if (p.act == 3) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] * af_sigmoid(1.702f * v[e]);
            }
          }
          if (ok_rows) {
            const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4_t*>(T + row * TS + col) = h;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) T[(col + e) * TT + row] = (half_t)v[e];
          }
        }
      }
      __syncthreads();
      if (ok_rows) {
        constexpr int CPR = BN / 8;                             // 16-byte chunks per output row
        for (int c = tid; c < BM * CPR; c += 64 * NW) {
          const int row = c / CPR, cc = c - row * CPR;
          const int m = tile_m * BM + row, n = tile_lo + cc * 8;
          if (m < p.M && n < p.split_col)
            *reinterpret_cast<half8_t*>(p.out + (size_t)m * p.ld_out + n) = *reinterpret_cast<const half8_t*>(T + row * TS + cc * 8);
        }
      } else {
        constexpr int CPC = BM / 8;                             // 16-byte chunks (8 tokens) per V^T row of the tile
        const int m0 = tile_m * BM;                             // the tile's rows are tokens tok0 .. tok0 + BM - 1 of batch item bt
        const int bt = m0 / p.rows_per_batch, tok0 = m0 - bt * p.rows_per_batch;
        half_t* o2 = p.out2 + ((size_t)bt * nv + (tile_lo - p.split_col)) * p.ld_out2 + tok0;
        for (int c = tid; c < BN * CPC; c += 64 * NW) {
          const int col = c / CPC, cc = c - col * CPC;
          if (tile_lo + col < p.N && m0 + cc * 8 < p.M)
            *reinterpret_cast<half8_t*>(o2 + (size_t)col * p.ld_out2 + cc * 8) = *reinterpret_cast<const half8_t*>(T + col * TT + cc * 8);
        }
        // the row pad tok rows_per_batch .. ld_out2 - 1 belongs to this output: zero (consumers multiply it by P = 0)
        if (tok0 + BM == p.rows_per_batch && p.ld_out2 > p.rows_per_batch) {
          const int padn = p.ld_out2 - p.rows_per_batch;
          for (int c = tid; c < BN * padn; c += 64 * NW) {
            const int col = c / padn, t = c - col * padn;
            if (tile_lo + col < p.N) o2[(size_t)col * p.ld_out2 + BM + t] = (half_t)0.f;
          }
        }
      }
      return;
    }
  }
  if (EPI == E3_GEGLU) {
    // W rows interleaved [16 value | 16 gate]: adjacent MFMA n-tiles pair up in the same lane / register
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = tile_m * BM + wm * 64 + tm * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int tn = 0; tn + 1 < TN; tn += 2) {
        const int nt = tile_n * BN + wn * TN * 16 + tn * 16;
        const int n0 = nt + 4 * fq, no = (nt >> 1) + 4 * fq;
        if (no >= (p.N >> 1)) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xv = acc[tn][tm][e], gv = acc[tn + 1][tm][e];
          if (p.bias) {
            xv += p.bias[n0 + e];
            gv += p.bias[n0 + 16 + e];
          }
          v[e] = xv * af_gelu_erf(gv);
        }
        const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ld_out + no) = h;
      }
    }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = tile_m * BM + wm * 64 + tm * 16 + fr;
    if (m >= p.M) continue;
    const int bidx = (p.rowbias || EPI == E3_SPLIT_T) ? m / p.rows_per_batch : 0;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n0 = tile_n * BN + wn * TN * 16 + tn * 16 + 4 * fq;
      if (n0 >= p.N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[tn][tm][e];
      if (p.bias) {
        const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + n0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bv[e];
      }
      if (p.rowbias) {
        const half4_t rv = *reinterpret_cast<const half4_t*>(p.rowbias + (size_t)bidx * p.ld_rowbias + n0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
      }
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = af_silu(v[e]);
      } else if (p.act == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * af_sigmoid(1.702f * v[e]);
      }
      if (EPI == E3_SPLIT_T && n0 >= p.split_col) {   // V columns: written transposed [B][N - split_col][ld_out2]
        const int tok = m - bidx * p.rows_per_batch;
        half_t* o2 = p.out2 + ((size_t)bidx * (p.N - p.split_col) + (n0 - p.split_col)) * p.ld_out2 + tok;
#pragma unroll
        for (int e = 0; e < 4; ++e) o2[(size_t)e * p.ld_out2] = (half_t)v[e];
        if (tok == p.rows_per_batch - 1) {            // the row pad tok + 1 .. ld_out2 - 1 is part of the output: zero (consumers multiply it by P = 0)
          for (int t = 1; tok + t < p.ld_out2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) o2[(size_t)e * p.ld_out2 + t] = (half_t)0.f;
        }
        continue;
      }
      if (p.residual) {
        const half4_t rv = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * p.N + n0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
      }
      const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ld_out + n0) = h;
    }
  }
}

template <int TAPS, int NWM, int NWN, int TN, int EPI, int NST>
__global__ __launch_bounds__(64 * NWM * NWN, NWM * NWN == 4 ? 2 : 1) void af_gemm3_kernel(Gemm3Dev p) {
  constexpr int TM = 4;
  constexpr int NW = NWM * NWN;
  constexpr int BM = NWM * 64, BN = NWN * TN * 16;
  constexpr int APW = (BM / 16) / NW;                 // A pieces (16 rows) per wave per stage
  constexpr int WPW = (BN / 16 + NW - 1) / NW;        // W pieces per wave per stage
  constexpr int DPS = APW + WPW;                      // DMA pieces per wave per stage (3 or 4)
  static_assert(APW * NW * 16 == BM && DPS >= 3 && DPS <= 5, "every wave must issue DPS DMA pieces per stage");
  constexpr int WROWS = WPW * NW * 16;                // W rows held per stage (>= BN)
  constexpr int STAGE = (BM + WROWS) * 64;            // bytes
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % NWM, wn = wave / NWM;

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

  // ---- loader state: wave w fills A pieces {w*APW + j} and W pieces {w + NW*j}
  const int lrow = lane >> 2;                                     // row inside a 16-row piece
  const int lc = (lane & 3) ^ ((0x78 >> (2 * (lane >> 4))) & 3);  // logical chunk this lane fetches (source-side swizzle)
  const int Cin = p.c1 + p.c2;
  int a_base[APW];
  unsigned a_mask[APW];
  const half_t* wptr[WPW];
  bool wok[WPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int m = tile_m * BM + (wave * APW + j) * 16 + lrow;
    if (TAPS == 9) {
      unsigned mk = 0;
      int base = 0;
      if (m < p.M) {
        const int b = m / p.HoWo;
        const int rem = m - b * p.HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int cy = oy * p.stride, cx = ox * p.stride;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
          const int iy = cy + t9 / 3 - 1, ix = cx + t9 % 3 - 1;
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t9;
        }
        base = (b * p.H + cy) * p.W + cx;
      }
      a_mask[j] = mk;
      a_base[j] = base;
    } else {
      a_mask[j] = m < p.M ? 1u : 0u;
      a_base[j] = m;
    }
  }
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int piece = wave + NW * j;
    const int n = tile_n * BN + piece * 16 + lrow;
    wok[j] = piece * 16 < BN && n < p.npad;            // padding pieces / rows beyond the packed weight -> zero page
    wptr[j] = p.wt + (size_t)(wok[j] ? n : 0) * p.kpad + lc * 8;
  }

  auto issue_stage = [&](int kt, int slot) {
    char* As = af_smem + slot * STAGE;
    char* Ws = As + BM * 64;
    const int k0 = kt * BK3;
    if (TAPS == 9) {
      const int tp = k0 / Cin;                      // workgroup-uniform (32 | c1, c2)
      const int c0 = k0 - tp * Cin;
      const bool first = c0 < p.c1;
      const half_t* src = first ? p.a1 : p.a2;
      const int cs = first ? p.c1 : p.c2;
      const int coff = (first ? c0 : c0 - p.c1) + lc * 8;
      const int dpix = (tp / 3 - 1) * p.W + (tp % 3 - 1);
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const bool ok = (a_mask[j] >> tp) & 1u;     // tp >= 9 (K padding): no bit set
        const half_t* g = ok ? src + (size_t)(a_base[j] + dpix) * cs + coff : p.zeros;
        if (p.ablate & 64) g = p.zeros + (((size_t)g >> 4) & 7) * 8;       // timing experiments: every DMA reads the hot 128-byte zero page
        glds16(g, As + (wave * APW + j) * 1024);
      }
    } else {
      const bool first = k0 < p.c1;                 // uniform: 32 | c1
      const half_t* src = first ? p.a1 : p.a2;
      const int ld = first ? p.lda1 : p.lda2;
      const int koff = (first ? k0 : k0 - p.c1) + lc * 8;
      const bool kval = k0 + lc * 8 < p.K;
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const bool ok = kval && a_mask[j];
        const half_t* g = ok ? src + (size_t)a_base[j] * ld + koff : p.zeros;
        glds16(g, As + (wave * APW + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
      const half_t* g = wok[j] ? wptr[j] + k0 : p.zeros;
      if (p.ablate & 64) g = p.zeros + (((size_t)g >> 4) & 7) * 8;
      glds16(g, Ws + (wave + NW * j) * 1024);
    }
  };

  floatx4 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int rd_off = fr * 64 + ((fq ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) * 16);   // swizzled fragment offset inside a 16-row piece

  const int nk_total = p.kpad / BK3;
  const int kt_begin = blockIdx.y * p.kt_per_split;
  const int kt_end = min(nk_total, kt_begin + p.kt_per_split);
  int nk = kt_end - kt_begin;
  if (p.ablate & 8) nk = 0;                       // timing experiments: launch + setup + epilogue only

  half8_t wf[TN], xf[TM];
  // own pieces of a stage have landed when at most `younger` later stages' DMAs (DPS each, issued in order) are outstanding
  auto wait_own = [&](int younger) {
    if (younger >= 3) {
      if (DPS == 5) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      else if (DPS == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else if (younger == 2) {
      if (DPS == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (DPS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (younger == 1) {
      if (DPS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (DPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  auto read_frags = [&](int slot) {
    const char* As = af_smem + slot * STAGE;
    const char* Ws = As + BM * 64;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + (wn * TN * 16 + tn * 16) * 64 + rd_off);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(As + (wm * 64 + tm * 16) * 64 + rd_off);
  };
  auto mfma_step = [&]() {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
  };

  {
    // ---- lock-step schedule: wait -> barrier -> DMA issue -> fragment reads -> MFMAs per 32-wide K step (a ping-pong schedule of
    // the two waves per SIMD was measured slower, profiles/r01r_gemm_diagnosis.txt)
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) issue_stage(kt_begin + s, s);
    for (int i = 0; i < nk; ++i) {
      wait_own(min(nk - 1 - i, NST - 2));
      if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();   // every wave's pieces of stage i are in LDS; everyone is done reading slot (i-1)%NST
      if (i + NST - 1 < nk && !(p.ablate & 1)) issue_stage(kt_begin + i + NST - 1, (i + NST - 1) % NST);
      if (!(p.ablate & 2) || i == 0) read_frags(i % NST);
      mfma_step();
    }
  }

  if (p.ablate & 16) {                            // timing experiments: no epilogue (one store keeps the accumulators alive)
    float sacc = 0.f;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) sacc += acc[tn][tm][0] + acc[tn][tm][1] + acc[tn][tm][2] + acc[tn][tm][3];
    if (sacc == 12345.678f) p.out[0] = (half_t)sacc;
    return;
  }
  gemm3_epilogue<EPI, NWM, NWN, TN, NST * STAGE>(p, acc, af_smem, tile_m, tile_n, wm, wn, fr, fq, tid);
}

template <int TAPS, int NWM, int NWN, int TN, int EPI = E3_STD, int NST = NST_DEFAULT>
bool launch3(const Gemm3Dev& p0, hipStream_t stream) {
  Gemm3Dev p = p0;
  constexpr int NW = NWM * NWN, BM = NWM * 64, BN = NWN * TN * 16;
  constexpr int WROWS = ((BN / 16 + NW - 1) / NW) * NW * 16;
  constexpr size_t lds = (size_t)NST * (BM + WROWS) * 64;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_m = tiles_m;
  if (p.counters && (TN > 5 || p.splits <= 1 || p.splits > 4 || tiles_m * p.tiles_n > AF_SPLITK_MAX_TILES)) p.counters = nullptr;
  p.n_major = af_gemm_n_major(p.M, p.N, p.K, TAPS == 9 ? p.c1 + p.c2 : p.K);
  static bool attr_set = false;
  const bool lds_ok = af_allow_dyn_lds(reinterpret_cast<const void*>(&af_gemm3_kernel<TAPS, NWM, NWN, TN, EPI, NST>), lds, attr_set, "af_gemm");
  dim3 grid(tiles_m * p.tiles_n, p.splits), block(64 * NW);
  if (lds_ok) hipLaunchKernelGGL((af_gemm3_kernel<TAPS, NWM, NWN, TN, EPI, NST>), grid, block, lds, stream, p);
  return p.counters != nullptr;      // true: the kernel reduced the K-slices itself
}

// ---------------------------------------------------------------------------------------------------------------------
// Whole-line variant (tile 7): 128 x 320 tile, 8 waves, 64-wide K stages whose rows are 128 bytes, so that every LDS-DMA wave
// instruction fetches 8 rows x one full 128-byte cache line (the 32-wide ring fetches 16 rows x half a line: twice the L2 requests,
// profiles/r01r_gemm_diagnosis.txt) and the wait -> barrier -> issue chain runs once per 64 K instead of once per 32.  Two 56 KB
// slots: stage i+1 streams in while stage i is computed.  56 pieces per stage = exactly 7 per wave (2 A + 5 W).  Fragment rows are
// 128 bytes apart, so 16 consecutive lanes alias two rows per 256 bytes of banks: the 16-byte chunk index is XOR-ed with
// (row >> 1) & 7 on the DMA source side.  Standard epilogue only (staged); channel counts and K padding multiples of 64.
template <int TAPS, int NWM, int NWN, int TN, int EPI, int NSLOT = 2>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN == 4 && NSLOT == 2) ? 2 : 1) void af_gemm3w_kernel(Gemm3Dev p) {
  constexpr int TM = 4, NW = NWM * NWN;
  constexpr int BM = NWM * 64, BN = NWN * TN * 16, BKW = 64;
  constexpr int APW = (BM / 8) / NW;                  // A pieces (8 rows x 128 B) per wave per stage
  constexpr int WPW = (BN / 8) / NW;                  // W pieces
  static_assert(APW * NW * 8 == BM && WPW * NW * 8 == BN, "pieces must divide evenly among the waves");
  constexpr int STAGE = (BM + BN) * 128;              // bytes (57344 for 128 x 320)
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % NWM, wn = wave / NWM;

  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.n_major) {
      tile_n = lid / p.tiles_m;
      tile_m = lid - tile_n * p.tiles_m;
    } else {
      tile_m = lid / p.tiles_n;
      tile_n = lid - tile_m * p.tiles_n;
    }
  }

  // ---- loader state: wave w fills A pieces {w*APW + j} and W pieces {w*WPW + j}; lane = (row in piece, chunk slot)
  const int prow = lane >> 3, slot = lane & 7;
  const int Cin = p.c1 + p.c2;
  int a_base[APW], a_lc[APW], a_par[APW];
  unsigned a_mask[APW];
  const half_t* wptr[WPW];
  bool wok[WPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int row = (wave * APW + j) * 8 + prow;              // row inside the tile
    a_lc[j] = slot ^ ((row >> 1) & 7);                       // logical chunk this lane fetches
    const int m = tile_m * BM + row;
    if (TAPS == 9) {
      unsigned mk = 0;
      int base = 0;
      int par = 0;
      if (m < p.M) {
        const int b = m / p.HoWo;
        const int rem = m - b * p.HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int cy = oy * p.stride, cx = ox * p.stride;
        const int He = p.upsample ? 2 * p.H : p.H, We = p.upsample ? 2 * p.W : p.W;     // the grid the taps walk on
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
          const int iy = cy + t9 / 3 - 1, ix = cx + t9 % 3 - 1;
          if ((unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We) mk |= 1u << t9;
        }
        if (p.upsample) {               // source pixel of tap (ty, tx): ((oy + ty - 1) >> 1, (ox + tx - 1) >> 1)
          base = (b * p.H + (oy >> 1)) * p.W + (ox >> 1);
          par = (oy & 1) | ((ox & 1) << 1);
        } else {
          base = (b * p.H + cy) * p.W + cx;
        }
      }
      a_mask[j] = mk;
      a_base[j] = base;
      a_par[j] = par;
    } else {
      a_mask[j] = m < p.M ? 1u : 0u;
      a_base[j] = m;
      a_par[j] = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int row = (wave * WPW + j) * 8 + prow;
    const int n = tile_n * BN + row;
    wok[j] = n < p.npad;
    wptr[j] = p.wt + (size_t)(wok[j] ? n : 0) * p.kpad + (slot ^ ((row >> 1) & 7)) * 8;
  }
  // Round 6 (vector-instruction diet, see af_conv3hd_kernel): everything per-lane about a piece's address is computed ONCE --
  //   * weight pieces and plain-row A pieces (taps 1, the K tail) use the scalar-base LDS-DMA form: a 32-bit per-lane byte offset here + a scalar
  //     pointer per stage.  Rows past M / past the packed weight are CLAMPED to a valid row instead of being pointed at the zero page: their
  //     products only reach output rows / columns the epilogue drops (m >= M, n >= N);
  //   * 3x3 taps keep a per-lane 64-bit pixel pointer per source; a stage adds the tap's scalar offset and selects the zero page for taps that
  //     fall outside the image (the nearest-x2 gather keeps the per-stage arithmetic: its tap offset depends on the lane's pixel parity).
  // Byte offsets are 32-bit: the host routes tensors of 2 GB and more to the register-staged kernel.
  unsigned w_off[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int row = (wave * WPW + j) * 8 + prow;
    const int n = min(tile_n * BN + row, p.npad - 1);
    w_off[j] = ((unsigned)n * (unsigned)p.kpad + (unsigned)((slot ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
  unsigned a_off1[APW], a_off2[APW];                  // plain rows: taps 1 (sources 1 / 2) or the K tail of a 3x3 launch (a3 / a4)
  const half_t* a_px1[APW];                           // 3x3: the centre pixel's chunk in source 1 / 2
  const half_t* a_px2[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int row = (wave * APW + j) * 8 + prow;
    const int mc = min(tile_m * BM + row, p.M - 1);
    const unsigned ch = (unsigned)(a_lc[j] * 8);
    if (TAPS == 9) {
      a_off1[j] = ((unsigned)mc * (unsigned)p.lda3 + ch) * 2u;
      a_off2[j] = ((unsigned)mc * (unsigned)p.lda4 + ch) * 2u;
      a_px1[j] = p.a1 + (size_t)a_base[j] * p.c1 + ch;
      a_px2[j] = p.a2 + (size_t)a_base[j] * p.c2 + ch;
    } else {
      a_off1[j] = ((unsigned)mc * (unsigned)p.lda1 + ch) * 2u;
      a_off2[j] = ((unsigned)mc * (unsigned)p.lda2 + ch) * 2u;
      a_px1[j] = a_px2[j] = nullptr;
    }
  }
  // A/B arms: 3x3 launches keep the round-5 per-stage arithmetic behind AF_GEMM3_ABLATE bit 512 (their nearest-x2 path needs its state anyway);
  // plain-row launches have it behind the compile-time AF_GEMM3W_DIET=0 only (a runtime switch would keep both sets of loader registers alive:
  // the 256 x 320 GEGLU tile then spills)
  const bool diet = TAPS == 9 ? (p.ablate & 512) == 0 : (AF_GEMM3W_DIET != 0);

  auto issue_stage = [&](int kt, int sl) {
    char* As = af_smem + sl * STAGE;
    char* Ws = As + BM * 128;
    const int k0 = kt * BKW;
    if (diet && !(TAPS == 9 && p.upsample)) {
      if (TAPS == 9 && p.c3 > 0 && k0 >= 9 * Cin) {
        const int kt0 = k0 - 9 * Cin;                // the K tail: plain rows of a3 | a4 (64 | c3, c4: no K padding behind it)
        const bool first = kt0 < p.c3;
        const half_t* sb = first ? p.a3 + kt0 : p.a4 + (kt0 - p.c3);
        // (a scalar branch, not `first ? a_off1[j] : a_off2[j]`: hipcc turns a select between two register arrays into a load from a selected
        // ADDRESS, i.e. both arrays go to scratch)
        if (first) {
#pragma unroll
          for (int j = 0; j < APW; ++j) glds16_sbase(sb, a_off1[j], As + (wave * APW + j) * 1024);
        } else {
#pragma unroll
          for (int j = 0; j < APW; ++j) glds16_sbase(sb, a_off2[j], As + (wave * APW + j) * 1024);
        }
      } else if (TAPS == 9) {
        const int tp = k0 / Cin;                     // workgroup-uniform (64 | c1, c2)
        const int c0 = k0 - tp * Cin;
        const bool first = c0 < p.c1;
        const int cs = first ? p.c1 : p.c2;
        const int ty = tp / 3, tx = tp - ty * 3;
        const long soff = (long)((ty - 1) * p.W + (tx - 1)) * cs + (first ? c0 : c0 - p.c1);    // elements from the centre pixel's chunk: scalar
#pragma unroll
        for (int j = 0; j < APW; ++j) {
          const bool ok = (a_mask[j] >> tp) & 1u;
          const half_t* px = a_px1[j];
          if (!first) px = a_px2[j];
          const half_t* g = ok ? px + soff : p.zeros;
          glds16(g, As + (wave * APW + j) * 1024);
        }
      } else {
        const bool first = k0 < p.c1;                // uniform: 64 | c1
        const half_t* sb = first ? p.a1 + k0 : p.a2 + (k0 - p.c1);
        if (first) {
#pragma unroll
          for (int j = 0; j < APW; ++j) glds16_sbase(sb, a_off1[j], As + (wave * APW + j) * 1024);
        } else {
#pragma unroll
          for (int j = 0; j < APW; ++j) glds16_sbase(sb, a_off2[j], As + (wave * APW + j) * 1024);
        }
      }
      const half_t* wb = p.wt + k0;
#pragma unroll
      for (int j = 0; j < WPW; ++j) glds16_sbase(wb, w_off[j], Ws + (wave * WPW + j) * 1024);
      return;
    }
    if (TAPS == 9 && p.c3 > 0 && k0 >= 9 * Cin) {
      // K tail: a 1x1 convolution of a second image on the output grid (stride 1: the tile row's pixel index IS its output row), c3 | c4 columns
      const int kt0 = k0 - 9 * Cin;
      const bool first = kt0 < p.c3;                // uniform: 64 | c3
      const half_t* src = first ? p.a3 : p.a4;
      const int ld = first ? p.lda3 : p.lda4;
      const int koff = first ? kt0 : kt0 - p.c3;
      const bool inside = kt0 < p.c3 + p.c4;        // K padding behind the tail: zeros
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const bool ok = inside && ((a_mask[j] >> 4) & 1u);          // the centre tap is valid exactly for the rows m < M
        const half_t* g = ok ? src + (size_t)a_base[j] * ld + koff + a_lc[j] * 8 : p.zeros;
        glds16(g, As + (wave * APW + j) * 1024);
      }
    } else if (TAPS == 9) {
      const int tp = k0 / Cin;                      // workgroup-uniform (64 | c1, c2)
      const int c0 = k0 - tp * Cin;
      const bool first = c0 < p.c1;
      const half_t* src = first ? p.a1 : p.a2;
      const int cs = first ? p.c1 : p.c2;
      const int coff = first ? c0 : c0 - p.c1;
      const int ty = tp / 3, tx = tp - ty * 3;
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const bool ok = (a_mask[j] >> tp) & 1u;     // tp >= 9 (K padding): no bit set
        const int dpix = p.upsample ? (((a_par[j] & 1) + ty - 1) >> 1) * p.W + (((a_par[j] >> 1) + tx - 1) >> 1)
                                    : (ty - 1) * p.W + (tx - 1);
        const half_t* g = ok ? src + (size_t)(a_base[j] + dpix) * cs + coff + a_lc[j] * 8 : p.zeros;
        glds16(g, As + (wave * APW + j) * 1024);
      }
    } else {
      const bool first = k0 < p.c1;                 // uniform: 64 | c1
      const half_t* src = first ? p.a1 : p.a2;
      const int ld = first ? p.lda1 : p.lda2;
      const int koff = first ? k0 : k0 - p.c1;
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const bool ok = a_mask[j] && (k0 + a_lc[j] * 8 < p.K);
        const half_t* g = ok ? src + (size_t)a_base[j] * ld + koff + a_lc[j] * 8 : p.zeros;
        glds16(g, As + (wave * APW + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < WPW; ++j) glds16(wok[j] ? wptr[j] + k0 : p.zeros, Ws + (wave * WPW + j) * 1024);
  };

  floatx4 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // fragment offsets inside a 16-row group (rows are group-aligned, so (row >> 1) & 7 == fr >> 1): K half kk, chunk kk*4 + fq
  const int rd0 = fr * 128 + (((0 * 4 + fq) ^ (fr >> 1)) * 16);
  const int rd1 = fr * 128 + (((1 * 4 + fq) ^ (fr >> 1)) * 16);

  const int nk_total = p.kpad / BKW;
  const int kt_begin = blockIdx.y * p.kt_per_split;
  const int kt_end = min(nk_total, kt_begin + p.kt_per_split);
  const int nk = kt_end - kt_begin;

  // weight-tile prefetch into this XCD's L2 (af_common.h): the tile's weight rows over this K split, shared out among the
  // workgroups with the same tile_n
  if (p.wpf > 0)
    af_prefetch_weight_tile(p.wt, p.kpad, p.npad, tile_n * BN, BN, kt_begin, nk, p.wpf_coop, tile_m % p.wpf_coop, p.wpf, NW, wave, lane,
                            af_smem + NSLOT * STAGE + BM * 8 + BN * 4);

  // NSLOT - 1 stages are in flight ahead of the one being computed (NSLOT = 2: the shipped tiles; NSLOT = 4: the deep-ring variants for
  // grids of at most one workgroup per CU, where nothing else hides the L2 / HBM latency of a stage -- the 16x16 level's GEMMs)
#pragma unroll
  for (int s0 = 0; s0 < NSLOT - 1; ++s0)
    if (s0 < nk) issue_stage(kt_begin + s0, s0);
  constexpr int DPS = APW + WPW;                          // LDS-DMA pieces per wave per stage
  half8_t wf[TN], xf[TM];
  // 8-wave tiles put two waves on every SIMD, and with one barrier per K step the partners walk the step in lock step: both issue
  // their DMA pieces (~60 - 100 cycles each, 7 per wave) at the same time while the MFMA pipe idles.  The second half of the waves
  // (the SIMD partners of the first half) therefore issues the next stage's DMA BETWEEN the two K halves, under the partner's
  // MFMAs, and each MFMA cluster runs at raised priority (the pair keeps the compiler from moving MFMAs across the barrier).
  // Measured in one process (profiles/r02c_gemm_variants.txt): conv 8x64x64 320->320 69.1 -> 64.1 us, 640->320 146.1 -> 125.1,
  // GEGLU M32768 N2560 K320 112.0 -> 101.2, M32768 N320 K1280 34.5 -> 32.4; the 4-wave tile (partners belong to different
  // workgroups, not in lock step) gets 1 - 3 % slower with it, so it keeps the plain order.
  // AF_GEMM3_ABLATE bits 128 / 256 switch the priority pair / the late DMA OFF (A/B runs).  Tried on top and dropped
  // (profiles/r02e_gemm_skew.txt): skewing the partner waves by one K half (their second-half MFMAs run after the barrier, under
  // the first half's DMA issue and reads) -- +86 VGPRs, faster on one shape, 5 - 17 % slower on four.
  const bool prio = NW == 8 && (p.ablate & 128) == 0;
  const bool late_dma = NW == 8 && (p.ablate & 256) == 0 && wave >= NW / 2;
  // LayerNorm folded into the GEMM: the rows' sum / sum of squares come from the A fragments the MFMAs consume anyway (lane (fr, fq) holds
  // row fr, K chunk fq of every 32-wide K step), two v_dot2_f32_f16 per register pair; the NWN waves that share a row group split its TM
  // fragments among themselves (wave wn takes tm % NWN == wn), so the cost is 16 VALU per wave per 64-wide stage at NWN = 4.
  constexpr int LNT = TM / NWN;
  static_assert(LNT * NWN == TM, "the row groups of a tile divide evenly among the waves that share them");
  float ln_s[LNT], ln_q[LNT];
#pragma unroll
  for (int j = 0; j < LNT; ++j) ln_s[j] = ln_q[j] = 0.f;
  // a RUNTIME flag on purpose: as a template parameter (straight-line statistics code the scheduler may spread through the MFMA clusters)
  // the GEGLU tiles ran 16 - 22 us slower, profiles/r03e_lnfold.txt; the branch keeps the statistics one block per stage
  const bool ln_on = p.ln_cs != nullptr;
  // the tile's column sums: fetched now (their L2 latency passes under the main loop), parked in LDS for the epilogue afterwards
  floatx4 ln_csv = {0.f, 0.f, 0.f, 0.f};
  if (ln_on && tid < BN / 4) ln_csv = *reinterpret_cast<const floatx4*>(p.ln_cs + tile_n * BN + tid * 4);   // packed rows are padded to 128: in range
  for (int i = 0; i < nk; ++i) {
    if (NSLOT == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage i (the only one in flight) has landed
    } else {                                                // stage i has landed when at most the younger stages' pieces are outstanding
      const int younger = min(NSLOT - 2, nk - 1 - i);
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                           // ... for every wave; everyone is done reading the slot refilled next
    if (!late_dma && i + NSLOT - 1 < nk) issue_stage(kt_begin + i + NSLOT - 1, (i + NSLOT - 1) % NSLOT);
    const char* As = af_smem + (i % NSLOT) * STAGE;
    const char* Ws = As + BM * 128;
    if (ln_on) {
      // one block per stage, in front of the MFMA clusters (whose straight-line schedule stays as it is): this wave's share of the row groups is read from LDS once more -- a runtime-indexed register fragment would need a branch per
      // fragment
      const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
      for (int j = 0; j < LNT; ++j) {
        const char* row = As + (wm * 64 + (j * NWN + wn) * 16) * 128;
        const half8_t f0 = *reinterpret_cast<const half8_t*>(row + rd0), f1 = *reinterpret_cast<const half8_t*>(row + rd1);
        float s0 = ln_s[j], q0 = ln_q[j], s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const half2_t a = {f0[e], f0[e + 1]}, b = {f1[e], f1[e + 1]};
          s0 = __builtin_amdgcn_fdot2(a, one2, s0, false);
          q0 = __builtin_amdgcn_fdot2(a, a, q0, false);
          s1 = __builtin_amdgcn_fdot2(b, one2, s1, false);
          q1 = __builtin_amdgcn_fdot2(b, b, q1, false);
        }
        ln_s[j] = s0 + s1;
        ln_q[j] = q0 + q1;
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int rd = kk ? rd1 : rd0;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + (wn * TN * 16 + tn * 16) * 128 + rd);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(As + (wm * 64 + tm * 16) * 128 + rd);
      if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
      if (prio) __builtin_amdgcn_s_setprio(0);
      if (kk == 0 && late_dma && i + NSLOT - 1 < nk) issue_stage(kt_begin + i + NSLOT - 1, (i + NSLOT - 1) % NSLOT);
    }
  }
  const float* lnst = nullptr;
  if (ln_on) {
    // row statistics -> LDS behind the ring (BM x (mean, rstd)); every wave of the row group reads them in the epilogue
    float* st = reinterpret_cast<float*>(af_smem + NSLOT * STAGE);
    const float inv_k = 1.0f / (float)p.K;
#pragma unroll
    for (int j = 0; j < LNT; ++j) {
      float s = ln_s[j], q = ln_q[j];
      s += __shfl_xor(s, 16, 64);
      q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (fq == 0) {
        const float mean = s * inv_k;
        const float var = fmaxf(q * inv_k - mean * mean, 0.f);
        const int row = wm * 64 + (j * NWN + wn) * 16 + fr;
        st[2 * row] = mean;
        st[2 * row + 1] = rsqrtf(var + p.ln_eps);
      }
    }
    if (tid < BN / 4) *reinterpret_cast<floatx4*>(st + 2 * BM + tid * 4) = ln_csv;
    __syncthreads();
    lnst = st;
  }
  gemm3_epilogue<EPI, NWM, NWN, TN, NSLOT * STAGE>(p, acc, af_smem, tile_m, tile_n, wm, wn, fr, fq, tid, lnst);
}

template <int TAPS, int NWM, int NWN, int TN, int EPI = E3_STD, int NSLOT = 2>
bool launch3w(const Gemm3Dev& p0, hipStream_t stream) {
  Gemm3Dev p = p0;
  constexpr int NW = NWM * NWN, BM = NWM * 64, BN = NWN * TN * 16;
  constexpr size_t lds = NSLOT * (size_t)(BM + BN) * 128 + (size_t)BM * 8 + (size_t)BN * 4 + AF_WPF_DUMP_BYTES;   // ring + the folded LayerNorm's row statistics and column sums + the prefetch dump
  static_assert(NSLOT == 2 || NSLOT == 4, "ring depths built: 2 and 4 (the counted waits cover at most two younger stages)");
  p.tiles_n = (p.N + BN - 1) / BN;
  p.tiles_m = (p.M + BM - 1) / BM;
  {
    static const int coop_env = getenv("AF_GEMM3_WPF_COOP") ? atoi(getenv("AF_GEMM3_WPF_COOP")) : 32;
    p.wpf_coop = p.tiles_m < coop_env ? p.tiles_m : coop_env;
    if (p.wpf > AF_WPF_MAX) p.wpf = AF_WPF_MAX;
  }
  if (p.counters && (TN > 5 || p.splits <= 1 || p.splits > 4 || p.tiles_m * p.tiles_n > AF_SPLITK_MAX_TILES)) p.counters = nullptr;
  p.n_major = af_gemm_n_major(p.M, p.N, p.K, TAPS == 9 ? p.c1 + p.c2 : p.K);
  static bool attr_set = false;
  const bool lds_ok = af_allow_dyn_lds(reinterpret_cast<const void*>(&af_gemm3w_kernel<TAPS, NWM, NWN, TN, EPI, NSLOT>), lds, attr_set, "af_gemm");
  dim3 grid(p.tiles_m * p.tiles_n, p.splits), block(64 * NW);
  if (lds_ok) hipLaunchKernelGGL((af_gemm3w_kernel<TAPS, NWM, NWN, TN, EPI, NSLOT>), grid, block, lds, stream, p);
  return p.counters != nullptr;
}


// ---------------------------------------------------------------------------------------------------------------------
// Halo-resident 3x3 convolution ("tile 14"): stride 1, pad 1, output width in {8, 16, 32, 64} (a tile = 256 pixels = whole rows of one image or whole images), one or two channel-concatenated sources, optionally on the
// nearest-x2 upsampled input (the halo pixel (y, x) of the 2H x 2W grid is source pixel (y >> 1, x >> 1)) (no K tail:
// a one-stage chunk would have to bring a whole halo in under ONE stage -- measured 15 - 80 % slower than the tap-by-tap tile, r05r).
// The tap-by-tap implicit GEMM above fetches a tile's activation rows NINE times per 64 input channels (once per tap, shifted), and with the
// 128 x 320 tile every 64-wide K stage moves 16 KB of activations + 40 KB of weights from L2 into LDS for 0.7 us of MFMA work -- the L2 -> LDS
// path is what the kernel waits for (profiles/r01r_gemm_diagnosis.txt: 20 of 89 us with real addresses).  Here a workgroup owns R = 256 / W
// WHOLE image rows (256 output pixels) x 160 output channels, and for each 64-channel chunk of the input:
//   * the (R + 2) x (W + 2) halo of those rows is brought into LDS ONCE (<= 396 pixels x 128 B = 50 KB; out-of-image pixels come from the
//     zero page), double-buffered: the next chunk's halo streams in, one 1-KB piece per wave per stage, under this chunk's nine taps;
//   * the nine taps are nine K stages over the SAME halo: tap (ky, kx) of output pixel (r, c) is halo pixel (r + ky, c + kx), i.e. the
//     MFMA fragment address plus a workgroup-uniform offset; only the tap's 160 x 64 weight tile (20 KB) is new per stage.
// Per stage 26 KB instead of 56 KB cross from L2 for the same 40 MFMAs per wave, and 3.2 instead of 7 LDS-DMA instructions per wave.
// LDS rows are 128 B with the 16-byte chunk index XOR-ed with pixel & 7 on the DMA source side.  (Rounds 3 - 5 used (pixel >> 1) & 7, which puts 16
// consecutive pixels on 16 distinct (bank half, chunk) slots -- conflict-free if a ds_read_b128 were served in groups of 16 CONSECUTIVE lanes.  It is
// served in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), and under those the old swizzle is two-way
// conflicted for every fragment start that is not a multiple of 4 pixels, i.e. for three of four (row, tap) shifts: SQ_LDS_BANK_CONFLICT 0.24 of the
// kernel's LDS cycles, profiles/r05ac.  pixel & 7 is conflict-free at every start under the real grouping -- checked exhaustively in
// tools/lds_swizzle_check.py, then with the counter.)  K is walked chunk-major (k = tap * Cin + chunk * 64 in the packed weight; Cin = c1 + c2, a chunk lies in ONE source);
// split-K slices are chunk ranges.  The tile is 256 consecutive output rows of the [M, N] result, so the staged epilogue is the shared one.
//
// Main loop (round 5): PING-PONG between the two waves of a SIMD.  The eight waves are two groups of four, one wave of each group per SIMD
// (waves w and w + 4 share one).  In the lock-step form of rounds 3 - 4 (every wave: wait, barrier, DMA, 18 fragment reads, 40 MFMAs) both
// waves of a SIMD want the matrix pipe right after the barrier and neither does before it (pipe busy 0.34, profiles/r05b_gemm_pmc.txt).
// Here a group runs the 40 MFMAs of a stage while the other reads ITS 18 fragments of the same stage and issues its LDS-DMA pieces (~100
// cycles each, during which an in-order wave issues nothing else), and the pipe is handed over at every barrier:
//     interval 2s    : group 0  L(s)   = fragment reads, its pieces of stage s + 1      group 1  M(s-1) = 40 MFMAs
//     interval 2s + 1: group 0  M(s)   = 40 MFMAs                                        group 1  L(s)   = fragment reads, its pieces of stage s + 2
// one barrier per interval, THREE weight slots (stage s in slot s % 3; 2 x 50 KB halo + 3 x 20 KB = the CU's 160 KB; the weight-prefetch
// dump aliases the second halo buffer, which nothing fills before the first barrier).  Write-after-read: slot (s + 2) % 3 held stage s - 1,
// whose last readers (group 1, interval 2s - 1) retired their reads (lgkmcnt(0)) in front of the barrier ending that interval.  Read-after-
// write: a wave retires its pieces (vmcnt(0)) at the end of the M part that follows their issue, i.e. in front of a barrier that precedes the
// first read of that stage by at least one more barrier (cdna guide: read a staged buffer one phase after the wait that retires it).
// Same accumulation order as the lock-step form: bit-identical results, 6 - 12 % faster per launch, 12 - 20 % against the tap-by-tap 128 x 320
// tile (profiles/r05r_conv3h_pingpong.txt; in the step: r05u).
constexpr int CH_BM = 256, CH_BN = 160, CH_NP_MAX = 50;          // output pixels, output channels, 8-pixel DMA pieces of the largest halo
constexpr int CH_ASZ = CH_NP_MAX * 1024, CH_WSZ = CH_BN * 128;   // one halo buffer, one weight stage
constexpr int CH_LDS = 2 * CH_ASZ + 3 * CH_WSZ;                  // 163,840 B = 160 KB

template <int ABL>                                                // ABL: profiling only (AF_GEMM3_ABLATE, compile-time): 1 no DMA in the loop, 2 fragments read once, 4 no barriers, 16 no MFMAs
__global__ __launch_bounds__(512) void af_conv3h_kernel(const Gemm3Dev p) {
  constexpr int NWM = 4, NWN = 2, TN = 5, TM = 4, NW = 8, BM = CH_BM, BN = CH_BN;
  constexpr int APW = (CH_NP_MAX + NW - 1) / NW;                 // 7 halo pieces per wave at most
  constexpr int WPW = (BN / 8 + NW - 1) / NW;                    // 3 weight pieces per wave at most (20 pieces)
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % NWM, wn = wave / NWM;
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.n_major) {
      tile_n = lid / p.tiles_m;
      tile_m = lid - tile_n * p.tiles_m;
    } else {
      tile_m = lid / p.tiles_n;
      tile_n = lid - tile_m * p.tiles_n;
    }
  }
  const int Wd = p.Wo, Hd = p.Ho, Wh = Wd + 2, R = BM / Wd;           // the grid the taps walk on (nearest x2 folded into the gather: Ho = 2 H, Wo = 2 W)
  // A tile is R whole rows of ONE image, or -- where an image has fewer than R rows (the 8 x 8 level: R = 32) -- R / Ho whole images, each with its own
  // halo block of (Hi + 2) x (W + 2) pixels (the tap offset ty * (W + 2) + tx stays workgroup-uniform)
  const int Hi = min(R, Hd), nimg = R / Hi, blk_px = (Hi + 2) * Wh;
  const int halo_px = nimg * blk_px, NP = (halo_px + 7) >> 3;
  const int m0 = tile_m * BM;
  const int bimg = m0 / p.HoWo;
  const int y0 = (m0 - bimg * p.HoWo) / Wd;
  const int Cin = p.c1 + p.c2;
  const int prow = lane >> 3, slot = lane & 7;

  // ---- the chunk list: the 64-channel chunks of the (concatenated) input, nine taps (= nine stages) each; split-K slices are chunk ranges
  const int nmain = Cin >> 6, nc1 = p.c1 >> 6;
  const int cb = blockIdx.y * p.kt_per_split;
  const int ce = min(nmain, cb + p.kt_per_split);
  const int nst = (ABL & 8) ? 0 : (ce - cb) * 9;
  struct Pos { int u, tap; };                                        // a stage = (chunk, tap)
  auto next_pos = [&](Pos q) { return q.tap < 8 ? Pos{q.u, q.tap + 1} : Pos{q.u + 1, 0}; };
  auto k0_of = [&](Pos q) { return q.tap * Cin + q.u * 64; };         // first weight column of the stage

  // ---- loaders: wave w owns halo pieces {w + 8 j} and weight pieces {w + 8 j}; lane = (pixel / row in the piece, physical chunk)
  int a_pix[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int hp = (wave + NW * j) * 8 + prow;
    const int im = hp / blk_px, hq = hp - im * blk_px;                                             // image of the tile, pixel of its halo block
    const int hy = hq / Wh, hx = hq - hy * Wh;
    const int iy = y0 - 1 + hy, ix = hx - 1;
    const bool ok = hp < halo_px && (unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd;
    const int sy = p.upsample ? iy >> 1 : iy, sx = p.upsample ? ix >> 1 : ix;                       // source pixel of grid pixel (iy, ix)
    a_pix[j] = ok ? ((((bimg + im) * p.H + sy) * p.W + sx) << 3) | (slot ^ (hp & 7)) : -1;         // pixel index and the lane's logical 16-byte chunk
  }
  const half_t* wptr[WPW];
  bool wok[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int row = (wave + NW * j) * 8 + prow;
    const int n = tile_n * BN + row;
    wok[j] = row < BN && n < p.npad;
    wptr[j] = p.wt + (size_t)(wok[j] ? n : 0) * p.kpad + (slot ^ ((row >> 1) & 7)) * 8;
  }
  // the kernel arguments the loops' loaders use, as opaque register values: left as kernarg loads, hipcc folds `first ? p.c1 : p.c2` into a scalar
  // load from a selected ADDRESS inside the loop, and every scalar load is waited for with lgkmcnt(0) -- i.e. behind the 18 fragment reads
  const half_t *a1r = p.a1, *a2r = p.a2, *zr = p.zeros;
  int c1r = p.c1, c2r = p.c2;
  asm volatile("" : "+s"(a1r), "+s"(a2r), "+s"(zr), "+s"(c1r), "+s"(c2r));
  auto issue_halo_piece = [&](int j, const half_t* src, int ld, int buf) {      // piece j of this wave; src = the chunk's first channel in its source
    if (wave + NW * j < NP) {
      const half_t* g = a_pix[j] >= 0 ? src + (size_t)(a_pix[j] >> 3) * ld + (a_pix[j] & 7) * 8 : zr;
      glds16_nowait(g, af_smem + buf * CH_ASZ + (wave + NW * j) * 1024);
    }
  };
  auto chunk_src = [&](int u, const half_t*& src, int& ld) {          // workgroup-uniform: a chunk lies in one source
    const bool first = u < nc1;
    src = first ? a1r + u * 64 : a2r + (u - nc1) * 64;
    ld = first ? c1r : c2r;
  };
  auto wslot = [&](int st) { return af_smem + 2 * CH_ASZ + (st % 3) * CH_WSZ; };
  auto issue_w = [&](int st, Pos q) {                                // this wave's weight pieces of stage st (at position q)
    if constexpr ((ABL & 1) != 0) return;
    const int k0 = k0_of(q);
    char* Ws = wslot(st);
#pragma unroll
    for (int j = 0; j < WPW; ++j)
      if (wave + NW * j < BN / 8) glds16_nowait(wok[j] ? wptr[j] + k0 : zr, Ws + (wave + NW * j) * 1024);
  };
  auto issue_halo = [&](Pos q) {                                     // the next chunk's halo: piece j under tap j
    if (q.u + 1 < ce && q.tap < APW && !(ABL & 1)) {
      const half_t* src;
      int ld;
      chunk_src(q.u + 1, src, ld);
#pragma unroll
      for (int j = 0; j < APW; ++j)
        if (q.tap == j) issue_halo_piece(j, src, ld, (q.u + 1 - cb) & 1);
    }
  };

  floatx4 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int rd0 = fr * 128 + (((0 * 4 + fq) ^ (fr >> 1)) * 16);      // weight fragments: 16-row groups are aligned
  const int rd1 = fr * 128 + (((1 * 4 + fq) ^ (fr >> 1)) * 16);
  int hpb[TM];                                                       // halo pixel of this lane's row for tap (0, 0), per 16-row group
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int q = wm * 64 + tm * 16 + fr;                            // this lane's output pixel inside the tile (16 | W except at the 8 x 8 level: per lane)
    const int r = q / Wd, im = r / Hi;
    hpb[tm] = im * blk_px + (r - im * Hi) * Wh + (q - r * Wd);
  }

  if (p.wpf > 0 && p.splits == 1)
    af_prefetch_weight_tile(p.wt, p.kpad, p.npad, tile_n * BN, BN, 0, p.kpad >> 6, p.wpf_coop, tile_m % p.wpf_coop, p.wpf, NW, wave, lane, af_smem + CH_ASZ);

  const int grp = wave >> 2;
  half8_t wf[TN], xf[TM], wf1[TN], xf1[TM];
  auto read_frags = [&](int st, Pos q) {
    if ((ABL & 2) && st > 0) return;
    const char* As = af_smem + ((q.u - cb) & 1) * CH_ASZ;
    const char* Ws = wslot(st);
    const int ty = (q.tap * 11) >> 5;                                // tap / 3 for 0 .. 8 (a compare chain becomes a constant-memory table whose s_load waits on every ds_read)
    const int toff = ty * Wh + (q.tap - ty * 3);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + (wn * TN * 16 + tn * 16) * 128 + rd0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int hp = hpb[tm] + toff;
      xf[tm] = *reinterpret_cast<const half8_t*>(As + hp * 128 + (((0 * 4 + fq) ^ (hp & 7)) * 16));
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) wf1[tn] = *reinterpret_cast<const half8_t*>(Ws + (wn * TN * 16 + tn * 16) * 128 + rd1);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int hp = hpb[tm] + toff;
      xf1[tm] = *reinterpret_cast<const half8_t*>(As + hp * 128 + (((1 * 4 + fq) ^ (hp & 7)) * 16));
    }
  };
  auto mfmas = [&]() {
    if constexpr ((ABL & 16) != 0) return;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[tn], xf1[tm], acc[tn][tm], 0, 0, 0);
  };
  auto barrier = [&]() {
    if constexpr ((ABL & 4) == 0) __builtin_amdgcn_s_barrier();
  };

  __builtin_amdgcn_s_setprio(1);                                     // every wave runs at priority 1 except inside its own MFMA burst: the loaders' address
                                                                     // arithmetic is not starved by the partner's MFMA issue (profiles/r05r: 1 - 2 %)
  Pos q0 = Pos{cb, 0}, q1 = next_pos(q0), q2 = next_pos(q1);         // positions of stages st, st + 1, st + 2
  if (nst > 0) {
    const half_t* src;
    int ld;
    chunk_src(cb, src, ld);
#pragma unroll
    for (int j = 0; j < APW; ++j) issue_halo_piece(j, src, ld, 0);
    issue_w(0, q0);
  }
  if (nst > 1) issue_w(1, q1);                                       // stage 1: every wave's pieces
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // ABL & 32 (timing experiments): waves 0 and 4 of workgroup 0 stamp s_memtime at the boundaries of their parts in stages 8 .. 15 into p.ws
  // (s_memtime is a scalar memory read: only at points where no ds_read is pending, or its own lgkmcnt(0) would move them)
  auto stamp = [&](int st, int slot_) {
    if constexpr ((ABL & 32) != 0) {
      if (blockIdx.x == 0 && blockIdx.y == 0 && (wave & 3) == 0 && st >= 8 && st < 16) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) reinterpret_cast<unsigned long long*>(p.ws)[((st - 8) * 2 + grp) * 8 + slot_] = t;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (grp == 0) {
#pragma nounroll
    for (int st = 0; st < nst; ++st) {
      stamp(st, 0);
      read_frags(st, q0);
      if (st >= 1 && st + 1 < nst) issue_w(st + 1, q1);
      issue_halo(q0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(st, 3);
      __builtin_amdgcn_sched_barrier(0);
      barrier();                                                     // ---- end of interval 2 st
      __builtin_amdgcn_sched_barrier(0);
      stamp(st, 4);
      __builtin_amdgcn_s_setprio(0);
      mfmas();
      __builtin_amdgcn_s_setprio(1);
      stamp(st, 5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(st, 6);
      __builtin_amdgcn_sched_barrier(0);
      barrier();                                                     // ---- end of interval 2 st + 1
      __builtin_amdgcn_sched_barrier(0);
      stamp(st, 7);
      q0 = q1, q1 = next_pos(q1);
    }
    barrier();                                                       // group 1's last interval
  } else {
#pragma nounroll
    for (int st = 0; st < nst; ++st) {
      stamp(st, 0);
      if (st > 0) {                                                  // M(st - 1)
        __builtin_amdgcn_s_setprio(0);
        mfmas();
        __builtin_amdgcn_s_setprio(1);
      }
      stamp(st, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(st, 2);
      __builtin_amdgcn_sched_barrier(0);
      barrier();                                                     // ---- end of interval 2 st
      __builtin_amdgcn_sched_barrier(0);
      stamp(st, 3);
      read_frags(st, q0);
      if (st + 2 < nst) issue_w(st + 2, q2);
      issue_halo(q0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(st, 6);
      __builtin_amdgcn_sched_barrier(0);
      barrier();                                                     // ---- end of interval 2 st + 1
      __builtin_amdgcn_sched_barrier(0);
      stamp(st, 7);
      q0 = q1, q1 = q2, q2 = next_pos(q2);
    }
    if (nst > 0) {                                                   // M(nst - 1)
      __builtin_amdgcn_s_setprio(0);
      mfmas();
      __builtin_amdgcn_s_setprio(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
  }
  __builtin_amdgcn_s_setprio(0);
  gemm3_epilogue<E3_STD, NWM, NWN, TN, CH_LDS>(p, acc, af_smem, tile_m, tile_n, wm, wn, fr, fq, tid);
}

// ---- Round 6: the same kernel on a vector-instruction diet ("d").  The counters of the kernel above (profiles/r05b_gemm_pmc.txt) read 2.5 vector
// instructions per MFMA in the main loop: per stage and wave ~50 for the 18 fragment-read addresses (tap offset + XOR swizzle recomputed from the
// pixel index every stage), ~8 per LDS-DMA piece (a 64-bit per-lane address, a zero-page select) -- ~80 against 40 MFMAs, and while the SIMD partner
// owns the matrix pipe an MFMA leaves 8 of every 16 cycles to vector issue: the L part (reads + address arithmetic + DMA issue) was pinned at about
// the length of the partner's M part by instruction COUNT alone.  Here nothing of that is left in the loop:
//   * the nine taps are unrolled and every (tap, 16-row group) fragment address is a register computed once (36 VGPRs); the second K half is that
//     address XOR 64; a chunk change adds +-(one halo buffer) to them (36 adds per nine stages);
//   * weight fragments: nine taps x three ring slots = the slot of a tap is tap % 3 at compile time, so the reads are two base registers + immediates;
//   * LDS-DMA pieces use the scalar-base form (global_load_lds v_offset, s[base]): a weight piece is a per-lane 32-bit offset computed once + a
//     scalar pointer per stage; a halo piece is one multiply-add (pixel x row pitch of the chunk's source) and lanes outside the image are switched
//     off in EXEC instead of being pointed at the zero page -- their 16 bytes of both halo buffers are zeroed once in the prologue and never written.
// Same stages, same hazards, same MFMA order as the kernel above: bit-identical results (tests compare the two).
// TN_ = 4 (round 6, the VAE's channel counts: 128-multiples that are no 160-multiples): a 256 x 128 tile, 16 KB weight stages.  PATCH (images wider than 64
// pixels: the VAE decoder's 128 / 256 / 512 levels): the tile is a 16 x 16-pixel patch of one image (halo 18 x 18 = 324 pixels, 41 pieces) instead of whole
// image rows; everything else -- stages, hazards, MFMA order per output element -- as before.
template <bool TAIL, int TN_ = 5, bool PATCH = false>             // TAIL: the K-concatenated 1x1 shortcut behind the chunks (its own instantiation: the plain kernel keeps its registers)
__global__ __launch_bounds__(512) void af_conv3hd_kernel(const Gemm3Dev p) {
  constexpr int NWM = 4, NWN = 2, TN = TN_, TM = 4, NW = 8, BM = CH_BM, BN = NWN * TN * 16;
  constexpr int WSZ = BN * 128;                                  // one weight stage (CH_WSZ at TN = 5)
  static_assert(!(TAIL && (PATCH || TN_ != 5)), "the K tail exists in the 256 x 160 whole-rows form only");
  constexpr int APW = (CH_NP_MAX + NW - 1) / NW;                 // 7 halo pieces per wave at most
  constexpr int WPW = (BN / 8 + NW - 1) / NW;                    // 3 weight pieces per wave at most (20 pieces; 2 of 16 at TN = 4)
  constexpr int WBASE = 2 * CH_ASZ;                              // the weight ring behind the two halo buffers
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % NWM, wn = wave / NWM;
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.n_major) {
      tile_n = lid / p.tiles_m;
      tile_m = lid - tile_n * p.tiles_m;
    } else {
      tile_m = lid / p.tiles_n;
      tile_n = lid - tile_m * p.tiles_n;
    }
  }
  // whole-rows form: the tile = R whole rows of one image (or R / Ho whole images); Wd = the tile's width in pixels.  PATCH: a 16 x 16 patch at (y0, x0)
  const int Wd = PATCH ? 16 : p.Wo, Hd = p.Ho, Wh = Wd + 2, R = BM / Wd;
  const int Hi = PATCH ? 16 : min(R, Hd), nimg = R / Hi, blk_px = (Hi + 2) * Wh;
  const int halo_px = nimg * blk_px, NP = (halo_px + 7) >> 3;
  const int m0 = tile_m * BM;
  int bimg, y0, x0 = 0;
  if constexpr (PATCH) {
    bimg = tile_m / p.patch_tpi;
    const int t = tile_m - bimg * p.patch_tpi, py = t / p.patch_tx;
    y0 = py * 16;
    x0 = (t - py * p.patch_tx) * 16;
  } else {
    bimg = m0 / p.HoWo;
    y0 = (m0 - bimg * p.HoWo) / Wd;
  }
  const int prow = lane >> 3, slot = lane & 7;
  const int nc1 = p.c1 >> 6;
  // K tail (round 6: the ResBlock's 1x1 shortcut K-concatenated behind the nine tap blocks, af_gemm_desc.a3 / a4): ntail 64-column stages of plain rows
  // (the tile's own 256 output pixels) behind the chunks.  They run in the LAST K split, and the chunk boundaries of the splits are placed so that every
  // split has about the same number of stages (9 per chunk + the tail's).
  const int nmain = (p.c1 + p.c2) >> 6, ntail = TAIL ? (p.c3 + p.c4) >> 6 : 0;
  int cb, ce;
  if (ntail == 0) {
    cb = blockIdx.y * p.kt_per_split;
    ce = min(nmain, cb + p.kt_per_split);
  } else {
    const int S = 9 * nmain + ntail, den = 9 * p.splits;
    cb = blockIdx.y == 0 ? 0 : min(nmain, ((int)blockIdx.y * S + den / 2) / den);
    ce = (int)blockIdx.y + 1 == p.splits ? nmain : min(nmain, (((int)blockIdx.y + 1) * S + den / 2) / den);
  }
  const bool tail_here = ntail > 0 && (int)blockIdx.y + 1 == p.splits;

  // ---- loaders.  Halo piece j of this wave = piece wave + 8 j: lane (prow, slot) fills physical chunk `slot` of halo pixel hp = piece * 8 + prow with the
  // pixel's logical chunk slot ^ (hp & 7) = slot ^ prow -- the same for every piece.  a_pix: the source pixel, -1 outside the image / the halo.
  // (index divisions by the float reciprocal: floor((x + 0.5) / d) is exact for these ranges -- x < 4096, and (2 x + 1) / (2 d) is at least 1 / (2 d) away
  // from an integer while the product's error is below (x + 0.5) * 2e-7 / d -- at 4 vector instructions instead of the ~25 of an integer division; the
  // kernel's prologue was ~900 of them, a fifth of a short launch: profiles/r06j)
  const float inv_blk = __builtin_amdgcn_rcpf((float)blk_px), inv_wh = __builtin_amdgcn_rcpf((float)Wh), inv_wd = __builtin_amdgcn_rcpf((float)Wd),
              inv_hi = __builtin_amdgcn_rcpf((float)Hi);
  auto fdiv = [](int x, float inv) { return (int)(((float)x + 0.5f) * inv); };
  int a_pix[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int hp = (wave + NW * j) * 8 + prow;
    const int im = fdiv(hp, inv_blk), hq = hp - im * blk_px;
    const int hy = fdiv(hq, inv_wh), hx = hq - hy * Wh;
    const int iy = y0 - 1 + hy, ix = x0 + hx - 1;
    const bool ok = hp < halo_px && (unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)p.Wo;
    const int sy = p.upsample ? iy >> 1 : iy, sx = p.upsample ? ix >> 1 : ix;
    a_pix[j] = ok ? ((bimg + im) * p.H + sy) * p.W + sx : -1;
  }
  const unsigned chunk16 = (unsigned)((slot ^ prow) * 16);
  // weight piece j = rows (wave + 8 j) * 8 + prow of the tile; the row's physical chunk `slot` takes logical chunk slot ^ ((row >> 1) & 7), and
  // (row >> 1) & 7 does not depend on j: a per-lane byte offset from the stage's scalar pointer wt + k0
  // (piece j + 1 lies 64 rows behind piece j: the same per-lane offset from a scalar pointer 64 rows further on -- one register instead of three; the
  // tile's rows exist in the packed weight: N % 160 == 0)
  // TAIL instantiation only (it needs the two registers): the plain kernel keeps one offset per piece
  unsigned w_off[TAIL ? 1 : WPW];
#pragma unroll
  for (int j = 0; j < (TAIL ? 1 : WPW); ++j) {
    const int row = (wave + NW * j) * 8 + prow;
    w_off[j] = ((unsigned)min(tile_n * BN + row, p.npad - 1) * (unsigned)p.kpad + (unsigned)((slot ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
  int kpad64 = p.kpad * 64;
  if constexpr (TAIL) asm volatile("" : "+s"(kpad64));
  // kernel arguments the loop uses, as opaque scalar registers (left as kernarg loads they are re-loaded inside the loop, and every scalar load is
  // waited for with lgkmcnt(0), behind the fragment reads)
  const half_t *a1r = p.a1, *a2r = p.a2, *wtr = p.wt;
  int c1r = p.c1, c2r = p.c2;
  asm volatile("" : "+s"(a1r), "+s"(a2r), "+s"(wtr), "+s"(c1r), "+s"(c2r));
  const int Cin = c1r + c2r;
  auto chunk_src = [&](int u, const half_t*& src, int& ld2) {        // workgroup-uniform: a chunk lies in one source; ld2 = row pitch in bytes
    const bool first = u < nc1;
    src = first ? a1r + u * 64 : a2r + (u - nc1) * 64;
    ld2 = (first ? c1r : c2r) * 2;
  };
  auto issue_halo_piece = [&](int j, const half_t* src, int ld2, int buf_off) {
    if (wave + NW * j < NP)
      glds16_sbase_masked(src, (TAIL ? __umul24((unsigned)a_pix[j], (unsigned)ld2) : (unsigned)a_pix[j] * (unsigned)ld2) + chunk16, a_pix[j],
                          af_smem + buf_off + (wave + NW * j) * 1024);   // (TAIL: pixel index, row pitch < 2^24)
  };
  auto issue_wk = [&](int k0, int sl) {                              // this wave's weight pieces of the stage whose first weight column is k0, ring slot sl
    const half_t* wb = wtr + (size_t)k0;
    char* Ws = af_smem + WBASE + sl * WSZ;
#pragma unroll
    for (int j = 0; j < WPW; ++j)
      if (wave + NW * j < BN / 8) {
        if constexpr (TAIL) glds16_sbase(wb + (size_t)j * kpad64, w_off[0], Ws + (wave + NW * j) * 1024);
        else glds16_sbase(wb, w_off[j], Ws + (wave + NW * j) * 1024);
      }
  };
  auto issue_w = [&](int u, int tap, int sl) { issue_wk(tap * Cin + u * 64, sl); };     // ... of the stage at (chunk u, tap)
  // K-tail stage t: the tile's 256 rows x 64 columns of a3 | a4 as 32 pieces of 8 rows (wave w: pieces w + 8 j, j < 4), row q at q * 128 of the buffer,
  // chunk index XOR (q & 7) = XOR prow: the same per-lane chunk as the halo pieces.  Every row is a valid output pixel (M % 256 == 0): no masking.
  // (kernel arguments read where they are used, behind the main loop: nothing of the tail is live inside it -- the loop has no register to spare)
  auto issue_tail_piece = [&](int j, int t, int buf_off) {
    const bool first = t * 64 < p.c3;
    const half_t* src = first ? p.a3 + t * 64 : p.a4 + (t * 64 - p.c3);
    const unsigned ld2 = (unsigned)(first ? p.lda3 : p.lda4) * 2u;
    const int pc = wave + NW * j;
    glds16_sbase(src, (unsigned)(m0 + pc * 8 + prow) * ld2 + chunk16, af_smem + buf_off + pc * 1024);
  };

#ifdef AF_CONV3H_GN_PROBE
  // TIMING PROBE ONLY (-DAF_CONV3H_GN_PROBE, tools/probes/r06ab_gn_in_conv_probe.sh; results are NOT a convolution of the input): what the round-5 review's
  // item 1.ii -- GroupNorm-apply + SiLU inside this kernel -- would add to the loop.  Halo piece j of chunk u, 16 bytes per lane, is read back from LDS once
  // it has landed, turned into silu(x * g[c] * r + s) with per-channel factors from memory and written back (pad positions stay zero), one piece per
  // stage in the L part of the stage after the one that requested it: the best schedule the loop offers.
  const float probe_r = 1.0f + p.ln_eps, probe_s = p.ln_eps;
  auto gn_probe_piece = [&](int j, int buf_off, int u) {
    if (wave + NW * j < NP) {
      char* q = af_smem + buf_off + (wave + NW * j) * 1024 + lane * 16;
      const half8_t v = *reinterpret_cast<const half8_t*>(q);
      const int ch = ((u * 64) & 255) + (int)(chunk16 >> 1);
      const floatx4 g0 = *reinterpret_cast<const floatx4*>(p.bias + ch), g1 = *reinterpret_cast<const floatx4*>(p.bias + ch + 4);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = af_silu(fmaf((float)v[e] * (e < 4 ? g0[e] : g1[e - 4]), probe_r, probe_s));
        o[e] = a_pix[j] >= 0 ? (half_t)y : (half_t)0.f;
      }
      *reinterpret_cast<half8_t*>(q) = o;
    }
  };
#define CHD_GN_PROBE(TAP) if (!last && (TAP) >= 1 && (TAP) - 1 < APW) gn_probe_piece((TAP) >= 1 ? (TAP) - 1 : 0, nbuf, u + 1);
#else
#define CHD_GN_PROBE(TAP)
#endif
  floatx4 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  // fragment addresses (bytes from af_smem), first K half; second half = address ^ 64 (chunk 4 + fq instead of fq under the XOR swizzle)
  const int wr0 = WBASE + (wn * TN * 16) * 128 + fr * 128 + ((fq ^ (fr >> 1)) * 16);
  int fa[9][TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int q = wm * 64 + tm * 16 + fr;
    const int r = fdiv(q, inv_wd), im = fdiv(r, inv_hi);
    const int hpb = im * blk_px + (r - im * Hi) * Wh + (q - r * Wd);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hp = hpb + (tap / 3) * Wh + (tap % 3);
      fa[tap][tm] = hp * 128 + ((fq ^ (hp & 7)) * 16);
    }
  }

  if (p.wpf > 0 && p.splits == 1)
    // (the dump goes to weight slot 2, which nothing fills before the first barrier: every wave's dump loads have landed by then -- its own vmcnt(0) --
    // while a dump in halo buffer 1, as in the kernel above, could land on a pad position AFTER another wave has zeroed it)
    af_prefetch_weight_tile(p.wt, p.kpad, p.npad, tile_n * BN, BN, 0, p.kpad >> 6, p.wpf_coop, tile_m % p.wpf_coop, p.wpf, NW, wave, lane, af_smem + WBASE + 2 * WSZ);

  const int grp = wave >> 2;
  half8_t wf[TN], xf[TM], wf1[TN], xf1[TM];
  auto mfmas = [&]() {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[tn], xf1[tm], acc[tn][tm], 0, 0, 0);
  };
#define CHD_READ_FRAGS(TAP)                                                                                                      \
  {                                                                                                                              \
    constexpr int wsl_ = ((TAP) % 3) * WSZ;                                                                                   \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(af_smem + wr0 + wsl_ + tn * 2048);               \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(af_smem + fa[TAP][tm]);                            \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) wf1[tn] = *reinterpret_cast<const half8_t*>(af_smem + (wr0 ^ 64) + wsl_ + tn * 2048);         \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) xf1[tm] = *reinterpret_cast<const half8_t*>(af_smem + (fa[TAP][tm] ^ 64));                    \
  }

  __builtin_amdgcn_s_setprio(1);
  const int nchunks = ce - cb;
  if (nchunks > 0) {
    const half_t* src;
    int ld2;
    chunk_src(cb, src, ld2);
#pragma unroll
    for (int j = 0; j < APW; ++j) issue_halo_piece(j, src, ld2, 0);
    issue_w(cb, 0, 0);
    issue_w(cb, 1, 1);                                               // stage 1: every wave's pieces
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AF_CONV3H_GN_PROBE
  if (nchunks > 0) {
#pragma unroll
    for (int j = 0; j < APW; ++j) gn_probe_piece(j, 0, cb);
  }
#endif
  // the halo's out-of-image positions: zero in BOTH buffers, once (no DMA ever writes them)
#pragma unroll
  for (int j = 0; j < APW; ++j)
    if (wave + NW * j < NP && a_pix[j] < 0) {
      char* z = af_smem + (wave + NW * j) * 1024 + lane * 16;
      *reinterpret_cast<uintx4_t*>(z) = uintx4_t{0u, 0u, 0u, 0u};
      *reinterpret_cast<uintx4_t*>(z + CH_ASZ) = uintx4_t{0u, 0u, 0u, 0u};
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  if (grp == 0) {
#pragma nounroll
    for (int u = cb; u < ce; ++u) {
      const bool first = u == cb, last = u + 1 == ce;
      const half_t* hsrc = a1r;
      int hld2 = 0;
      if (!last) chunk_src(u + 1, hsrc, hld2);
      const int nbuf = ((u + 1 - cb) & 1) * CH_ASZ;
#define CHD_STAGE0(TAP)                                                                                                          \
      {                                                                                                                          \
        CHD_READ_FRAGS(TAP)                                                                                                      \
        if (!(first && (TAP) == 0) && !(last && (TAP) == 8)) issue_w((TAP) < 8 ? u : u + 1, ((TAP) + 1) % 9, ((TAP) + 1) % 3);   \
        if (!last && (TAP) < APW) issue_halo_piece((TAP) < APW ? (TAP) : 0, hsrc, hld2, nbuf);                                   \
        CHD_GN_PROBE(TAP)                                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                                           \
        mfmas();                                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      }
      CHD_STAGE0(0) CHD_STAGE0(1) CHD_STAGE0(2) CHD_STAGE0(3) CHD_STAGE0(4) CHD_STAGE0(5) CHD_STAGE0(6) CHD_STAGE0(7) CHD_STAGE0(8)
#undef CHD_STAGE0
      const int d = ((u - cb) & 1) ? -CH_ASZ : CH_ASZ;               // the next chunk's halo lies in the other buffer
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) fa[tap][tm] += d;
    }
    __builtin_amdgcn_s_barrier();                                    // group 1's last interval
  } else {
#pragma nounroll
    for (int u = cb; u < ce; ++u) {
      const bool first = u == cb, last = u + 1 == ce;
      const half_t* hsrc = a1r;
      int hld2 = 0;
      if (!last) chunk_src(u + 1, hsrc, hld2);
      const int nbuf = ((u + 1 - cb) & 1) * CH_ASZ;
#define CHD_STAGE1(TAP)                                                                                                          \
      {                                                                                                                          \
        if (!(first && (TAP) == 0)) {                                                                                            \
          __builtin_amdgcn_s_setprio(0);                                                                                         \
          mfmas();                                                                                                               \
          __builtin_amdgcn_s_setprio(1);                                                                                         \
        }                                                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        CHD_READ_FRAGS(TAP)                                                                                                      \
        if (!(last && (TAP) >= 7)) issue_w((TAP) < 7 ? u : u + 1, ((TAP) + 2) % 9, ((TAP) + 2) % 3);                             \
        if (!last && (TAP) < APW) issue_halo_piece((TAP) < APW ? (TAP) : 0, hsrc, hld2, nbuf);                                   \
        CHD_GN_PROBE(TAP)                                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      }
      CHD_STAGE1(0) CHD_STAGE1(1) CHD_STAGE1(2) CHD_STAGE1(3) CHD_STAGE1(4) CHD_STAGE1(5) CHD_STAGE1(6) CHD_STAGE1(7) CHD_STAGE1(8)
#undef CHD_STAGE1
      const int d = ((u - cb) & 1) ? -CH_ASZ : CH_ASZ;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) fa[tap][tm] += d;
    }
    if (nchunks > 0) {                                               // M(last stage)
      __builtin_amdgcn_s_setprio(0);
      mfmas();
      __builtin_amdgcn_s_setprio(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef CHD_READ_FRAGS
#undef CHD_GN_PROBE
  if (TAIL && tail_here) {
    // ---- the K tail, lock step (every wave: wait, barrier, issue the next stage, 18 fragment reads, 40 MFMAs): a tail stage moves a whole 32 KB A tile
    // + 20 KB of weights, the tap-by-tap tile's traffic, and one stage of look-ahead is all the three-slot ring and the two halo buffers allow.
    // Stage 0's operands are requested here (every read and LDS-DMA of the main loop has retired behind its last barrier): one exposed round trip
    // per launch; prefetching them under the last chunk would cost the main loop registers it does not have (it sits at 256).
    int tb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_tail_piece(j, 0, tb);
    issue_wk(9 * Cin, 0);
    int ft[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) ft[tm] = (wm * 64 + tm * 16 + fr) * 128 + ((fq ^ (fr & 7)) * 16);
#pragma nounroll
    for (int t = 0; t < ntail; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int tn_ = tb == 0 ? CH_ASZ : 0;
      if (t + 1 < ntail) {
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_tail_piece(j, t + 1, tn_);
        issue_wk(9 * Cin + (t + 1) * 64, (t + 1) % 3);
      }
      const int ws = (t % 3) * WSZ;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(af_smem + wr0 + ws + tn * 2048);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(af_smem + tb + ft[tm]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) wf1[tn] = *reinterpret_cast<const half8_t*>(af_smem + (wr0 ^ 64) + ws + tn * 2048);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) xf1[tm] = *reinterpret_cast<const half8_t*>(af_smem + tb + (ft[tm] ^ 64));
      __builtin_amdgcn_s_setprio(0);
      mfmas();
      __builtin_amdgcn_s_setprio(1);
      tb = tn_;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // the staged epilogue reuses the LDS
  }
  __builtin_amdgcn_s_setprio(0);
  gemm3_epilogue<E3_STD, NWM, NWN, TN, CH_LDS, PATCH>(p, acc, af_smem, tile_m, tile_n, wm, wn, fr, fq, tid);
}

// scope of the halo-resident kernel
// 0 = outside; 1 = the 256 x 160 tile on whole image rows (the U-Net's levels); 2 = the 256 x 128 tile on whole image rows; 3 = the 256 x 128 tile on
// 16 x 16-pixel patches (images wider than 64 pixels: the VAE decoder's 128 / 256 / 512 levels)
static int conv3h_variant(const af_gemm_desc* d) {
  if (d->taps != 9 || (d->upsample != 0 && d->upsample != 1) || d->tap_shift || d->c1 % 64 != 0 || d->c2 % 64 != 0) return 0;
  const bool n160 = d->N % CH_BN == 0;
  if (!n160 && d->N % 128 != 0) return 0;
  if (d->c3 || d->c4) {                                            // K tail (round 6): plain rows of a3 | a4 on the output grid, whole 64-column stages
    if (!n160 || d->c3 <= 0 || d->c3 % 64 != 0 || d->c4 < 0 || d->c4 % 64 != 0 || d->a3 == nullptr || (d->c4 > 0 && d->a4 == nullptr) || d->upsample) return 0;
    if ((long)d->M * std::max(d->lda3 ? d->lda3 : d->c3, d->lda4 ? d->lda4 : d->c4) * 2 >= (1L << 32)) return 0;
  }
  const int up = d->upsample ? 2 : 1;                              // nearest x2 folded into the halo gather
  if ((d->stride ? d->stride : 1) != 1 || d->Ho != up * d->H || d->Wo != up * d->W) return 0;
  if (d->M % CH_BM != 0 || d->M != d->B * d->Ho * d->Wo) return 0;
  if (d->act == AF_ACT_GEGLU || d->out_mode == AF_OUT_SPLIT_T || d->ln_colsum != nullptr || d->kpad % 64 != 0) return 0;
  if ((long)d->B * d->H * d->W * std::max(d->c1, d->c2) * 2 >= (1L << 32)) return 0;      // the halo gather's 32-bit byte offsets
  if (d->Wo > 64) {                                                // patches: no K tail, no split-K (the caller checks), no 160-wide form
    if (n160 || d->Wo % 16 != 0 || d->Ho % 16 != 0 || d->c3 || d->c4) return 0;
    return 3;
  }
  if (d->Wo != 8 && d->Wo != 16 && d->Wo != 32 && d->Wo != 64) return 0;
  const int rows = CH_BM / d->Wo;                                   // a tile: `rows` whole rows of one image, or rows / Ho whole images
  if (d->Ho % rows != 0 && rows % d->Ho != 0) return 0;
  if (d->Ho < rows && (rows / d->Ho) * (d->Ho + 2) * (d->Wo + 2) > CH_NP_MAX * 8) return 0;     // the images' halo blocks must fit one halo buffer
  return n160 ? 1 : 2;
}
static bool conv3h_eligible(const af_gemm_desc* d) { return conv3h_variant(d) != 0; }

static int conv3h_chunks(const af_gemm_desc* d) { return (d->c1 + d->c2) / 64; }

static bool launch_conv3h(const Gemm3Dev& p0, hipStream_t stream, bool r5_loop, int variant) {
  Gemm3Dev p = p0;
  p.tiles_n = p.N / (variant == 1 ? CH_BN : 128);
  p.tiles_m = p.M / CH_BM;
  p.patch_tx = p.Wo / 16;
  p.patch_tpi = p.patch_tx * (p.Ho / 16);
  {
    static const int coop_env = getenv("AF_GEMM3_WPF_COOP") ? atoi(getenv("AF_GEMM3_WPF_COOP")) : 32;
    p.wpf_coop = p.tiles_m < coop_env ? p.tiles_m : coop_env;
    if (p.wpf > AF_WPF_MAX) p.wpf = AF_WPF_MAX;
  }
  if (p.counters && (p.splits <= 1 || p.splits > 4 || p.tiles_m * p.tiles_n > AF_SPLITK_MAX_TILES)) p.counters = nullptr;
  p.n_major = af_gemm_n_major(p.M, p.N, p.K, p.c1 + p.c2);
  dim3 grid(p.tiles_m * p.tiles_n, p.splits), block(512);
  if (variant >= 2) {                                              // 256 x 128 tile (round 6): the diet loop only
    static bool set_2 = false, set_3 = false;
    if (variant == 2) {
      if (af_allow_dyn_lds(reinterpret_cast<const void*>(&af_conv3hd_kernel<false, 4, false>), CH_LDS, set_2, "af_gemm"))
        hipLaunchKernelGGL((af_conv3hd_kernel<false, 4, false>), grid, block, CH_LDS, stream, p);
    } else if (af_allow_dyn_lds(reinterpret_cast<const void*>(&af_conv3hd_kernel<false, 4, true>), CH_LDS, set_3, "af_gemm")) {
      hipLaunchKernelGGL((af_conv3hd_kernel<false, 4, true>), grid, block, CH_LDS, stream, p);
    }
    return p.counters != nullptr;
  }
#define AF_CONV3H_CASE(A)                                                                                                                     \
  case A: {                                                                                                                                   \
    static bool set_ = false;                                                                                                                 \
    if (af_allow_dyn_lds(reinterpret_cast<const void*>(&af_conv3h_kernel<A>), CH_LDS, set_, "af_gemm"))                                       \
      hipLaunchKernelGGL(af_conv3h_kernel<A>, grid, block, CH_LDS, stream, p);                                                                \
    break;                                                                                                                                    \
  }
  static const bool diet_env = !(getenv("AF_CONV3H_DIET") && atoi(getenv("AF_CONV3H_DIET")) == 0);    // A/B switch: 0 = the round-5 loop
  static const bool diet_dynamic = getenv("AF_GEMM3_ABLATE_DYNAMIC") != nullptr;
  const bool diet = diet_dynamic ? !(getenv("AF_CONV3H_DIET") && atoi(getenv("AF_CONV3H_DIET")) == 0) : diet_env;
  if ((diet || p.c3 > 0) && !r5_loop && (p.ablate & 63) == 0) {
    static bool set_d = false, set_t = false;
    if (p.c3 > 0) {
      if (af_allow_dyn_lds(reinterpret_cast<const void*>(&af_conv3hd_kernel<true>), CH_LDS, set_t, "af_gemm"))
        hipLaunchKernelGGL(af_conv3hd_kernel<true>, grid, block, CH_LDS, stream, p);
    } else if (af_allow_dyn_lds(reinterpret_cast<const void*>(&af_conv3hd_kernel<false>), CH_LDS, set_d, "af_gemm")) {
      hipLaunchKernelGGL(af_conv3hd_kernel<false>, grid, block, CH_LDS, stream, p);
    }
    return p.counters != nullptr;
  }
  switch (p.ablate & 63) {
#ifdef AF_CONV3H_ABLATIONS                                         // timing experiments only (tools/probes/r05s_conv3hp_ablate.sh builds with this)
    AF_CONV3H_CASE(1) AF_CONV3H_CASE(2) AF_CONV3H_CASE(3) AF_CONV3H_CASE(4) AF_CONV3H_CASE(7) AF_CONV3H_CASE(8) AF_CONV3H_CASE(16) AF_CONV3H_CASE(17) AF_CONV3H_CASE(18) AF_CONV3H_CASE(19) AF_CONV3H_CASE(32)
#endif
    default:
    AF_CONV3H_CASE(0)
  }
#undef AF_CONV3H_CASE
  return p.counters != nullptr;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused feed-forward of a transformer block at C = 320 (the 64 x 64 level: 4096 tokens per image):
//     out = x + b2 + W2 ( v * gelu(g) ),   [v | g] = W1 LN(x) + b1                 (attention.py:31-58, 242-252)
// in ONE launch per layer.  Unfused, the GEGLU projection writes a [tokens, 1280] intermediate (84 MB at U-Net batch 8) that the
// second GEMM reads straight back, and both GEMMs are short-K (K = 320 / epilogue-heavy): 111 + 35 us per block in the step.
// Here a workgroup owns 128 tokens for the whole feed-forward:
//   * the token tile X [128 x 320] is brought into LDS once (80 KB, the five 64-wide K chunks of the whole-line layout) and its
//     LayerNorm statistics are taken from it once (the norm itself is folded into W1 as in af_gemm_desc.ln_colsum);
//   * the hidden dimension is walked in 20 chunks of 64 GEGLU outputs (128 interleaved W1 rows = per wave one value and one gate
//     MFMA tile): GEMM1 streams the chunk's W1 rows through a two-slot ring (5 stages of 16 KB), its GEGLU epilogue leaves the
//     chunk's activations G [128 x 64] in LDS as the next MFMA operand (in the ring slot the last stage has just left), and GEMM2
//     accumulates OUT [128 x 320] += G W2[:, chunk]^T from a 40 KB tile of W2 that arrived under GEMM1;
//   * OUT lives in registers across the chunks (80 VGPRs per wave) and leaves through the staged epilogue (bias, residual).
// LDS: 80 (X) + 32 (W1 ring / G) + 40 (W2 tile) + 1 (row statistics) = 153 KB; 8 waves as 2 x 4.
constexpr int FF_C = 320, FF_BM = 128, FF_HC = 64;              // channels, tokens per workgroup, GEGLU outputs per hidden chunk
constexpr int FF_XS = 5 * FF_BM * 128, FF_W1S = 2 * FF_HC * 128, FF_W2B = FF_C * 128;
constexpr int FF_LDS = FF_XS + 2 * FF_W1S + FF_W2B + FF_BM * 8;

struct FfDev {
  const half_t* x;
  const half_t* w1;      // packed, GEGLU-interleaved [2 * inner][kpad1]
  const float* b1;       // [2 * inner] (interleaved; the folded LayerNorm's beta term included)
  const float* cs1;      // column sums of the packed W1 rows (folded LayerNorm); any readable [2 * inner] floats when ln_on = 0
  int ln_on;
  const half_t* w2;      // packed [npad >= 320][kpad2]
  const half_t* zeros;
  int M, inner, kpad1, kpad2;
  float ln_eps;
  Gemm3Dev epi;          // what the final (standard, staged) epilogue reads: bias = b2, residual, out, M, N = 320, ld_out
  // TAILP (af_ff_chain, round 6): the SpatialTransformer's proj_out + residual behind the feed-forward (attention.py:287-304): out = x_in + b_p + W_p x3 with
  // x3 = the feed-forward's own result (epi: b2 + residual, never written to memory); nullptr / unused otherwise
  const half_t* wp;      // packed [>= 320][kpad_p]
  int kpad_p;
  Gemm3Dev epi_p;        // the tail's epilogue: bias = b_p, residual = x_in, out, GroupNorm partials of the block's output
};

// TAILP: x3 = OUT + b2 + residual goes into the X tile's LDS (the swizzled fp16 layout GEMM1 read x from) instead of memory, and a third GEMM, 128 x 320 x 320 with
// W_p, runs over it: W_p streams as 2 row halves x 5 K stages of [160 rows x 64 K] (20 KB) through a three-slot ring in the W1 / W2 area (72 KB); the waves keep
// their 64 x 80 output tiles, so a row half is computed by the four waves that own its columns (a 26 MFLOP GEMM: the MFMA time is not what it costs).
template <bool TAILP>
__global__ __launch_bounds__(512, 1) void af_ff320_kernel(FfDev p) {
  constexpr int NW = 8, NWM = 2, TM = 4;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  char* Xs = af_smem;
  char* W1r = af_smem + FF_XS;
  char* W2b = W1r + 2 * FF_W1S;
  float* st = reinterpret_cast<float*>(W2b + FF_W2B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % NWM, wn = wave / NWM;
  const int tile_m = blockIdx.x;
  const int prow = lane >> 3, slot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int rd0 = fr * 128 + (((0 * 4 + fq) ^ (fr >> 1)) * 16);
  const int rd1 = fr * 128 + (((1 * 4 + fq) ^ (fr >> 1)) * 16);
  const int nchunk = p.inner / FF_HC;

  // weights of both GEMMs towards this XCD's L2 (af_common.h): 1/32 of each per workgroup
  {
    const int coop = gridDim.x < 32 ? gridDim.x : 32;
    af_prefetch_weight_tile(p.w1, p.kpad1, 2 * p.inner, 0, 2 * p.inner, 0, FF_C / 64, coop, tile_m % coop, 1, NW, wave, lane, af_smem + FF_LDS);
    af_prefetch_weight_tile(p.w2, p.kpad2, FF_C, 0, FF_C, 0, p.inner / 64, coop, tile_m % coop, 1, NW, wave, lane, af_smem + FF_LDS);
  }

  // ---- X tile: chunk c = rows x 64 halves, 16 pieces of 8 rows; wave w takes pieces {2w, 2w+1} of every chunk
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 8 + prow;
      const int m = tile_m * FF_BM + row;
      const int lc = slot ^ ((row >> 1) & 7);
      const half_t* g = m < p.M ? p.x + (size_t)m * FF_C + c * 64 + lc * 8 : p.zeros;
      glds16(g, Xs + c * (FF_BM * 128) + (wave * 2 + j) * 1024);
    }
  // ---- W1 / W2 loaders: a W1 stage is 128 rows x 64 halves (16 pieces: 2 per wave), the W2 tile 320 rows x 64 halves (40 pieces: 5 per wave)
  auto issue_w1 = [&](int g) {                                   // global stage g = chunk * 5 + K chunk
    const int jc = g / 5, sc = g - jc * 5;
    char* dst = W1r + (g & 1) * FF_W1S;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      glds16(p.w1 + (size_t)(jc * 2 * FF_HC + row) * p.kpad1 + sc * 64 + lc * 8, dst + (wave * 2 + j) * 1024);
    }
  };
  auto issue_w2 = [&](int jc) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int row = (wave * 5 + j) * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      glds16(p.w2 + (size_t)row * p.kpad2 + jc * 64 + lc * 8, W2b + (wave * 5 + j) * 1024);
    }
  };
  issue_w1(0);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");               // the X tile (and the L2 touches) have landed; stage 0 of W1 may still fly
  __builtin_amdgcn_s_barrier();

  // ---- LayerNorm statistics of the 128 rows, once: wave w takes rows 16 w .. 16 w + 15
  {
    const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
    float s = 0.f, q = 0.f;
    const char* row = Xs + (wave * 16) * 128;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const half8_t f0 = *reinterpret_cast<const half8_t*>(row + c * (FF_BM * 128) + rd0), f1 = *reinterpret_cast<const half8_t*>(row + c * (FF_BM * 128) + rd1);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const half2_t a = {f0[e], f0[e + 1]}, b = {f1[e], f1[e + 1]};
        s = __builtin_amdgcn_fdot2(a, one2, s, false);
        q = __builtin_amdgcn_fdot2(a, a, q, false);
        s = __builtin_amdgcn_fdot2(b, one2, s, false);
        q = __builtin_amdgcn_fdot2(b, b, q, false);
      }
    }
    s += __shfl_xor(s, 16, 64);
    q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    if (fq == 0) {
      const float mean = s * (1.0f / FF_C), var = fmaxf(q * (1.0f / FF_C) - mean * mean, 0.f);
      st[2 * (wave * 16 + fr)] = p.ln_on ? mean : 0.f;
      st[2 * (wave * 16 + fr) + 1] = p.ln_on ? rsqrtf(var + p.ln_eps) : 1.f;
    }
  }
  __syncthreads();
  float mean[TM], rstd[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = wm * 64 + tm * 16 + fr;
    mean[tm] = st[2 * row];
    rstd[tm] = st[2 * row + 1];
  }

  floatx4 acc2[5][TM];
#pragma unroll
  for (int tn = 0; tn < 5; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc2[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};

  for (int jc = 0; jc < nchunk; ++jc) {
    floatx4 acc1[2][TM];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc1[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int nv = jc * 2 * FF_HC + wn * 32 + 4 * fq;            // packed W1 row of the lane's value columns; their gates are 16 rows further
    floatx4 bv, bg, cv, cg;
    // ---- GEMM1: H [128 x 128 W1 rows] = X W1_chunk^T over the five K chunks
#pragma unroll
    for (int sc = 0; sc < 5; ++sc) {
      const int g = jc * 5 + sc;
      // stage g was issued one stage ago.  In stage 0 it is followed by the chunk's bias / column-sum loads and the 5 pieces of the W2
      // tile, which may stay in flight through stage 1 (counted wait); from stage 2 on everything older must have landed
      if (sc == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // ... for every wave; everyone is done with the slot refilled next
      if (g + 1 < nchunk * 5) issue_w1(g + 1);
      if (sc == 0) {
        // this chunk's GEGLU bias / column sums for the lane's four value and four gate columns, then the W2 tile: both land under
        // GEMM1 (the previous chunk's GEMM2, which read the W2 buffer, is behind the barrier above)
        bv = *reinterpret_cast<const floatx4*>(p.b1 + nv);
        bg = *reinterpret_cast<const floatx4*>(p.b1 + nv + 16);
        cv = *reinterpret_cast<const floatx4*>(p.cs1 + nv);
        cg = *reinterpret_cast<const floatx4*>(p.cs1 + nv + 16);
        issue_w2(jc);
      }
      const char* As = Xs + sc * (FF_BM * 128);
      const char* Ws = W1r + (g & 1) * FF_W1S;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int rd = kk ? rd1 : rd0;
        half8_t wf[2], xf[TM];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + (wn * 32 + tn * 16) * 128 + rd);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(As + (wm * 64 + tm * 16) * 128 + rd);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) acc1[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc1[tn][tm], 0, 0, 0);
      }
    }
    // ---- GEGLU epilogue of the chunk -> G [128 x 64] in the ring slot the last stage has just left (every wave must be out of it)
    __builtin_amdgcn_s_barrier();
    {
      char* G = W1r + ((jc * 5 + 4) & 1) * FF_W1S;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = wm * 64 + tm * 16 + fr;
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xv = rstd[tm] * (acc1[0][tm][e] - mean[tm] * cv[e]) + bv[e];
          const float gv = rstd[tm] * (acc1[1][tm][e] - mean[tm] * cg[e]) + bg[e];
          h[e] = (half_t)(xv * af_gelu_erf(gv));
        }
        // column wn * 16 + 4 fq + e of the chunk: 16-byte slot (wn * 2 + fq / 2), swizzled as the readers expect, low / high half by fq & 1
        const int sl = (wn * 2 + (fq >> 1)) ^ ((row >> 1) & 7);
        *reinterpret_cast<half4_t*>(G + row * 128 + sl * 16 + (fq & 1) * 8) = h;
      }
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // the W2 tile is older than the stage issued last (2 pieces): landed
    __syncthreads();                                             // G complete, W2 tile complete for every wave
    // ---- GEMM2: OUT [128 x 320] += G W2[:, chunk]^T
    {
      const char* G = W1r + ((jc * 5 + 4) & 1) * FF_W1S;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int rd = kk ? rd1 : rd0;
        half8_t wf[5], xf[TM];
#pragma unroll
        for (int tn = 0; tn < 5; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(W2b + (wn * 80 + tn * 16) * 128 + rd);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(G + (wm * 64 + tm * 16) * 128 + rd);
#pragma unroll
        for (int tn = 0; tn < 5; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) acc2[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc2[tn][tm], 0, 0, 0);
      }
    }
  }
  if constexpr (!TAILP) {
    gemm3_epilogue<E3_STD, 2, 4, 5, FF_XS + 2 * FF_W1S + FF_W2B>(p.epi, acc2, af_smem, tile_m, 0, wm, wn, fr, fq, tid, nullptr);
  } else {
    constexpr int PST = 160 * 128;                                 // one W_p stage: 160 rows x 64 K
    static_assert(3 * PST <= 2 * FF_W1S + FF_W2B, "the three-slot W_p ring lives in the W1 ring + W2 tile area");
    auto issue_wp = [&](int sidx) {                                // stage sidx = (row half sidx / 5, K chunk sidx % 5): 20 pieces of 8 rows, wave w takes w, w + 8, (w + 16)
      const int nh = sidx >= 5 ? 1 : 0, kc = sidx - 5 * nh;
      char* dst = W1r + (sidx % 3) * PST;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pc = wave + NW * j;
        if (pc < 20) {
          const int row = pc * 8 + prow;
          const int lc = slot ^ ((row >> 1) & 7);
          glds16(p.wp + (size_t)(nh * 160 + row) * p.kpad_p + kc * 64 + lc * 8, dst + pc * 1024);
        }
      }
    };
    __syncthreads();                                               // every wave is done with x, G and the W2 tile
    issue_wp(0);
    issue_wp(1);
    // x3 = OUT + b2 + residual (the arithmetic and order of the standard epilogue) -> fp16 -> the X tile
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = wm * 64 + tm * 16 + fr, m = tile_m * FF_BM + row;
#pragma unroll
      for (int tn = 0; tn < 5; ++tn) {
        const int c = wn * 80 + tn * 16 + 4 * fq;
        floatx4 v = acc2[tn][tm];
        half4_t o = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if (m < p.M) {
          if (p.epi.bias) {
            const floatx4 bv = *reinterpret_cast<const floatx4*>(p.epi.bias + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
          }
          if (p.epi.residual) {
            const half4_t rv = *reinterpret_cast<const half4_t*>(p.epi.residual + (size_t)m * FF_C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
          }
          o = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
        *reinterpret_cast<half4_t*>(Xs + (c >> 6) * (FF_BM * 128) + row * 128 + ((((c & 63) >> 3) ^ ((row >> 1) & 7)) * 16) + ((c >> 2) & 1) * 8) = o;
      }
    }
    floatx4 acc3[5][TM];
#pragma unroll
    for (int tn = 0; tn < 5; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc3[tn][tm] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int sidx = 0; sidx < 10; ++sidx) {
      // stage sidx must have landed; the stage behind it (this wave's 3 / 2 pieces, issued one iteration ago) may still fly
      if (sidx == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (wave < 4) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                // ... for every wave (first pass: the x3 tile is complete); the slot refilled next is free
      if (sidx + 2 < 10) issue_wp(sidx + 2);
      const int nh = sidx >= 5 ? 1 : 0, kc = sidx - 5 * nh;
      if ((wn >> 1) == nh) {
        const char* Ws = W1r + (sidx % 3) * PST;
        const char* As = Xs + kc * (FF_BM * 128);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int rd = kk ? rd1 : rd0;
          half8_t wf[5], xf[TM];
#pragma unroll
          for (int tn = 0; tn < 5; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + ((wn & 1) * 80 + tn * 16) * 128 + rd);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(As + (wm * 64 + tm * 16) * 128 + rd);
#pragma unroll
          for (int tn = 0; tn < 5; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc3[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc3[tn][tm], 0, 0, 0);
        }
      }
    }
    gemm3_epilogue<E3_STD, 2, 4, 5, FF_XS + 2 * FF_W1S + FF_W2B>(p.epi_p, acc3, af_smem, tile_m, 0, wm, wn, fr, fq, tid, nullptr);
  }
}

}  // namespace

// Which form of the halo-resident kernel (tile 14) a descriptor would run (include/adaface_hip.h)
extern "C" int af_gemm_halo_variant(const af_gemm_desc* d) { return d != nullptr ? conv3h_variant(d) : 0; }

// Called by af_gemm (af_gemm.hip) for tile == 3 after the common argument validation.  Returns 1 if the shape is
// outside this kernel's scope (caller falls back), 0 after a launch.
int af_gemm3_try_launch(const af_gemm_desc* d, int splits, int wide, hipStream_t stream) {
  // wide: 0 = 128 x 128, 1 = 128 x 320 (GEGLU 128 x 256), 2 = 256 x 256, 3 = 256 x 320 (8 waves as 4 x 2); whole-line kernel:
  // 4 = 128 x 320 (GEGLU 128 x 256), 5 = 128 x 128, 6 = GEGLU 256 x 320, 7 = GEGLU 256 x 256, 8 = 128 x 160 (4 waves, 2 workgroups per CU),
  // 9 / 10 = 128 x 128 / 128 x 160 with a four-slot ring (three stages in flight, one workgroup per CU)
  const bool geglu = d->act == AF_ACT_GEGLU, split_t = d->out_mode == AF_OUT_SPLIT_T;
  const bool halo_r5 = wide == 16;                              // tile 19: tile 14 with the round-5 main loop (A/B arm and bit-identity reference of tile 14)
  if (halo_r5) wide = 11;
  const int halo_variant = wide == 11 ? conv3h_variant(d) : 0;
  if (wide == 11 && halo_variant == 0) return 1;                // halo-resident 3x3 kernel (tile 14)
  if (halo_r5 && (d->c3 > 0 || halo_variant != 1)) return 1;    // (the K tail and the 256 x 128 forms exist in the round-6 loop only)
  if (halo_variant == 3 && splits > 1) return 1;                // (patches: never split)
  if (d->upsample && !((wide == 4 || wide == 5 || (wide >= 8 && wide <= 12)) && d->upsample == 1 && d->taps == 9)) return 1;   // nearest x2: whole-line kernel only
  if (d->c1 % BK3 != 0 || d->c2 % BK3 != 0 || d->zeros == nullptr) return 1;
  if (d->c3 > 0 && (wide < 4 || (wide > 10 && wide != 12 && wide != 11) || d->taps != 9 || d->upsample || (d->stride != 0 && d->stride != 1))) return 1;   // K tail: whole-line tap-by-tap tiles + the halo-resident kernel
  if ((geglu || split_t) && (d->taps != 1 || splits > 1)) return 1;
  if (geglu && (!wide || d->N % (wide == 5 ? 128 : 256) != 0)) return 1;   // GEGLU: 128 x 256 tile, the 256-row tiles, or 128 x 128 whole-line
  if (!geglu && wide == 1 && d->N % 320 != 0) return 1;
  if (wide >= 4) {                                            // whole-line variants: 64-multiples of channels and K padding
    if (d->c1 % 64 != 0 || d->c2 % 64 != 0 || d->kpad % 64 != 0) return 1;
    if (wide == 4 && (geglu ? d->N % 256 != 0 : d->N % 320 != 0)) return 1;
    if (wide == 5 && geglu && d->N % 128 != 0) return 1;
    if (wide == 6 && (!geglu || d->N % 320 != 0)) return 1;
    if (wide == 7 && (!geglu || d->N % 256 != 0)) return 1;
    if ((wide == 8 || wide == 10) && (geglu || split_t || d->N % 160 != 0)) return 1;
    if (wide == 9 && (geglu || split_t)) return 1;
    if (wide == 12 && (geglu || split_t || d->N % 128 != 0)) return 1;        // 256 x 128, 8 waves as 4 x 2 (tile 15): standard epilogue
    if (wide == 13 && (geglu || d->N % 128 != 0 || d->c3 > 0)) return 1;   // 64 x 128, 4 waves as 1 x 4 (tile 16): up to three workgroups per CU
    if (wide == 14 && (geglu || d->N % 64 != 0 || d->c3 > 0)) return 1;    // 128 x 64, 4 waves as 2 x 2 (tile 17)
    if (wide > 14) return 1;
  }
  if (wide == 2 && (d->N % 256 != 0 || d->taps != 1 || split_t || splits > 1)) return 1;
  if (wide == 3 && (d->N % 320 != 0 || d->taps != 1 || split_t || splits > 1)) return 1;
  if (d->ln_colsum != nullptr && (wide < 4 || d->taps != 1 || d->c2 != 0 || splits > 1)) return 1;   // folded LayerNorm: whole-line tiles only
  if (d->gn_partials != nullptr && !((wide == 4 || wide == 8 || wide == 10 || wide == 11) && !geglu && !split_t && splits <= 1)) return 1;   // GroupNorm partials: staged standard epilogue
  Gemm3Dev p;
  bool fused = false;
  p.ln_cs = (const float*)d->ln_colsum;
  p.ln_eps = d->ln_eps;
  p.a1 = (const half_t*)d->a1;
  p.a2 = (const half_t*)d->a2;
  p.wt = (const half_t*)d->wt;
  p.zeros = (const half_t*)d->zeros;
  p.bias = (const float*)d->bias;
  p.rowbias = (const half_t*)d->rowbias;
  p.residual = (const half_t*)d->residual;
  p.out = (half_t*)d->out;
  p.out2 = (half_t*)d->out2;
  p.split_col = d->split_col;
  p.ld_out2 = d->ld_out2;
  p.M = d->M;
  p.N = d->N;
  p.K = d->K;
  p.kpad = d->kpad;
  p.npad = (d->N + 127) / 128 * 128;   // rows present in the packed weight
  p.c1 = d->c1;
  p.c2 = d->c2;
  p.lda1 = d->lda1 ? d->lda1 : d->c1;
  p.lda2 = d->lda2 ? d->lda2 : d->c2;
  p.gn_ws = (float*)d->gn_partials;
  p.gn_cpg = d->gn_cpg;
  p.a3 = (const half_t*)d->a3;
  p.a4 = (const half_t*)d->a4;
  p.c3 = d->c3;
  p.c4 = d->c4;
  p.lda3 = d->lda3 ? d->lda3 : d->c3;
  p.lda4 = d->lda4 ? d->lda4 : d->c4;
  p.H = d->H;
  p.W = d->W;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.HoWo = d->Ho * d->Wo;
  p.stride = d->stride ? d->stride : 1;
  p.upsample = d->upsample;
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.ld_rowbias = d->ld_rowbias;
  p.act = d->act == AF_ACT_SILU ? 1 : (d->act == AF_ACT_QUICKGELU ? 3 : 0);
  p.ld_out = d->ld_out ? d->ld_out : (geglu ? d->N / 2 : (split_t ? d->split_col : d->N));
  p.tiles_n = 0;
  const int nk = p.kpad / BK3;
  p.splits = splits > 1 ? splits : 1;
  if (p.splits > nk) p.splits = nk;
  p.kt_per_split = (nk + p.splits - 1) / p.splits;
  p.splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
  p.ws = (float*)d->workspace;
  // in-kernel split-K reduction (af_gemm_desc.splitk_fused): the last AF_SPLITK_COUNTER_BYTES of the workspace are the tile counters
  p.counters = (d->splitk_fused && d->workspace && d->workspace_bytes > AF_SPLITK_COUNTER_BYTES)
                   ? reinterpret_cast<int*>(static_cast<char*>(d->workspace) + d->workspace_bytes - AF_SPLITK_COUNTER_BYTES) : nullptr;
  static const int wpf_env = getenv("AF_GEMM3_WPREFETCH") ? atoi(getenv("AF_GEMM3_WPREFETCH")) : 4;
  static const bool wpf_dynamic = getenv("AF_GEMM3_ABLATE_DYNAMIC") != nullptr;
  p.wpf = wpf_dynamic ? (getenv("AF_GEMM3_WPREFETCH") ? atoi(getenv("AF_GEMM3_WPREFETCH")) : 4) : wpf_env;
  p.wpf_coop = 1;
  static const int ablate = getenv("AF_GEMM3_ABLATE") ? atoi(getenv("AF_GEMM3_ABLATE")) : 0;
  static const bool ablate_dynamic = getenv("AF_GEMM3_ABLATE_DYNAMIC") != nullptr;   // experiments: re-read per call (in-process A/B)
  p.ablate = ablate_dynamic ? (getenv("AF_GEMM3_ABLATE") ? atoi(getenv("AF_GEMM3_ABLATE")) : 0) : ablate;
  {
    // the whole-line kernel's loaders hold 32-bit byte offsets into the row-addressed operands (round 6): anything larger goes to the register-staged kernel
    const long ldmax = std::max(std::max((long)(d->lda1 ? d->lda1 : d->c1), (long)(d->lda2 ? d->lda2 : d->c2)), std::max((long)(d->lda3 ? d->lda3 : d->c3), (long)(d->lda4 ? d->lda4 : d->c4)));
    if (wide >= 4 && wide != 11 && (long)d->M * ldmax * 2 >= (1L << 32)) return 1;
  }
  {
    const int ncols = geglu ? d->N / 2 : d->N;
    p.stage_ok = ncols % 8 == 0 && p.ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
  }
  if (wide == 11) {                                            // split-K slices are ranges of 64-channel chunks (nine taps each)
    const int nchunk = conv3h_chunks(d);
    p.splits = splits > 1 ? splits : 1;
    if (p.splits > nchunk) p.splits = nchunk;
    p.kt_per_split = (nchunk + p.splits - 1) / p.splits;
    p.splits = (nchunk + p.kt_per_split - 1) / p.kt_per_split;
    if (halo_variant == 3) {
      if (!p.stage_ok) return 1;                                 // (the patch form leaves through the staged epilogue only)
      p.splits = 1;
    }
    fused = launch_conv3h(p, stream, halo_r5, halo_variant);
    return (p.splits > 1 && !fused) ? 2 : 0;
  }
  if (wide >= 4) {
    // whole-line variants (64-wide K stages): 4 = 128 x 320 (GEGLU: 128 x 256), 5 = 128 x 128 (4 waves), 6 / 7 = GEGLU 256 x 320 / 256 x 256
    const int nk64 = p.kpad / 64;
    p.splits = splits > 1 ? splits : 1;
    if (p.splits > nk64) p.splits = nk64;
    p.kt_per_split = (nk64 + p.splits - 1) / p.splits;
    p.splits = (nk64 + p.kt_per_split - 1) / p.kt_per_split;
    if (wide == 4) {
      if (geglu) fused = launch3w<1, 2, 4, 4, E3_GEGLU>(p, stream);
      else if (split_t) fused = launch3w<1, 2, 4, 5, E3_SPLIT_T>(p, stream);
      else if (d->taps == 9) fused = launch3w<9, 2, 4, 5>(p, stream);
      else fused = launch3w<1, 2, 4, 5>(p, stream);
    } else if (wide == 5) {
      if (geglu) fused = launch3w<1, 2, 2, 4, E3_GEGLU>(p, stream);
      else if (split_t) fused = launch3w<1, 2, 2, 4, E3_SPLIT_T>(p, stream);
      else if (d->taps == 9) fused = launch3w<9, 2, 2, 4>(p, stream);
      else fused = launch3w<1, 2, 2, 4>(p, stream);
    } else if (wide == 8) {                                  // 128 x 160, 4 waves, two workgroups per CU
      if (d->taps == 9) fused = launch3w<9, 2, 2, 5>(p, stream); else fused = launch3w<1, 2, 2, 5>(p, stream);
    } else if (wide == 9) {                                  // 128 x 128, 4 waves, FOUR slots (three stages in flight), one workgroup per CU
      if (d->taps == 9) fused = launch3w<9, 2, 2, 4, E3_STD, 4>(p, stream); else fused = launch3w<1, 2, 2, 4, E3_STD, 4>(p, stream);
    } else if (wide == 12) {                                 // 256 x 128, 8 waves as 4 x 2, one workgroup per CU: narrow outputs over many rows (the VAE decoder's convolutions)
      if (d->taps == 9) fused = launch3w<9, 4, 2, 4>(p, stream); else fused = launch3w<1, 4, 2, 4>(p, stream);
    } else if (wide == 13) {                                 // 64 x 128, 4 waves side by side: 25 KB stages, three workgroups per CU -- short-K GEMMs whose
                                                             // step is a memory round trip live on the OTHER workgroups' MFMAs (profiles/r04y_small_tiles.txt)
      if (split_t) fused = launch3w<1, 1, 4, 2, E3_SPLIT_T>(p, stream);
      else if (d->taps == 9) fused = launch3w<9, 1, 4, 2>(p, stream); else fused = launch3w<1, 1, 4, 2>(p, stream);
    } else if (wide == 14) {                                 // 128 x 64, 4 waves as 2 x 2
      if (split_t) fused = launch3w<1, 2, 2, 2, E3_SPLIT_T>(p, stream);
      else if (d->taps == 9) fused = launch3w<9, 2, 2, 2>(p, stream); else fused = launch3w<1, 2, 2, 2>(p, stream);
    } else if (wide == 10) {                                 // 128 x 160, 4 waves, four slots
      if (d->taps == 9) fused = launch3w<9, 2, 2, 5, E3_STD, 4>(p, stream); else fused = launch3w<1, 2, 2, 5, E3_STD, 4>(p, stream);
    } else {
      if (wide == 6) fused = launch3w<1, 4, 2, 10, E3_GEGLU>(p, stream); else fused = launch3w<1, 4, 2, 8, E3_GEGLU>(p, stream);
    }
    return (p.splits > 1 && !fused) ? 2 : 0;
  }
  if (wide == 2) {
    if (geglu) fused = launch3<1, 4, 2, 8, E3_GEGLU>(p, stream); else fused = launch3<1, 4, 2, 8>(p, stream);
  } else if (wide == 3) {
    if (geglu) fused = launch3<1, 4, 2, 10, E3_GEGLU>(p, stream); else fused = launch3<1, 4, 2, 10>(p, stream);
  } else if (geglu) {
    fused = launch3<1, 2, 4, 4, E3_GEGLU>(p, stream);
  } else if (split_t) {
    if (wide) fused = launch3<1, 2, 4, 5, E3_SPLIT_T>(p, stream); else fused = launch3<1, 2, 2, 4, E3_SPLIT_T>(p, stream);
  } else if (wide) {
    if (d->taps == 9) fused = launch3<9, 2, 4, 5>(p, stream); else fused = launch3<1, 2, 4, 5>(p, stream);
  } else {
    if (d->taps == 9) fused = launch3<9, 2, 2, 4>(p, stream); else fused = launch3<1, 2, 2, 4>(p, stream);
  }
  return (p.splits > 1 && !fused) ? 2 : 0;   // 2: caller must run the split-K reduce pass with p.splits
}

int af_gemm3_effective_splits(const af_gemm_desc* d, int splits, int wide) {
  if (wide == 16) wide = 11;                                     // tile 19 = tile 14's scope
  const int nk = (wide == 11 && conv3h_eligible(d)) ? conv3h_chunks(d) : d->kpad / (wide >= 4 ? 64 : BK3);
  int s = splits > 1 ? splits : 1;
  if (s > nk) s = nk;
  const int per = (nk + s - 1) / s;
  return (nk + per - 1) / per;
}

// Fused LayerNorm -> GEGLU projection -> output projection (+ bias, + residual) of a transformer block's feed-forward at C = 320
// (af_ff320_kernel above).  w1 / b1 / ln_colsum as a GEGLU GEMM with a folded LayerNorm would take them (ln_colsum may be NULL: x is
// then used as it is); w2 the packed second projection; out = residual + b2 + W2 (v * gelu(g)).
extern "C" int af_ff_fused(const void* x, const void* w1, const void* b1, const void* ln_colsum, float ln_eps, int kpad1, const void* w2, const void* b2,
                           int kpad2, const void* residual, void* out, int M, int C, int inner, const void* zeros, void* stream) {
  AF_REQUIRE(x && w1 && b1 && w2 && out && zeros, "af_ff_fused: null pointer");
  AF_SUPPORTED(C == FF_C, "af_ff_fused: built for C = 320 (the 64 x 64 level of SD-1.5)");
  AF_REQUIRE(M > 0 && inner > 0 && inner % FF_HC == 0, "af_ff_fused: inner must be a positive multiple of 64");
  AF_REQUIRE(kpad1 >= C && kpad1 % 64 == 0 && kpad2 >= inner && kpad2 % 64 == 0, "af_ff_fused: weight row strides must be 64-multiples covering K");
  FfDev p{};
  p.x = (const half_t*)x;
  p.w1 = (const half_t*)w1;
  p.b1 = (const float*)b1;
  p.cs1 = ln_colsum ? (const float*)ln_colsum : (const float*)b1;
  p.ln_on = ln_colsum != nullptr;
  p.w2 = (const half_t*)w2;
  p.zeros = (const half_t*)zeros;
  p.M = M;
  p.inner = inner;
  p.kpad1 = kpad1;
  p.kpad2 = kpad2;
  p.ln_eps = ln_eps;
  Gemm3Dev& e = p.epi;
  e = Gemm3Dev{};
  e.bias = (const float*)b2;
  e.residual = (const half_t*)residual;
  e.out = (half_t*)out;
  e.M = M;
  e.N = C;
  e.K = inner;
  e.ld_out = C;
  e.splits = 1;
  e.rows_per_batch = M;
  e.stage_ok = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  AfLaunchScope scope(AF_FAM_GEMM, stream);
  static bool attr_set = false;
  const bool lds_ok = af_allow_dyn_lds(reinterpret_cast<const void*>(&af_ff320_kernel<false>), FF_LDS + AF_WPF_DUMP_BYTES, attr_set, "af_gemm");
  if (lds_ok) hipLaunchKernelGGL(af_ff320_kernel<false>, dim3((M + FF_BM - 1) / FF_BM), dim3(512), FF_LDS + AF_WPF_DUMP_BYTES, (hipStream_t)stream, p);
  return af_check_launch("af_ff_fused");
}

// The feed-forward with the SpatialTransformer's proj_out + residual behind it (af_ff320_kernel<true>; attention.py:287-304: x = proj_out(blocks(proj_in(norm(x)))) + x_in):
//     x3 = residual + b2 + W2 (v * gelu(g))      (what af_ff_fused writes; here it stays in LDS)
//     out = x_in + b_p + W_p x3
// gn_partials / gn_cpg / rows_per_batch: GroupNorm partial statistics of `out` as af_gemm_desc.gn_partials leaves them ([M / rows_per_batch][128][32][2] floats; NULL = none).
extern "C" int af_ff_chain(const void* x, const void* w1, const void* b1, const void* ln_colsum, float ln_eps, int kpad1, const void* w2, const void* b2, int kpad2,
                           const void* residual, const void* wp, const void* bp, int kpad_p, const void* x_in, void* out, void* gn_partials, int gn_cpg,
                           int rows_per_batch, int M, int C, int inner, const void* zeros, void* stream) {
  AF_REQUIRE(x && w1 && b1 && w2 && wp && out && zeros, "af_ff_chain: null pointer");
  AF_SUPPORTED(C == FF_C, "af_ff_chain: built for C = 320 (the 64 x 64 level of SD-1.5)");
  AF_REQUIRE(M > 0 && inner > 0 && inner % FF_HC == 0, "af_ff_chain: inner must be a positive multiple of 64");
  AF_REQUIRE(kpad1 >= C && kpad1 % 64 == 0 && kpad2 >= inner && kpad2 % 64 == 0 && kpad_p >= C && kpad_p % 64 == 0, "af_ff_chain: weight row strides must be 64-multiples covering K");
  AF_REQUIRE((((uintptr_t)out | (uintptr_t)residual | (uintptr_t)x_in | (uintptr_t)b2 | (uintptr_t)bp) & 15) == 0, "af_ff_chain: out / residual / x_in / b2 / bp must be 16-byte aligned");
  if (gn_partials != nullptr) {
    AF_REQUIRE(gn_cpg > 0 && gn_cpg % 2 == 0 && C % gn_cpg == 0 && C / gn_cpg <= 32 && rows_per_batch > 0 && rows_per_batch % FF_BM == 0 && M % rows_per_batch == 0 &&
                   rows_per_batch / 128 <= 128,
               "af_ff_chain: gn_partials needs an even gn_cpg dividing C into at most 32 groups and whole 128-row blocks per batch item (at most 128)");
  }
  FfDev p{};
  p.x = (const half_t*)x;
  p.w1 = (const half_t*)w1;
  p.b1 = (const float*)b1;
  p.cs1 = ln_colsum ? (const float*)ln_colsum : (const float*)b1;
  p.ln_on = ln_colsum != nullptr;
  p.w2 = (const half_t*)w2;
  p.zeros = (const half_t*)zeros;
  p.M = M;
  p.inner = inner;
  p.kpad1 = kpad1;
  p.kpad2 = kpad2;
  p.ln_eps = ln_eps;
  p.wp = (const half_t*)wp;
  p.kpad_p = kpad_p;
  p.epi = Gemm3Dev{};
  p.epi.bias = (const float*)b2;
  p.epi.residual = (const half_t*)residual;
  Gemm3Dev& e = p.epi_p;
  e = Gemm3Dev{};
  e.bias = (const float*)bp;
  e.residual = (const half_t*)x_in;
  e.out = (half_t*)out;
  e.M = M;
  e.N = C;
  e.K = C;
  e.ld_out = C;
  e.splits = 1;
  e.rows_per_batch = gn_partials ? rows_per_batch : M;
  e.stage_ok = 1;
  e.gn_ws = (float*)gn_partials;
  e.gn_cpg = gn_cpg;
  AfLaunchScope scope(AF_FAM_GEMM, stream);
  static bool attr_set = false;
  const bool lds_ok = af_allow_dyn_lds(reinterpret_cast<const void*>(&af_ff320_kernel<true>), FF_LDS + AF_WPF_DUMP_BYTES, attr_set, "af_gemm");
  if (lds_ok) hipLaunchKernelGGL(af_ff320_kernel<true>, dim3((M + FF_BM - 1) / FF_BM), dim3(512), FF_LDS + AF_WPF_DUMP_BYTES, (hipStream_t)stream, p);
  return af_check_launch("af_ff_chain");
}
