// af_attn.hip -- fused attention core  O = softmax(Q K^T * scale + keybias) V  on MFMA,
// without materialising the [b*h, N, L] score tensor (attention.py:181-202 does).
//
// Mapping (gfx950, wave64, v_mfma_f32_32x32x16_f16):
//   * a workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries and
//     keeps the query on the MFMA *lane* from the first product to the store:
//       S^T[key, query] = K . Q^T          (A = K tile rows from LDS, B = Q fragments in registers)
//       O^T[d,   query] = V^T . P^T        (A = V^T rows from LDS,   B = exp(S^T) straight from
//                                           the accumulator registers -- no LDS round trip, no
//                                           cross-lane shuffle: the accumulator's row index is
//                                           the next product's k index)
//     so the online-softmax state (running max m, running sum l) is one scalar per lane, the
//     two half-waves (which hold different keys of the same query) exchange one value per
//     64-key stage, and rescaling O is a per-lane multiply;
//   * head dims 40 / 80 / 160 (SD-1.5: C/8) are consumed in k-steps of 16 (d padded to 48 / 80 /
//     160) and produce ceil(d/32) 32-row O^T tiles;
//   * V arrives TRANSPOSED ([B, C, L], key index contiguous -- written that way by the
//     projection GEMM's epilogue), so V^T fragments are two 8-byte LDS reads; LDS rows are
//     padded (K: +16 B, V^T: 72 B) to keep ds_read_b128 / ds_read_b64 conflict-free;
//   * K / V^T stages of 64 keys are double-buffered: global loads of stage t+1 are issued before
//     the MFMAs of stage t and written to LDS after them (one barrier per stage);
//   * masked keys (img_mask, attention.py:185-194) and the padding of L up to a stage boundary
//     are both expressed as an additive per-key bias (-FLT_MAX / excluded), which reproduces
//     masked_fill_(~mask, -finfo.max) exactly, including the all-masked (uniform) row.
#include <float.h>
#include <stdlib.h>

#include "af_common.h"

#ifndef AF_ATTN_NW_DEFAULT
#define AF_ATTN_NW_DEFAULT 8
#endif

namespace {

struct AttnArgs {
  const half_t* q;
  const half_t* k;
  const half_t* vt;
  half_t* o;
  const float* kbias;
  float* lse2;
  int ld_lse;
  int causal_m;  // > 0: key j is visible to query i iff j / causal_m <= i (CLIP causal mask, m keys per token)
  int B, Nq, L, heads, d;
  int ldq, ldk, ldo, ldv, ldb;
  long vbs;  // elements between consecutive batch items of vt (heads*d*ldv when vt is its own [B, C, ldv] tensor)
  float c;  // scale * log2(e)
};

constexpr int KB = 64;     // keys per stage
constexpr int VST = 36;    // V^T LDS row stride in halves (72 B)
constexpr float kLazy = 8.0f;  // log2 of the largest p the mask-free kernel tolerates before it moves the softmax reference

// ONES: the padded V^T tile has a spare row (32*DT > d); its LAST row is set to 1 for every valid key, so the row
// sum l = sum_j p_ij falls out of the PV MFMA (accumulator o[DT-1][15] of the upper half-wave) instead of 32 VALU
// adds per stage, and is rescaled together with O.  At d = 40 the softmax bookkeeping, not the MFMA, bounds this kernel.
// GENERAL = false: no key bias, no causal mask and L % 64 == 0 -- every self-attention of the U-Net -- compiles to a
// stage body with no per-score work besides max / sub / exp2 / cvt.
template <int DS, bool ONES, bool GENERAL>
__global__ __launch_bounds__(256) void af_attn_kernel(AttnArgs a) {
  constexpr int DP = 16 * DS;         // padded head dim for the QK^T k-loop
  constexpr int DT = (DS + 1) / 2;    // 32-row O^T tiles
  constexpr int DV = 32 * DT;         // V^T rows staged
  constexpr int KST = DP + 8;         // K LDS row stride (halves)
  constexpr int KBUF = KB * KST;      // halves
  constexpr int VBUF = 2 * DV * VST;  // halves
  constexpr int STAGE = KBUF + VBUF;
  constexpr int KCH = KB * (DP / 8);  // 16-byte chunks of a K stage
  constexpr int NKC = (KCH + 255) / 256;
  constexpr int VCH = DV * 8;  // 16-byte chunks of a V^T stage
  constexpr int NVC = (VCH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  half_t* lds = reinterpret_cast<half_t*>(af_smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs (private 4 MB L2s).  All query
  // blocks of one (batch, head) share its K / V, so they are placed on the SAME XCD (bid % 8) and walked
  // consecutively there: each L2 then holds the K/V of the few (batch, head) groups it is working on instead of
  // every XCD streaming every group from the Infinity Cache.  Speed only; any placement is correct.
  const int qblocks = (a.Nq + 127) / 128;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bh = (idx / qblocks) * 8 + xcd;
  if (bh >= a.B * a.heads) return;
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int query = (idx % qblocks) * 128 + wave * 32 + r;


  // ---- Q fragments (B operand of S^T = K Q^T), pre-multiplied by scale*log2(e): lane (r, hh) holds
  //      Q[query][16 s + 8 hh .. +7]
  half8_t qf[DS];
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  {
    const half_t* qp = a.q + ((size_t)b * a.Nq + (query < a.Nq ? query : 0)) * a.ldq + h * a.d;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = 16 * s + 8 * hh;
      half8_t v = (query < a.Nq && dc < a.d) ? *reinterpret_cast<const half8_t*>(qp + dc) : zero8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * a.c);
      qf[s] = v;
    }
  }

  half8_t rk[NKC], rv[NVC];
  // per-thread staging sources are fixed up to the stage's key offset: hoist the address arithmetic out of the loop
  const half_t* kptr[NKC];
  bool kok[NKC];
#pragma unroll
  for (int j = 0; j < NKC; ++j) {
    const int i = tid + 256 * j;
    const int row = i / (DP / 8), ch = i - row * (DP / 8);
    kok[j] = i < KCH && ch * 8 < a.d;
    kptr[j] = a.k + ((size_t)b * a.L + row) * a.ldk + h * a.d + ch * 8;
  }
  const half_t* vptr[NVC];
  bool vok[NVC], vones[NVC];
#pragma unroll
  for (int j = 0; j < NVC; ++j) {
    const int i = tid + 256 * j;
    const int row = i >> 3, ch = i & 7;
    vok[j] = i < VCH && row < a.d;
    vones[j] = ONES && i < VCH && row == DV - 1;
    vptr[j] = a.vt + (size_t)b * a.vbs + (size_t)(h * a.d + (row < a.d ? row : 0)) * a.ldv + ch * 8;
  }
  const half8_t ones8 = {1, 1, 1, 1, 1, 1, 1, 1};
  auto load_stage = [&](int key0) {
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
      const int row = (tid + 256 * j) / (DP / 8);
      const bool ok = kok[j] && (!GENERAL || key0 + row < a.L);
      rk[j] = ok ? *reinterpret_cast<const half8_t*>(kptr[j] + (size_t)key0 * a.ldk) : zero8;
    }
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
      const int kk = key0 + ((tid + 256 * j) & 7) * 8;
      const bool ok = vok[j] && (!GENERAL || kk < a.L);
      // The fill value must not depend on the loaded data: any VALU touch of the destination right after the load
      // makes the compiler wait for it here (vmcnt(0)) and the prefetch turns synchronous.  Tail masking of a
      // partially valid chunk (L % 8 != 0) therefore happens in store_stage, after the stage's compute.
      half8_t fill = zero8;
      if (ONES) {
        if (GENERAL) {
          if (vones[j]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fill[e] = (kk + e < a.L) ? (half_t)1 : (half_t)0;
          }
        } else {
          fill = vones[j] ? ones8 : zero8;
        }
      }
      rv[j] = ok ? *reinterpret_cast<const half8_t*>(vptr[j] + key0) : fill;
    }
  };
  auto store_stage = [&](int buf, int key0_store) {
    half_t* Ks = lds + buf * STAGE;
    half_t* Vs = Ks + KBUF;
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
      const int i = tid + 256 * j;
      if (i < KCH) {
        const int row = i / (DP / 8), ch = i - row * (DP / 8);
        *reinterpret_cast<half8_t*>(Ks + row * KST + ch * 8) = rk[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
      const int i = tid + 256 * j;
      if (i < VCH) {
        const int row = i >> 3, ch = i & 7;
        half_t* dst = Vs + ((ch >> 2) * DV + row) * VST + (ch & 3) * 8;
        if (GENERAL) {   // keys >= L inside a partially valid chunk carry whatever the padding holds: zero them
          const int kk = key0_store + ch * 8;
          if (kk + 8 > a.L) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (kk + e >= a.L) rv[j][e] = (half_t)0;
          }
        }
        const half4_t lo = {rv[j][0], rv[j][1], rv[j][2], rv[j][3]};
        const half4_t hi = {rv[j][4], rv[j][5], rv[j][6], rv[j][7]};
        *reinterpret_cast<half4_t*>(dst) = lo;
        *reinterpret_cast<half4_t*>(dst + 4) = hi;
      }
    }
  };

  floatx16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m = -FLT_MAX, l = 0.f;

  const int nstage = (a.L + KB - 1) / KB;
  load_stage(0);
  store_stage(0, 0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int key0 = st * KB;
    const bool more = st + 1 < nstage;
    if (more) load_stage(key0 + KB);

    const half_t* Ks = lds + (st & 1) * STAGE;
    const half_t* Vs = Ks + KBUF;
    const int nsub = GENERAL ? ((key0 + 32 < a.L) ? 2 : 1) : 2;

    // ---- S^T = K Q^T (already in the base-2 softmax domain) for up to two 32-key sub-tiles.
    // Mask-free kernel: the accumulator starts at -m (row constant as the MFMA's C operand), so the MFMA
    // delivers S^T - m and exp2 applies directly; only when some lane's max grew is a correction subtracted.
    floatx16 sT[2];
    const float cinit = (GENERAL || st == 0) ? 0.f : -m;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sT[sub][i] = cinit;
      if (sub < nsub) {
#pragma unroll
        for (int s = 0; s < DS; ++s) {
          const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (sub * 32 + r) * KST + 16 * s + 8 * hh);
          sT[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sT[sub], 0, 0, 0);
        }
      }
    }
    float mx = -FLT_MAX;
    if (GENERAL) {
      // ---- bias, causal / key-range mask
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (sub < nsub) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int kb = key0 + sub * 32 + 8 * g + 4 * hh;  // keys kb .. kb+3 live in regs 4g .. 4g+3
            floatx4 bias = {0.f, 0.f, 0.f, 0.f};
            if (a.kbias) bias = *reinterpret_cast<const floatx4*>(a.kbias + (size_t)b * a.ldb + kb);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = sT[sub][4 * g + e] + bias[e];
              bool vis = kb + e < a.L;
              if (a.causal_m > 0) vis = vis && ((kb + e) / a.causal_m <= query);
              t = vis ? t : -INFINITY;
              sT[sub][4 * g + e] = t;
              mx = fmaxf(mx, t);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {   // v_max3_f32 chains: 16 instructions for 32 scores
        mx = fmaxf(fmaxf(mx, sT[0][i]), sT[0][i + 1]);
        mx = fmaxf(fmaxf(mx, sT[1][i]), sT[1][i + 1]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (GENERAL) {
      // ---- rescale only when some lane's running max grew (wave-uniform branch; exact, not a threshold)
      if (__builtin_amdgcn_ballot_w64(mx > m) != 0) {
        const float m_new = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        if (!ONES) l *= alpha;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
      }
      sT[0] = sT[0] - m;
      sT[1] = sT[1] - m;
    } else if (st == 0) {
      m = mx;                       // first stage: scores are absolute, O and l are still zero
      sT[0] = sT[0] - m;
      sT[1] = sT[1] - m;
    } else if (__builtin_amdgcn_ballot_w64(mx > kLazy) != 0) {
      // LAZY reference: m is a softmax reference, not necessarily the running max.  It is only moved when some lane's new
      // scores exceed it by more than 2^kLazy; until then p = exp2(s - m) may be as large as 2^kLazy, which fp16 P and the
      // fp32 accumulators hold without loss, and the final O / l is unchanged (softmax is shift invariant).  With the exact
      // running max the wave-uniform branch below was taken on ~70 % of the stages for random data (any of 32 queries
      // seeing a new maximum): 32 v_sub + 16 v_pk_mul of the 151 VALU instructions per stage of a VALU-issue-bound kernel.
      const float delta = fmaxf(mx, 0.f);   // scores are relative to m: these lanes move their reference up by delta
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m += delta;
      if (!ONES) l *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
      sT[0] = sT[0] - delta;
      sT[1] = sT[1] - delta;
    }

    // ---- P^T = exp2(S^T - m): accumulator registers become the B operand of the PV product
    half8_t pf[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub < nsub) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float p = __builtin_amdgcn_exp2f(sT[sub][8 * s2 + j]);
            if (!ONES) l += p;
            pf[sub][s2][j] = (half_t)p;
          }
        }
      } else {
        pf[sub][0] = zero8;
        pf[sub][1] = zero8;
      }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (sub < nsub) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const half_t* vp = Vs + (sub * DV + 32 * t + r) * VST + 16 * s2 + 4 * hh;
            const half4_t lo = *reinterpret_cast<const half4_t*>(vp);
            const half4_t hi = *reinterpret_cast<const half4_t*>(vp + 8);
            const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][s2], o[t], 0, 0, 0);
          }
        }
      }
    }

    if (more) store_stage((st + 1) & 1, key0 + KB);
    __syncthreads();
  }

  // ---- normalise and store: lane holds d rows {32t + 8g + 4hh + e} of its query
  if (ONES) {
    // row DV-1 of O^T (the ones row) lives in register 15 of the last tile on the upper half-wave
    const float mine = o[DT - 1][15];
    const float other = __shfl_xor(mine, 32, 64);
    l = hh ? mine : other;
  } else {
    l += __shfl_xor(l, 32, 64);
  }
  const float inv = 1.0f / l;
  if (a.lse2 && query < a.Nq && hh == 0) a.lse2[((size_t)b * a.heads + h) * a.ld_lse + query] = m + __builtin_amdgcn_logf(l);
  if (query < a.Nq) {
    half_t* op = a.o + ((size_t)b * a.Nq + query) * a.ldo + h * a.d;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * t + 8 * g + 4 * hh;
        if (dd < a.d) {
          const half4_t v = {(half_t)(o[t][4 * g + 0] * inv), (half_t)(o[t][4 * g + 1] * inv),
                             (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
          *reinterpret_cast<half4_t*>(op + dd) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Two-chain variant of the mask-free flash kernel (no key bias, no causal mask, L % 64 == 0: the U-Net's self-attention).
// PMC (profiles/r01c, r01f): the one-chain kernel is VALU-issue bound at d = 40 and, inside a wave, its MFMA and VALU phases
// alternate (S^T MFMAs -> softmax VALU -> PV MFMAs), so the two pipes overlap only across waves.  Here every wave carries
// TWO independent 32-query chains (a workgroup = 256 queries) and orders a stage as
//     S^T(A), S^T(B) [MFMA]  ->  softmax(A) [VALU, under S^T(B)]  ->  PV(A) [MFMA]  ->  softmax(B) [VALU, under PV(A)]  ->  PV(B)
// so an MFMA batch of one chain is always in flight under the other chain's softmax, and the K / V^T fragments read from LDS are
// shared by both chains (half the LDS reads per query: the LDS pipe was ~54 % busy).
// SLOT (round 4; needs d = 16 DS - 8, i.e. d = 40): the softmax reference rides in a spare k slot of the S^T product -- K column d holds 1.0, element d of
// the Q fragment holds -m as fp16 (any per-query constant cancels in the normalisation as long as every key sees the same one) -- so the S^T
// accumulators start from the inline constant 0 instead of 32 v_mov of -m per chain and stage; the half-wave maximum is exchanged by
// v_permlane32_swap instead of a ds_bpermute round trip.
// NW (round 6): waves per workgroup.  4 = rounds 2 - 5: 256 queries per workgroup, two workgroups per CU, so the two waves of a SIMD belong to DIFFERENT
// workgroups and nothing orders them -- they settle into lock step (both want the matrix pipe for their S^T / P.V batches, then both issue their exp2
// blocks), and a stage costs the SUM of its matrix and vector cycles (profiles/r04c).  8 = one workgroup of 512 queries per CU whose waves w and w + 4
// share a SIMD, the upper half at a STATIC raised priority (s_setprio 1 once, no flips: MI355X_MICROARCH.md, two waves per SIMD, item 4): when both
// partners want the matrix pipe the upper wave takes it and the lower one gets it while the upper one is in its exp2 block, so the partners run out of
// phase by construction; K / V^T tiles are staged once per 512 queries.  Same arithmetic per chain in the same order: bit-identical results.
template <int DS, bool ONES, bool SLOT = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void af_attn2_kernel(AttnArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr int KST = DP + 8, KBUF = KB * KST, VBUF = 2 * DV * VST, STAGE = KBUF + VBUF;
  constexpr int KCH = KB * (DP / 8), NKC = (KCH + NT - 1) / NT, VCH = DV * 8, NVC = (VCH + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  half_t* lds = reinterpret_cast<half_t*>(af_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int qblocks = (a.Nq + NT - 1) / NT;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bh = (idx / qblocks) * 8 + xcd;
  if (bh >= a.B * a.heads) return;
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int q0 = (idx % qblocks) * NT + wave * 64 + r;          // chain c handles query q0 + 32 c
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  half8_t qf[2][DS];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int query = q0 + 32 * c;
    const half_t* qp = a.q + ((size_t)b * a.Nq + (query < a.Nq ? query : 0)) * a.ldq + h * a.d;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = 16 * s + 8 * hh;
      half8_t v = (query < a.Nq && dc < a.d) ? *reinterpret_cast<const half8_t*>(qp + dc) : zero8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * a.c);
      qf[c][s] = v;
    }
  }

  half8_t rk[NKC], rv[NVC];
  const half_t* kptr[NKC];
  bool kok[NKC], kone[NKC];
#pragma unroll
  for (int j = 0; j < NKC; ++j) {
    const int i = tid + NT * j;
    const int row = i / (DP / 8), ch = i - row * (DP / 8);
    kok[j] = i < KCH && ch * 8 < a.d;
    kone[j] = SLOT && i < KCH && ch * 8 == a.d;      // K column d = 1.0: multiplies the -m slot of Q
    kptr[j] = a.k + ((size_t)b * a.L + row) * a.ldk + h * a.d + ch * 8;
  }
  const half8_t one1 = {1, 0, 0, 0, 0, 0, 0, 0};
  const half_t* vptr[NVC];
  bool vok[NVC], vones[NVC];
#pragma unroll
  for (int j = 0; j < NVC; ++j) {
    const int i = tid + NT * j;
    const int row = i >> 3, ch = i & 7;
    vok[j] = i < VCH && row < a.d;
    vones[j] = ONES && i < VCH && row == DV - 1;
    vptr[j] = a.vt + (size_t)b * a.vbs + (size_t)(h * a.d + (row < a.d ? row : 0)) * a.ldv + ch * 8;
  }
  const half8_t ones8 = {1, 1, 1, 1, 1, 1, 1, 1};
  auto load_stage = [&](int key0) {
#pragma unroll
    for (int j = 0; j < NKC; ++j) rk[j] = kok[j] ? *reinterpret_cast<const half8_t*>(kptr[j] + (size_t)key0 * a.ldk) : (kone[j] ? one1 : zero8);
#pragma unroll
    for (int j = 0; j < NVC; ++j) rv[j] = vok[j] ? *reinterpret_cast<const half8_t*>(vptr[j] + key0) : (vones[j] ? ones8 : zero8);
  };
  auto store_stage = [&](int buf) {
    half_t* Ks = lds + buf * STAGE;
    half_t* Vs = Ks + KBUF;
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
      const int i = tid + NT * j;
      if (i < KCH) {
        const int row = i / (DP / 8), ch = i - row * (DP / 8);
        *reinterpret_cast<half8_t*>(Ks + row * KST + ch * 8) = rk[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
      const int i = tid + NT * j;
      if (i < VCH) {
        const int row = i >> 3, ch = i & 7;
        half_t* dst = Vs + ((ch >> 2) * DV + row) * VST + (ch & 3) * 8;
        const half4_t lo = {rv[j][0], rv[j][1], rv[j][2], rv[j][3]};
        const half4_t hi = {rv[j][4], rv[j][5], rv[j][6], rv[j][7]};
        *reinterpret_cast<half4_t*>(dst) = lo;
        *reinterpret_cast<half4_t*>(dst + 4) = hi;
      }
    }
  };

  floatx16 o[2][DT];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[c][t][i] = 0.f;
  float m[2] = {-FLT_MAX, -FLT_MAX}, l[2] = {0.f, 0.f};

  // softmax of one chain's 64 scores of this stage (scores arrive relative to m[c] except in stage 0) -> P^T fragments
  auto softmax = [&](int c, int st, floatx16 (&sT)[2], half8_t (&pf)[2][2]) {
    float mx = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      mx = fmaxf(fmaxf(mx, sT[0][i]), sT[0][i + 1]);
      mx = fmaxf(fmaxf(mx, sT[1][i]), sT[1][i + 1]);
    }
    if constexpr (SLOT) {
      const unsigned mb = __builtin_bit_cast(unsigned, mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
      mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    } else {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    }
    if (st == 0) {
      const float m0 = SLOT ? (float)(half_t)mx : mx;              // SLOT: the reference is what fp16 holds
      m[c] = m0;
      sT[0] = sT[0] - m0;
      sT[1] = sT[1] - m0;
      if (SLOT && hh) qf[c][DS - 1][0] = (half_t)(-m0);
    } else if (__builtin_amdgcn_ballot_w64(mx > kLazy) != 0) {       // lazy reference, see af_attn_kernel
      float delta = fmaxf(mx, 0.f);
      if constexpr (SLOT) {
        const float mn = (float)(half_t)(m[c] + delta);
        delta = mn - m[c];
        if (hh) qf[c][DS - 1][0] = (half_t)(-mn);
      }
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m[c] += delta;
      if (!ONES) l[c] *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[c][t][i] *= alpha;
      sT[0] = sT[0] - delta;
      sT[1] = sT[1] - delta;
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(sT[sub][8 * s2 + j]);
          if (!ONES) l[c] += p;
          pf[sub][s2][j] = (half_t)p;
        }
  };

  const int nstage = a.L / KB;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  if (NW == 8 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);      // the SIMD partners of waves 0 - 3: static priority
  for (int st = 0; st < nstage; ++st) {
    const bool more = st + 1 < nstage;
    if (more) load_stage((st + 1) * KB);
    const half_t* Ks = lds + (st & 1) * STAGE;
    const half_t* Vs = Ks + KBUF;

    // ---- S^T of both chains from shared K fragments
    floatx16 sA[2], sB[2];
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (!SLOT) {
      const float ca = st == 0 ? 0.f : -m[0], cb = st == 0 ? 0.f : -m[1];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sA[sub][i] = ca;
          sB[sub][i] = cb;
        }
      }
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (sub * 32 + r) * KST + 16 * s + 8 * hh);
        sA[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0][s], (SLOT && s == 0) ? zero16 : sA[sub], 0, 0, 0);
      }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (sub * 32 + r) * KST + 16 * s + 8 * hh);
        sB[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[1][s], (SLOT && s == 0) ? zero16 : sB[sub], 0, 0, 0);
      }
    // ---- chain A softmax (its scores are done first; chain B's MFMAs are still running)
    half8_t pA[2][2], pB[2][2];
    softmax(0, st, sA, pA);
    // ---- PV(A) in flight under chain B's softmax
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half_t* vp = Vs + (sub * DV + 32 * t + r) * VST + 16 * s2 + 4 * hh;
          const half4_t lo = *reinterpret_cast<const half4_t*>(vp);
          const half4_t hi = *reinterpret_cast<const half4_t*>(vp + 8);
          const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pA[sub][s2], o[0][t], 0, 0, 0);
        }
    softmax(1, st, sB, pB);
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half_t* vp = Vs + (sub * DV + 32 * t + r) * VST + 16 * s2 + 4 * hh;
          const half4_t lo = *reinterpret_cast<const half4_t*>(vp);
          const half4_t hi = *reinterpret_cast<const half4_t*>(vp + 8);
          const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pB[sub][s2], o[1][t], 0, 0, 0);
        }
    if (more) store_stage((st + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int query = q0 + 32 * c;
    float lc = l[c];
    if (ONES) {
      const float mine = o[c][DT - 1][15];
      const float other = __shfl_xor(mine, 32, 64);
      lc = hh ? mine : other;
    } else {
      lc += __shfl_xor(lc, 32, 64);
    }
    const float inv = 1.0f / lc;
    if (a.lse2 && query < a.Nq && hh == 0) a.lse2[((size_t)b * a.heads + h) * a.ld_lse + query] = m[c] + __builtin_amdgcn_logf(lc);
    if (query < a.Nq) {
      half_t* op = a.o + ((size_t)b * a.Nq + query) * a.ldo + h * a.d;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 32 * t + 8 * g + 4 * hh;
          if (dd < a.d) {
            const half4_t v = {(half_t)(o[c][t][4 * g + 0] * inv), (half_t)(o[c][t][4 * g + 1] * inv),
                               (half_t)(o[c][t][4 * g + 2] * inv), (half_t)(o[c][t][4 * g + 3] * inv)};
            *reinterpret_cast<half4_t*>(op + dd) = v;
          }
        }
    }
  }
}

template <int DS, bool ONES, bool SLOT = false>
int launch_attn2chain(const AttnArgs& a, hipStream_t stream) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr size_t lds = (size_t)2 * (KB * (DP + 8) + 2 * DV * VST) * sizeof(half_t);
  const int bh8 = (a.B * a.heads + 7) / 8 * 8;
  // AF_ATTN_NW (read per call: the tests compare both forms in one process): 8 = one 512-query workgroup per CU with the static-priority SIMD partners
  const char* nw_s = getenv("AF_ATTN_NW");
  const int nw = nw_s ? atoi(nw_s) : AF_ATTN_NW_DEFAULT;
  if (nw == 8 && SLOT && a.Nq % 512 == 0) {          // (the SLOT instantiations fit 256 registers at 512 threads; the others would spill)
    static bool attr8 = false;
    if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_attn2_kernel<DS, ONES, SLOT, 8>), lds, attr8, "af_attention")) return af_check_launch("af_attention");
    hipLaunchKernelGGL((af_attn2_kernel<DS, ONES, SLOT, 8>), dim3((a.Nq / 512) * bh8), dim3(512), lds, stream, a);
    return af_check_launch("af_attention(two-chain, 8 waves)");
  }
  static bool attr_set = false;  // benign race: idempotent attribute
  // AF_ATTN_LDS_PAD (experiments only): extra dynamic LDS per workgroup, e.g. 90000 leaves room for ONE 4-wave workgroup per CU (one wave per SIMD)
  static const size_t pad = getenv("AF_ATTN_LDS_PAD") ? (size_t)atol(getenv("AF_ATTN_LDS_PAD")) : 0;
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_attn2_kernel<DS, ONES, SLOT>), lds + pad, attr_set, "af_attention")) return af_check_launch("af_attention");
  const int qblocks = (a.Nq + 255) / 256;
  hipLaunchKernelGGL((af_attn2_kernel<DS, ONES, SLOT>), dim3(qblocks * bh8), dim3(256), lds + pad, stream, a);
  return af_check_launch("af_attention(two-chain)");
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined form of the two-chain SLOT kernel above (round 5).  The ISA of af_attn2_kernel shows what its source order
// cannot avoid: the lazy-reference branch of a chain sits between that chain's maximum and its 32 exp2, so every exp2 block lands in
// a basic block whose only MFMAs (the chain's own P.V) DEPEND on it -- per stage the wave issues 12 S^T MFMAs, 32 exp2 with the matrix
// pipe idle, 8 P.V MFMAs, 32 exp2, 8 P.V MFMAs, and the SIMD's two waves (VALU issue is arbitrated by age, MI355X_MICROARCH.md "Two
// waves per SIMD") do not fill each other's holes: SIMD time per wave and stage equals the SUM of its matrix and vector time.
// Here the two chains run half a stage apart, so that each basic block holds 14 MFMAs and 32 exp2 that do not depend on each other:
//     block B(h):  S^T_0(h) [K(h)]   P.V_0(h-1) [V(h-1)]   ||  exp2 of chain 1's scores of stage h-1   -> max_0(h), reference branch
//     block A(h):  S^T_1(h) [K(h)]   P.V_1(h-1) [V(h-1)]   ||  exp2 of chain 0's scores of stage h     -> max_1(h), reference branch
// (an MFMA holds the vector issue port for 8 of its 32 cycles: 2 exp2 + 1 cvt_pk fit each gap).  A "shifted stage" h therefore reads
// K of key block h and V^T of key block h-1 from the same double-buffered LDS slot; one barrier per shifted stage as before; the
// first and the last shifted stage are peeled.  Per chain the arithmetic and its order are those of af_attn2_kernel<DS, true, true>:
// the results are bit-identical (tests/test_hip_kernels.py).  Measured 1-4 % faster alone (profiles/r05q_attn_pipe.txt), like round 4's
// af_attn3 (r04c) it does not change what bounds the kernel (vector issue per score); AF_ATTN_PIPE=1 selects it, the default is the unpipelined kernel.
// one MFMA, then the vector work its 32 cycles leave room for (24 issue cycles: 2 exp2 + 1 cvt_pk, or 5-6 plain VALU)
#define AF_ATTN_PIPE_GROUPS()                                                         \
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                  \
  _Pragma("unroll") for (int g_ = 0; g_ < 14; ++g_) {                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                \
  }
template <int DS>
__global__ __launch_bounds__(256, 2) void af_attn2p_kernel(AttnArgs a) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr int KST = DP + 8, KBUF = KB * KST, VBUF = 2 * DV * VST, STAGE = KBUF + VBUF;
  constexpr int KCH = KB * (DP / 8), NKC = (KCH + 255) / 256, VCH = DV * 8, NVC = (VCH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  half_t* lds = reinterpret_cast<half_t*>(af_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int qblocks = (a.Nq + 255) / 256;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bh = (idx / qblocks) * 8 + xcd;
  if (bh >= a.B * a.heads) return;
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int q0 = (idx % qblocks) * 256 + wave * 64 + r;          // chain c handles query q0 + 32 c
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  half8_t qf[2][DS];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int query = q0 + 32 * c;
    const half_t* qp = a.q + ((size_t)b * a.Nq + (query < a.Nq ? query : 0)) * a.ldq + h * a.d;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = 16 * s + 8 * hh;
      half8_t v = *reinterpret_cast<const half8_t*>(qp + (dc < a.d ? dc : 0));
      if (!(query < a.Nq && dc < a.d)) v = zero8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * a.c);
      qf[c][s] = v;
    }
  }

  // staging: every load is unconditional on a clamped address and the constant chunks are selected afterwards (a load under a
  // condition compiles to a branch of its own)
  half8_t rk[NKC], rv[NVC];
  const half_t* kptr[NKC];
  bool kok[NKC], kone[NKC];
#pragma unroll
  for (int j = 0; j < NKC; ++j) {
    const int i = tid + 256 * j;
    const int row = (i / (DP / 8)) & (KB - 1), ch = i % (DP / 8);
    kok[j] = i < KCH && ch * 8 < a.d;
    kone[j] = i < KCH && ch * 8 == a.d;             // K column d = 1.0: multiplies the -m slot of Q
    kptr[j] = a.k + ((size_t)b * a.L + row) * a.ldk + h * a.d + (kok[j] ? ch * 8 : 0);
  }
  const half8_t one1 = {1, 0, 0, 0, 0, 0, 0, 0};
  const half8_t ones8 = {1, 1, 1, 1, 1, 1, 1, 1};
  const half_t* vptr[NVC];
  bool vok[NVC], vones[NVC];
#pragma unroll
  for (int j = 0; j < NVC; ++j) {
    const int i = tid + 256 * j;
    const int row = i >> 3, ch = i & 7;
    vok[j] = i < VCH && row < a.d;
    vones[j] = i < VCH && row == DV - 1;
    vptr[j] = a.vt + (size_t)b * a.vbs + (size_t)(h * a.d + (row < a.d ? row : 0)) * a.ldv + ch * 8;
  }
  auto load_stage = [&](int kkey0, int vkey0) {
#pragma unroll
    for (int j = 0; j < NKC; ++j) rk[j] = *reinterpret_cast<const half8_t*>(kptr[j] + (size_t)kkey0 * a.ldk);
#pragma unroll
    for (int j = 0; j < NVC; ++j) rv[j] = *reinterpret_cast<const half8_t*>(vptr[j] + vkey0);
  };
  auto store_stage = [&](int buf) {
    half_t* Ks = lds + buf * STAGE;
    half_t* Vs = Ks + KBUF;
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
      const int i = tid + 256 * j;
      if (i < KCH) {
        const int row = i / (DP / 8), ch = i - row * (DP / 8);
        *reinterpret_cast<half8_t*>(Ks + row * KST + ch * 8) = kok[j] ? rk[j] : (kone[j] ? one1 : zero8);
      }
    }
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
      const int i = tid + 256 * j;
      if (i < VCH) {
        const int row = i >> 3, ch = i & 7;
        const half8_t v = vok[j] ? rv[j] : (vones[j] ? ones8 : zero8);
        half_t* dst = Vs + ((ch >> 2) * DV + row) * VST + (ch & 3) * 8;
        const half4_t lo = {v[0], v[1], v[2], v[3]};
        const half4_t hi = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<half4_t*>(dst) = lo;
        *reinterpret_cast<half4_t*>(dst + 4) = hi;
      }
    }
  };

  floatx16 o[2][DT];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[c][t][i] = 0.f;
  float m[2] = {-FLT_MAX, -FLT_MAX};
  const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // S^T of chain c against the stage's keys: scores relative to m[c] (the slot) except before the first reference is set
  auto qk = [&](int c, const half_t* Ks, floatx16 (&sT)[2]) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (sub * 32 + r) * KST + 16 * s + 8 * hh);
        sT[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[c][s], s == 0 ? zero16 : sT[sub], 0, 0, 0);
      }
  };
  auto pv = [&](int c, const half_t* Vs, const half8_t (&pf)[2][2]) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half_t* vp = Vs + (sub * DV + 32 * t + r) * VST + 16 * s2 + 4 * hh;
          const half4_t lo = *reinterpret_cast<const half4_t*>(vp);
          const half4_t hi = *reinterpret_cast<const half4_t*>(vp + 8);
          const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][s2], o[c][t], 0, 0, 0);
        }
  };
  auto exps = [&](const floatx16 (&sT)[2], half8_t (&pf)[2][2]) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[sub][s2][j] = (half_t)__builtin_amdgcn_exp2f(sT[sub][8 * s2 + j]);
  };
  // the chain's maximum over the stage's 64 scores and its reference: set in the first stage, moved (rarely) afterwards
  auto head = [&](int c, bool first, floatx16 (&sT)[2]) {
    float mx = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      mx = fmaxf(fmaxf(mx, sT[0][i]), sT[0][i + 1]);
      mx = fmaxf(fmaxf(mx, sT[1][i]), sT[1][i + 1]);
    }
    const unsigned mb = __builtin_bit_cast(unsigned, mx);
    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    if (first) {
      const float m0 = (float)(half_t)mx;              // the reference is what fp16 holds
      m[c] = m0;
      sT[0] = sT[0] - m0;
      sT[1] = sT[1] - m0;
      if (hh) qf[c][DS - 1][0] = (half_t)(-m0);
    } else if (__builtin_amdgcn_ballot_w64(mx > kLazy) != 0) {       // lazy reference, see af_attn_kernel
      float delta = fmaxf(mx, 0.f);
      const float mn = (float)(half_t)(m[c] + delta);
      delta = mn - m[c];
      if (hh) qf[c][DS - 1][0] = (half_t)(-mn);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m[c] += delta;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[c][t][i] *= alpha;
      sT[0] = sT[0] - delta;
      sT[1] = sT[1] - delta;
    }
  };

  const int nstage = a.L / KB;                     // key blocks; shifted stages 0 .. nstage
  load_stage(0, 0);
  store_stage(0);
  __syncthreads();

  floatx16 s0[2], s1[2];
  half8_t p0[2][2], p1[2][2];
  // ---- shifted stage 0: no P.V yet
  {
    load_stage(nstage > 1 ? KB : 0, 0);
    const half_t* Ks = lds;
    qk(0, Ks, s0);
    head(0, true, s0);
    qk(1, Ks, s1);
    exps(s0, p0);
    head(1, true, s1);
    store_stage(1);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- shifted stages 1 .. nstage-1
  for (int st = 1; st < nstage; ++st) {
    load_stage((st + 1 < nstage ? st + 1 : st) * KB, st * KB);
    const half_t* Ks = lds + (st & 1) * STAGE;
    const half_t* Vs = Ks + KBUF;
    // chain 1's scores enter the iteration through an opaque definition: without it the optimiser computes their exp2 where the scores
    // are produced -- at the end of the previous iteration, in front of the barrier, with no MFMA beside them
    asm volatile("; af_pin s1" : "+v"(s1[0]), "+v"(s1[1]));
    // block B
    qk(0, Ks, s0);
    exps(s1, p1);
    pv(0, Vs, p0);
    AF_ATTN_PIPE_GROUPS();
    asm volatile("; af_pin p1" : "+v"(p1[0][0]), "+v"(p1[0][1]), "+v"(p1[1][0]), "+v"(p1[1][1]));      // ... and are not sunk to their use either
    head(0, false, s0);
    // block A (chain 1's previous scores were consumed by block B's exp2)
    qk(1, Ks, s1);
    exps(s0, p0);
    pv(1, Vs, p1);
    AF_ATTN_PIPE_GROUPS();
    asm volatile("; af_pin p0" : "+v"(p0[0][0]), "+v"(p0[0][1]), "+v"(p0[1][0]), "+v"(p0[1][1]));
    head(1, false, s1);
    store_stage((st + 1) & 1);
    __syncthreads();
  }
  // ---- shifted stage nstage: the last key block's P.V
  {
    const half_t* Vs = lds + (nstage & 1) * STAGE + KBUF;
    exps(s1, p1);
    pv(0, Vs, p0);
    pv(1, Vs, p1);
  }

#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int query = q0 + 32 * c;
    const float mine = o[c][DT - 1][15];
    const float other = __shfl_xor(mine, 32, 64);
    const float lc = hh ? mine : other;
    const float inv = 1.0f / lc;
    if (a.lse2 && query < a.Nq && hh == 0) a.lse2[((size_t)b * a.heads + h) * a.ld_lse + query] = m[c] + __builtin_amdgcn_logf(lc);
    if (query < a.Nq) {
      half_t* op = a.o + ((size_t)b * a.Nq + query) * a.ldo + h * a.d;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 32 * t + 8 * g + 4 * hh;
          if (dd < a.d) {
            const half4_t v = {(half_t)(o[c][t][4 * g + 0] * inv), (half_t)(o[c][t][4 * g + 1] * inv),
                               (half_t)(o[c][t][4 * g + 2] * inv), (half_t)(o[c][t][4 * g + 3] * inv)};
            *reinterpret_cast<half4_t*>(op + dd) = v;
          }
        }
    }
  }
}

template <int DS>
int launch_attn2pipe(const AttnArgs& a, hipStream_t stream) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr size_t lds = (size_t)2 * (KB * (DP + 8) + 2 * DV * VST) * sizeof(half_t);
  static bool attr_set = false;  // benign race: idempotent attribute
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_attn2p_kernel<DS>), lds, attr_set, "af_attention")) return af_check_launch("af_attention");
  const int qblocks = (a.Nq + 255) / 256, bh8 = (a.B * a.heads + 7) / 8 * 8;
  hipLaunchKernelGGL((af_attn2p_kernel<DS>), dim3(qblocks * bh8), dim3(256), lds, stream, a);
  return af_check_launch("af_attention(two-chain, pipelined)");
}

// ---------------------------------------------------------------------------------------------------------------
// Short-key variant (L <= 128, no key bias, no causal mask): the U-Net's cross-attention cores (77 context tokens).
// These launches are HBM/latency bound (76 FLOP per algorithmic byte at C = 320): the flash kernel above spends its time
// re-staging the same 77 keys for every 128-query workgroup (2 x 64-key stages with barriers) on the ragged-L path.
// Here the K and V^T of one (batch, head) are staged into LDS ONCE per workgroup, then every wave streams 32-query groups
// with no further barrier: S^T for all keys at once (<= 4 sub-tiles of 32 keys), one plain softmax (no running max /
// rescale), O^T = V^T P^T, store; the next group's Q fragments are fetched under the current group's arithmetic.
template <int DS, bool ONES>
__global__ __launch_bounds__(256) void af_xattn_kernel(AttnArgs a, int groups_per_wave, int nchunk) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr int KST = DP + 8;
  constexpr int LP = 128;                     // keys staged (zero beyond L)
  constexpr int KCH = LP * (DP / 8);          // 16-byte chunks of K
  constexpr int VCH = DV * (LP / 8);          // 16-byte chunks of V^T
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  half_t* Ks = reinterpret_cast<half_t*>(af_smem);
  half_t* Vs = Ks + LP * KST;                 // [4 sub][DV][VST]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bh = (idx / nchunk) * 8 + xcd;
  if (bh >= a.B * a.heads) return;
  const int chunk = idx % nchunk;
  const int b = bh / a.heads, h = bh - b * a.heads;

  const int nsub = (a.L + 31) >> 5;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- first Q group (issued before the staging so its latency overlaps it)
  const int group0 = (chunk * 4 + wave) * groups_per_wave;          // this wave's first 32-query group
  auto load_q = [&](int grp, half8_t (&qf)[DS]) {
    const int query = grp * 32 + r;
    const half_t* qp = a.q + ((size_t)b * a.Nq + (query < a.Nq ? query : 0)) * a.ldq + h * a.d;
#pragma unroll
    for (int s2 = 0; s2 < DS; ++s2) {
      const int dc = 16 * s2 + 8 * hh;
      qf[s2] = (query < a.Nq && dc < a.d) ? *reinterpret_cast<const half8_t*>(qp + dc) : zero8;
    }
  };
  half8_t qraw[DS], qnext[DS];
  load_q(group0, qraw);

  // ---- stage K [LP][KST] and V^T [4][DV][VST] once.  All global loads are issued before the first LDS store (two
  //      phases over register arrays): a load -> store loop serialises one memory round trip per iteration, which was the
  //      whole cost of these small launches (10.9 us at N = 64).
  {
    constexpr int NKC = (KCH + 255) / 256, NVC = (VCH + 255) / 256;
    half8_t rk[NKC], rv[NVC];
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
      const int i = tid + 256 * j;
      const int row = i / (DP / 8), ch = i - row * (DP / 8);
      rk[j] = (i < KCH && row < a.L && ch * 8 < a.d)
                  ? *reinterpret_cast<const half8_t*>(a.k + ((size_t)b * a.L + row) * a.ldk + h * a.d + ch * 8) : zero8;
    }
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
      const int i = tid + 256 * j;
      const int row = i >> 4, kk = (i & 15) * 8;
      rv[j] = (i < VCH && row < a.d && kk < a.L)
                  ? *reinterpret_cast<const half8_t*>(a.vt + (size_t)b * a.vbs + (size_t)(h * a.d + row) * a.ldv + kk) : zero8;
    }
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
      const int i = tid + 256 * j;
      const int row = i / (DP / 8), ch = i - row * (DP / 8);
      if (i < KCH) *reinterpret_cast<half8_t*>(Ks + row * KST + ch * 8) = rk[j];
    }
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
      const int i = tid + 256 * j;
      if (i < VCH) {
        const int row = i >> 4, ch = i & 15, kk = ch * 8;
        half8_t v = rv[j];
        if (row < a.d) {
          if (kk < a.L && kk + 8 > a.L) {        // the padding of V^T rows beyond L holds anything: zero it
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (kk + e >= a.L) v[e] = (half_t)0;
          }
        } else if (ONES && row == DV - 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (kk + e < a.L) ? (half_t)1 : (half_t)0;
        }
        half_t* dst = Vs + ((ch >> 2) * DV + row) * VST + (ch & 3) * 8;
        const half4_t lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<half4_t*>(dst) = lo;
        *reinterpret_cast<half4_t*>(dst + 4) = hi;
      }
    }
  }
  __syncthreads();

  for (int gi = 0; gi < groups_per_wave; ++gi) {
    const int grp = group0 + gi;
    if (grp * 32 >= a.Nq) break;
    const int query = grp * 32 + r;
    if (gi + 1 < groups_per_wave) load_q(grp + 1, qnext);
    half8_t qf[DS];
#pragma unroll
    for (int s2 = 0; s2 < DS; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[s2][e] = (half_t)((float)qraw[s2][e] * a.c);

    // ---- two rolled passes over the <= 4 sub-tiles of 32 keys (S^T is recomputed in the second: these launches are
    //      latency bound, the MFMA is idle anyway, and the rolled form keeps the code ~3x smaller than four unrolled
    //      sub-tiles -- instruction fetch is part of a cold 10 us launch)
    auto score_tile = [&](int sub) {
      floatx16 sc;
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < DS; ++s2) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (sub * 32 + r) * KST + 16 * s2 + 8 * hh);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s2], sc, 0, 0, 0);
      }
      if (sub == nsub - 1) {                     // keys >= L (only the last sub-tile can hold some) -> -inf
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = sub * 32 + 8 * (i >> 2) + 4 * hh + (i & 3);
          if (key >= a.L) sc[i] = -INFINITY;
        }
      }
      return sc;
    };
    float mx = -FLT_MAX;
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      const floatx16 sc = score_tile(sub);
#pragma unroll
      for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, sc[i]), sc[i + 1]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    floatx16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float l = 0.f;
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      const floatx16 sc = score_tile(sub);
      half8_t pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = __builtin_amdgcn_exp2f(sc[8 * s2 + j] - mx);
          if (!ONES) l += pv;
          pf[s2][j] = (half_t)pv;
        }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half_t* vp = Vs + (sub * DV + 32 * t + r) * VST + 16 * s2 + 4 * hh;
          const half4_t lo = *reinterpret_cast<const half4_t*>(vp);
          const half4_t hi = *reinterpret_cast<const half4_t*>(vp + 8);
          const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s2], o[t], 0, 0, 0);
        }
    }
    if (ONES) {
      const float mine = o[DT - 1][15];
      const float other = __shfl_xor(mine, 32, 64);
      l = hh ? mine : other;
    } else {
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    if (a.lse2 && query < a.Nq && hh == 0) a.lse2[((size_t)b * a.heads + h) * a.ld_lse + query] = mx + __builtin_amdgcn_logf(l);
    if (query < a.Nq) {
      half_t* op = a.o + ((size_t)b * a.Nq + query) * a.ldo + h * a.d;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 32 * t + 8 * g + 4 * hh;
          if (dd < a.d) {
            const half4_t v = {(half_t)(o[t][4 * g + 0] * inv), (half_t)(o[t][4 * g + 1] * inv),
                               (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
            *reinterpret_cast<half4_t*>(op + dd) = v;
          }
        }
    }
    if (gi + 1 < groups_per_wave) {
#pragma unroll
      for (int s2 = 0; s2 < DS; ++s2) qraw[s2] = qnext[s2];
    }
  }
}

template <int DS, bool ONES>
int launch_xattn(const AttnArgs& a, hipStream_t stream) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr size_t lds = (size_t)(128 * (DP + 8) + 4 * DV * VST) * sizeof(half_t);
  static bool attr_set = false;  // benign race: idempotent attribute
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_xattn_kernel<DS, ONES>), lds, attr_set, "af_attention")) return af_check_launch("af_attention");
  const int groups = (a.Nq + 31) / 32;                       // 32-query groups per (batch, head)
  // a workgroup = 4 waves x groups_per_wave groups; aim at >= 512 workgroups while K / V^T staging stays amortised
  int gpw = groups / 32;                                      // N = 4096 -> 4 groups per wave, 8 workgroups per (b, h)
  gpw = gpw < 1 ? 1 : (gpw > 8 ? 8 : gpw);
  const int nchunk = (groups + 4 * gpw - 1) / (4 * gpw);
  const int bh8 = (a.B * a.heads + 7) / 8 * 8;
  hipLaunchKernelGGL((af_xattn_kernel<DS, ONES>), dim3(nchunk * bh8), dim3(256), lds, stream, a, gpw, nchunk);
  return af_check_launch("af_attention(short-key)");
}

template <int DS, bool ONES, bool GENERAL>
int launch_attn2(const AttnArgs& a, hipStream_t stream) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT;
  constexpr size_t lds = (size_t)2 * (KB * (DP + 8) + 2 * DV * VST) * sizeof(half_t);
  static bool attr_set = false;  // benign race: idempotent attribute
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_attn_kernel<DS, ONES, GENERAL>), lds, attr_set, "af_attention")) return af_check_launch("af_attention");
  const int qblocks = (a.Nq + 127) / 128, bh8 = (a.B * a.heads + 7) / 8 * 8;
  dim3 grid(qblocks * bh8), block(256);
  hipLaunchKernelGGL((af_attn_kernel<DS, ONES, GENERAL>), grid, block, lds, stream, a);
  return af_check_launch("af_attention");
}

template <int DS>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
  constexpr int DV = 32 * ((DS + 1) / 2);
  static const bool no_short = getenv("AF_NO_XATTN") != nullptr;     // A/B switch for profiling
  if (a.kbias == nullptr && a.causal_m == 0 && a.L <= 128 && !no_short)
    return DV > a.d ? launch_xattn<DS, true>(a, stream) : launch_xattn<DS, false>(a, stream);
  const bool general = a.kbias != nullptr || a.causal_m > 0 || a.L % KB != 0;
  static const int two_chain = getenv("AF_ATTN_TWO_CHAIN") ? atoi(getenv("AF_ATTN_TWO_CHAIN")) : 1;
  if constexpr (DS <= 3) {      // d <= 48: two chains fit the register file at 2 waves per SIMD
    // AF_ATTN_SLOT (default 1): the reference in a spare k slot where the head dim leaves one (d = 16 DS - 8: the 64 x 64 level's d = 40)
    static const int slot_env = getenv("AF_ATTN_SLOT") ? atoi(getenv("AF_ATTN_SLOT")) : 1;
    const char* pipe_s = getenv("AF_ATTN_PIPE");                         // read per call: the tests compare both forms in one process
    const int pipe_env = pipe_s ? atoi(pipe_s) : 0;                      // off: 1-4 % in isolation, < 1 % of a step (profiles/r05q_attn_pipe.txt)
    if (!general && two_chain && slot_env && pipe_env && a.Nq >= 512 && a.d == 16 * DS - 8 && DV > a.d) return launch_attn2pipe<DS>(a, stream);
    if (!general && two_chain && slot_env && a.Nq >= 512 && a.d == 16 * DS - 8 && DV > a.d) return launch_attn2chain<DS, true, true>(a, stream);
    if (!general && two_chain && a.Nq >= 512) return DV > a.d ? launch_attn2chain<DS, true>(a, stream) : launch_attn2chain<DS, false>(a, stream);
  }
  if (DV > a.d) return general ? launch_attn2<DS, true, true>(a, stream) : launch_attn2<DS, true, false>(a, stream);
  return general ? launch_attn2<DS, false, true>(a, stream) : launch_attn2<DS, false, false>(a, stream);
}

// explicit scores / probabilities for the capture path: one wave per (b, h, query) row, L <= 128
__global__ __launch_bounds__(256) void af_scores_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                        float* __restrict__ score, float* __restrict__ prob, int B, int Nq,
                                                        int L, int heads, int d, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nrows = (long)B * heads * Nq;
  if (row >= nrows) return;
  const int i = (int)(row % Nq);
  const int h = (int)((row / Nq) % heads);
  const int b = (int)(row / ((long)Nq * heads));
  const int C = heads * d;
  const half_t* qp = q + ((size_t)b * Nq + i) * C + h * d;
  float s[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = lane + 64 * jj;
    float acc = 0.f;
    if (j < L) {
      const half_t* kp = k + ((size_t)b * L + j) * C + h * d;
      for (int c0 = 0; c0 < d; c0 += 8) {
        const half8_t qv = *reinterpret_cast<const half8_t*>(qp + c0);
        const half8_t kv = *reinterpret_cast<const half8_t*>(kp + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)qv[e] * (float)kv[e];
      }
    }
    s[jj] = j < L ? acc * scale : -INFINITY;
  }
  const float mx = af_wave_max(fmaxf(s[0], s[1]));
  const float e0 = __expf(s[0] - mx), e1 = __expf(s[1] - mx);
  const float inv = 1.0f / af_wave_sum(e0 + e1);
  float* sp = score + (size_t)row * L;
  float* pp = prob + (size_t)row * L;
  if (lane < L) {
    sp[lane] = s[0];
    pp[lane] = e0 * inv;
  }
  if (lane + 64 < L) {
    sp[lane + 64] = s[1];
    pp[lane + 64] = e1 * inv;
  }
}

}  // namespace

extern "C" int af_attention_lse(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse,
                                const void* keybias, int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldo,
                                int ldv, int ldb, float scale, void* stream) {
  return af_attention_ex(q, k, vt, o, lse2, ld_lse, keybias, 0, B, Nq, L, heads, d, ldq, ldk, ldo, ldv, ldb, scale, stream);
}

extern "C" int af_attention_strided(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse, const void* keybias,
                                    int causal_m, int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb,
                                    int64_t vt_batch_stride, float scale, void* stream);

extern "C" int af_attention_ex(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse,
                               const void* keybias, int causal_m, int B, int Nq, int L, int heads, int d, int ldq, int ldk,
                               int ldo, int ldv, int ldb, float scale, void* stream) {
  return af_attention_strided(q, k, vt, o, lse2, ld_lse, keybias, causal_m, B, Nq, L, heads, d, ldq, ldk, ldo, ldv, ldb,
                              (int64_t)heads * d * ldv, scale, stream);
}

extern "C" int af_attention_strided(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse, const void* keybias,
                                    int causal_m, int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb,
                                    int64_t vt_batch_stride, float scale, void* stream) {
  AF_REQUIRE(q && k && vt && o, "af_attention: null pointer");
  AF_REQUIRE(vt_batch_stride >= (int64_t)heads * d * ldv && vt_batch_stride % 8 == 0, "af_attention: vt_batch_stride must cover heads*d rows of ldv and keep 16-byte alignment");
  AF_REQUIRE(B > 0 && Nq > 0 && L > 0 && heads > 0 && d > 0, "af_attention: bad sizes");
  AF_REQUIRE(d % 8 == 0, "af_attention: head dim must be a multiple of 8");
  AF_SUPPORTED(d <= 160, "af_attention: head dim > 160");
  const int C = heads * d;
  AF_REQUIRE(ldq >= C && ldk >= C && ldo >= C && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0,
             "af_attention: bad row strides");
  AF_REQUIRE(ldv % 8 == 0 && ldv >= ((L + 7) / 8) * 8, "af_attention: ldv must be a multiple of 8 and >= roundup(L, 8)");
  if (keybias) AF_REQUIRE(ldb >= ((L + KB - 1) / KB) * KB && ldb % 4 == 0, "af_attention: ldb must cover L rounded up to 64");
  AttnArgs a;
  a.q = (const half_t*)q;
  a.k = (const half_t*)k;
  a.vt = (const half_t*)vt;
  a.o = (half_t*)o;
  a.kbias = (const float*)keybias;
  a.lse2 = (float*)lse2;
  a.ld_lse = ld_lse;
  a.causal_m = causal_m;
  AF_REQUIRE(causal_m >= 0, "af_attention: causal_m < 0");
  if (lse2) AF_REQUIRE(ld_lse >= Nq, "af_attention: ld_lse < Nq");
  a.B = B;
  a.Nq = Nq;
  a.L = L;
  a.heads = heads;
  a.d = d;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldo = ldo;
  a.ldv = ldv;
  a.ldb = ldb;
  a.vbs = vt_batch_stride;
  a.c = scale * 1.4426950408889634f;
  AfLaunchScope scope(L < Nq ? AF_FAM_XATTN : AF_FAM_ATTN, stream);
  hipStream_t s = (hipStream_t)stream;
  const int ds = (d + 15) / 16;
  switch (ds) {
    case 1: return launch_attn<1>(a, s);
    case 2: return launch_attn<2>(a, s);
    case 3: return launch_attn<3>(a, s);
    case 4: return launch_attn<4>(a, s);
    case 5: return launch_attn<5>(a, s);
    case 6: return launch_attn<6>(a, s);
    case 8: return launch_attn<8>(a, s);
    case 10: return launch_attn<10>(a, s);
    default: return af_fail(AF_E_UNSUPPORTED, "af_attention: unsupported head dim (need ceil(d/16) in {1,2,3,4,5,6,8,10})");
  }
}

extern "C" int af_attention(const void* q, const void* k, const void* vt, void* o, const void* keybias, int B, int Nq,
                            int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb, float scale,
                            void* stream) {
  return af_attention_lse(q, k, vt, o, nullptr, 0, keybias, B, Nq, L, heads, d, ldq, ldk, ldo, ldv, ldb, scale, stream);
}

extern "C" int af_attention_scores(const void* q, const void* k, void* score, void* prob, int B, int Nq, int L, int heads,
                                   int d, float scale, void* stream) {
  AF_REQUIRE(q && k && score && prob, "af_attention_scores: null pointer");
  AF_REQUIRE(B > 0 && Nq > 0 && L > 0 && heads > 0 && d > 0 && d % 8 == 0, "af_attention_scores: bad sizes");
  AF_SUPPORTED(L <= 128, "af_attention_scores: L > 128 (capture path is for cross-attention only)");
  const long rows = (long)B * heads * Nq;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  AfLaunchScope scope(AF_FAM_ATTN, stream);
  hipLaunchKernelGGL(af_scores_kernel, grid, block, 0, (hipStream_t)stream, (const half_t*)q, (const half_t*)k,
                     (float*)score, (float*)prob, B, Nq, L, heads, d, scale);
  return af_check_launch("af_attention_scores");
}
