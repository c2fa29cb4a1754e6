// af_attn_bwd.hip -- flash-style attention backward (dQ, dK, dV) on MFMA, recomputing P from
// Q, K and the forward's base-2 log-sum-exp instead of storing the [b*h, N, L] probabilities.
//
//   z = scale * q.k (+ keybias) ; P = softmax_j(z) ; O = P V
//   delta_i = sum_d dO_id O_id ;  dP = dO V^T ;  dz = P o (dP - delta) ;
//   dQ = scale * dz K ;  dK = scale * dz^T Q ;  dV = P^T dO
//
// Two kernels with the same "reduction index in the accumulator registers" trick as the forward
// (v_mfma_f32_32x32x16_f16, accumulator tile reused directly as the next product's B operand):
//   * dQ kernel: QUERY on the lane.  S^T = K Q^T, dP^T = V dO^T (A = K / V rows from LDS,
//     B = Q / dO fragments in registers), dz^T in registers -> dQ^T += K^T dz^T (A = K^T rows).
//   * dK/dV kernel: KEY on the lane.  S = Q K^T, dP = dO V^T (A = Q / dO rows from LDS, B = K / V
//     fragments in registers) -> dV^T += dO^T P, dK^T += Q^T dz (A = dO^T / Q^T rows).
// The transposed operands (Q^T, K^T, dO^T: [B, C, tokens], token index contiguous) are produced
// once per call by a tiled transpose into the caller's scratch buffer.  Stages are 32 tokens; the stage tiles exist twice in LDS:
// stage st + 1 is stored into the other copy (from registers loaded one stage earlier) under stage st's MFMAs, stage st + 2's
// global loads are issued right after, and ONE workgroup barrier separates stages.
#include <float.h>

#include <type_traits>

#include "af_common.h"

namespace {

constexpr int TST = 36;  // transposed-tile LDS row stride in halves (72 B, conflict-free ds_read_b64)

struct BwdArgs {
  const half_t* q;    // [B, Nq, ldq]
  const half_t* k;    // [B, L, ldk]
  const half_t* v;    // [B, L, ldv]   (row-major V)
  const half_t* dout; // [B, Nq, ldo]
  const half_t* qt;   // [B, C, ldqt]  (Q^T)
  const half_t* kt;   // [B, C, ldkt]  (K^T)
  const half_t* dot;  // [B, C, ldqt]  (dO^T)
  const float* lse2;  // [B, heads, nqpad]
  const float* delta; // [B, heads, nqpad]
  const float* kbias; // [B, ldb] or null
  half_t* dq;         // [B, Nq, lddq]
  half_t* dk;         // [B, L, lddk]
  half_t* dv;         // [B, L, lddv]
  int B, Nq, L, heads, d;
  int ldq, ldk, ldv, ldo, ldqt, ldkt, lddq, lddk, lddv, ldb, nqpad, ld_lse;
  int causal_m;  // > 0: key j visible to query i iff j / causal_m <= i
  float c;      // scale * log2(e)
  float scale;
};

// delta[b,h,q] = sum_d dO * O
__global__ __launch_bounds__(256) void attn_delta_kernel(const half_t* __restrict__ o, const half_t* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int Nq, int heads, int d, int ldo_o,
                                                         int ldo_d, int nqpad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * heads * Nq) return;
  const int qi = (int)(idx % Nq);
  const int h = (int)((idx / Nq) % heads);
  const int b = (int)(idx / ((long)Nq * heads));
  const half_t* op = o + ((size_t)b * Nq + qi) * ldo_o + h * d;
  const half_t* dp = dout + ((size_t)b * Nq + qi) * ldo_d + h * d;
  float s = 0.f;
  for (int c0 = 0; c0 < d; c0 += 8) {
    const half8_t a = *reinterpret_cast<const half8_t*>(op + c0), g = *reinterpret_cast<const half8_t*>(dp + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)g[e];
  }
  delta[((size_t)b * heads + h) * nqpad + qi] = s;
}

// rows [32][DP] row-major tile loader (zero-filled beyond d / beyond ntok)
template <int DP, int NC>
__device__ __forceinline__ void load_rows(half8_t (&r)[NC], const half_t* base, int ld, int tok0, int ntok, int hd, int d, int tid) {
  constexpr int CH = 32 * (DP / 8);
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int i = tid + 256 * j;
    const int row = i / (DP / 8), ch = i - row * (DP / 8);
    const bool ok = i < CH && tok0 + row < ntok && ch * 8 < d;
    r[j] = ok ? *reinterpret_cast<const half8_t*>(base + (size_t)(tok0 + row) * ld + hd + ch * 8) : zero8;
  }
}
template <int DP, int NC>
__device__ __forceinline__ void store_rows(const half8_t (&r)[NC], half_t* lds, int tid) {
  constexpr int CH = 32 * (DP / 8);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int i = tid + 256 * j;
    if (i < CH) {
      const int row = i / (DP / 8), ch = i - row * (DP / 8);
      *reinterpret_cast<half8_t*>(lds + row * (DP + 8) + ch * 8) = r[j];
    }
  }
}
// transposed [DV][32 tokens] tile loader from a [C][ld] token-contiguous source
template <int DV, int NC>
__device__ __forceinline__ void load_tr(half8_t (&r)[NC], const half_t* base, int ld, int tok0, int ntok, int d, int tid) {
  constexpr int CH = DV * 4;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int i = tid + 256 * j;
    const int row = i >> 2, ch = i & 3;
    const int tk = tok0 + ch * 8;
    const bool ok = i < CH && row < d && tk < ntok;
    // no VALU touch of the loaded chunk here (it would force a vmcnt(0) wait at the load); the transposed sources are
    // produced by af_launch_transpose_tokens, which zero-fills tokens >= ntok up to the row stride, so no tail mask is needed
    r[j] = ok ? *reinterpret_cast<const half8_t*>(base + (size_t)row * ld + tk) : zero8;
  }
}
template <int DV, int NC>
__device__ __forceinline__ void store_tr(const half8_t (&r)[NC], half_t* lds, int tid) {
  constexpr int CH = DV * 4;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int i = tid + 256 * j;
    if (i < CH) {
      const int row = i >> 2, ch = i & 3;
      half_t* dst = lds + row * TST + ch * 8;
      const half4_t lo = {r[j][0], r[j][1], r[j][2], r[j][3]};
      const half4_t hi = {r[j][4], r[j][5], r[j][6], r[j][7]};
      *reinterpret_cast<half4_t*>(dst) = lo;
      *reinterpret_cast<half4_t*>(dst + 4) = hi;
    }
  }
}
// A fragment of a transposed tile: lane (r, hh), k-step s2: tokens 16 s2 + 4 hh + {0..3} and + 8
__device__ __forceinline__ half8_t tr_frag(const half_t* lds, int row, int s2, int hh) {
  const half_t* p = lds + row * TST + 16 * s2 + 4 * hh;
  const half4_t lo = *reinterpret_cast<const half4_t*>(p);
  const half4_t hi = *reinterpret_cast<const half4_t*>(p + 8);
  const half8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

// ------------------------------------------------------------------------------------ dQ
template <int DS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(BwdArgs a) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT, RST = DP + 8;
  constexpr int NRC = (32 * (DP / 8) + 255) / 256, NTC = (DV * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  // two copies of the stage tiles: stage st + 1 is written into the other copy under stage st's MFMAs, ONE barrier per stage
  constexpr int BUF = 2 * 32 * RST + DV * TST;      // halves per copy: K rows [32][RST], V rows [32][RST], K^T [DV][TST]
  half_t* const smem = reinterpret_cast<half_t*>(af_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int query = blockIdx.x * 128 + wave * 32 + r;
  const int C = a.heads * a.d, hd = h * a.d;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  half8_t qf[DS], gf[DS];  // Q and dO fragments (B operands)
  {
    const size_t row = (size_t)b * a.Nq + (query < a.Nq ? query : 0);
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = 16 * s + 8 * hh;
      const bool ok = query < a.Nq && dc < a.d;
      qf[s] = ok ? *reinterpret_cast<const half8_t*>(a.q + row * a.ldq + hd + dc) : zero8;
      gf[s] = ok ? *reinterpret_cast<const half8_t*>(a.dout + row * a.ldo + hd + dc) : zero8;
    }
  }
  const int qsafe = query < a.Nq ? query : 0;
  const float lse = a.lse2[((size_t)b * a.heads + h) * a.ld_lse + qsafe];
  const float delta = a.delta[((size_t)b * a.heads + h) * a.nqpad + qsafe];

  const half_t* kb = a.k + (size_t)b * a.L * a.ldk;
  const half_t* vb = a.v + (size_t)b * a.L * a.ldv;
  const half_t* ktb = a.kt + ((size_t)b * C + hd) * a.ldkt;
  half8_t rk[NRC], rv[NRC], rt[NTC];
  floatx16 dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[t][i] = 0.f;

  const int nstage = (a.L + 31) / 32;
  load_rows<DP, NRC>(rk, kb, a.ldk, 0, a.L, hd, a.d, tid);
  load_rows<DP, NRC>(rv, vb, a.ldv, 0, a.L, hd, a.d, tid);
  load_tr<DV, NTC>(rt, ktb, a.ldkt, 0, a.L, a.d, tid);
  store_rows<DP, NRC>(rk, smem, tid);
  store_rows<DP, NRC>(rv, smem + 32 * RST, tid);
  store_tr<DV, NTC>(rt, smem + 2 * 32 * RST, tid);
  if (nstage > 1) {
    load_rows<DP, NRC>(rk, kb, a.ldk, 32, a.L, hd, a.d, tid);
    load_rows<DP, NRC>(rv, vb, a.ldv, 32, a.L, hd, a.d, tid);
    load_tr<DV, NTC>(rt, ktb, a.ldkt, 32, a.L, a.d, tid);
  }
  for (int st = 0; st < nstage; ++st) {
    const int key0 = st * 32;
    __syncthreads();  // copy st & 1 is complete; everyone is done reading the other copy (stage st - 1)
    const half_t* Ks = smem + (st & 1) * BUF;
    const half_t* Vs = Ks + 32 * RST;
    const half_t* KTs = Vs + 32 * RST;
    if (st + 1 < nstage) {
      half_t* nx = smem + ((st + 1) & 1) * BUF;
      store_rows<DP, NRC>(rk, nx, tid);
      store_rows<DP, NRC>(rv, nx + 32 * RST, tid);
      store_tr<DV, NTC>(rt, nx + 2 * 32 * RST, tid);
      if (st + 2 < nstage) {
        load_rows<DP, NRC>(rk, kb, a.ldk, key0 + 64, a.L, hd, a.d, tid);
        load_rows<DP, NRC>(rv, vb, a.ldv, key0 + 64, a.L, hd, a.d, tid);
        load_tr<DV, NTC>(rt, ktb, a.ldkt, key0 + 64, a.L, a.d, tid);
      }
    }
    floatx16 sT, pT;
#pragma unroll
    for (int i = 0; i < 16; ++i) sT[i] = pT[i] = 0.f;
    // every fragment of the group is requested before its first MFMA (round 5: hipcc otherwise issues each ds_read right in front of the MFMA that
    // needs it and waits lgkmcnt(0) there -- one LDS round trip per MFMA in the ISA, tools/isa_seq.py)
    // (d = 40 only: at d = 80 / 160 the extra live fragments cost the second wave per SIMD)
    constexpr bool AHEAD = DS <= 3;
    half8_t kfa[AHEAD ? DS : 1], vfa[AHEAD ? DS : 1], ktf[AHEAD ? DT : 1][2];
    if constexpr (AHEAD) {
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        kfa[s] = *reinterpret_cast<const half8_t*>(Ks + r * RST + 16 * s + 8 * hh);
        vfa[s] = *reinterpret_cast<const half8_t*>(Vs + r * RST + 16 * s + 8 * hh);
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) ktf[t][s2] = tr_frag(KTs, 32 * t + r, s2, hh);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const half8_t kf = AHEAD ? kfa[AHEAD ? s : 0] : *reinterpret_cast<const half8_t*>(Ks + r * RST + 16 * s + 8 * hh);
      sT = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sT, 0, 0, 0);
      const half8_t vf = AHEAD ? vfa[AHEAD ? s : 0] : *reinterpret_cast<const half8_t*>(Vs + r * RST + 16 * s + 8 * hh);
      pT = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, gf[s], pT, 0, 0, 0);
    }
    half8_t zf[2];
    // a stage whose 32 keys all exist and are all visible needs no per-element test (32 v_cndmask + compares per stage: the loop is
    // VALU-bound); keys past L only occur in the last stage, the causal test only in the CLIP encoders
    auto softmax_block = [&](auto masked_c) {
      constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kk = key0 + 8 * g + 4 * hh;
        floatx4 bias = {0.f, 0.f, 0.f, 0.f};
        if (a.kbias) bias = *reinterpret_cast<const floatx4*>(a.kbias + (size_t)b * a.ldb + kk);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const float t = sT[i] * a.c + bias[e];
          float p = __builtin_amdgcn_exp2f(t - lse);
          if (MASKED) {
            bool vis = kk + e < a.L;
            if (a.causal_m > 0) vis = vis && ((kk + e) / a.causal_m <= query);
            p = vis ? p : 0.f;
          }
          zf[i >> 3][i & 7] = (half_t)(p * (pT[i] - delta));
        }
      }
    };
    if (a.causal_m > 0 || key0 + 32 > a.L) softmax_block(std::true_type{});
    else softmax_block(std::false_type{});
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AHEAD ? ktf[AHEAD ? t : 0][s2] : tr_frag(KTs, 32 * t + r, s2, hh), zf[s2], dq[t], 0, 0, 0);
  }
  if (query < a.Nq) {
    half_t* op = a.dq + ((size_t)b * a.Nq + query) * a.lddq + hd;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * t + 8 * g + 4 * hh;
        if (dd < a.d) {
          const half4_t o = {(half_t)(dq[t][4 * g] * a.scale), (half_t)(dq[t][4 * g + 1] * a.scale),
                             (half_t)(dq[t][4 * g + 2] * a.scale), (half_t)(dq[t][4 * g + 3] * a.scale)};
          *reinterpret_cast<half4_t*>(op + dd) = o;
        }
      }
  }
}

// ------------------------------------------------------------------------------------ dK, dV
template <int DS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(BwdArgs a) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT, RST = DP + 8;
  constexpr int NRC = (32 * (DP / 8) + 255) / 256, NTC = (DV * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  constexpr int BUF = 2 * 32 * RST + 2 * DV * TST;  // halves per copy: Q rows [32][RST], dO rows [32][RST], Q^T [DV][TST], dO^T [DV][TST]
  half_t* const smem = reinterpret_cast<half_t*>(af_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int key = blockIdx.x * 128 + wave * 32 + r;
  const int C = a.heads * a.d, hd = h * a.d;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  half8_t kf[DS], vf[DS];  // K and V fragments of this lane's key (B operands)
  {
    const size_t row = (size_t)b * a.L + (key < a.L ? key : 0);
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int dc = 16 * s + 8 * hh;
      const bool ok = key < a.L && dc < a.d;
      kf[s] = ok ? *reinterpret_cast<const half8_t*>(a.k + row * a.ldk + hd + dc) : zero8;
      vf[s] = ok ? *reinterpret_cast<const half8_t*>(a.v + row * a.ldv + hd + dc) : zero8;
    }
  }
  float kbias = 0.f;
  if (a.kbias && key < a.L) kbias = a.kbias[(size_t)b * a.ldb + key];
  const bool key_ok = key < a.L;

  const half_t* qb = a.q + (size_t)b * a.Nq * a.ldq;
  const half_t* gb = a.dout + (size_t)b * a.Nq * a.ldo;
  const half_t* qtb = a.qt + ((size_t)b * C + hd) * a.ldqt;
  const half_t* gtb = a.dot + ((size_t)b * C + hd) * a.ldqt;
  const float* lseb = a.lse2 + ((size_t)b * a.heads + h) * a.ld_lse;
  const float* delb = a.delta + ((size_t)b * a.heads + h) * a.nqpad;
  half8_t rq[NRC], rg[NRC], rqt[NTC], rgt[NTC];
  floatx16 dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[t][i] = dv[t][i] = 0.f;

  const int nstage = (a.Nq + 31) / 32;
  load_rows<DP, NRC>(rq, qb, a.ldq, 0, a.Nq, hd, a.d, tid);
  load_rows<DP, NRC>(rg, gb, a.ldo, 0, a.Nq, hd, a.d, tid);
  load_tr<DV, NTC>(rqt, qtb, a.ldqt, 0, a.Nq, a.d, tid);
  load_tr<DV, NTC>(rgt, gtb, a.ldqt, 0, a.Nq, a.d, tid);
  store_rows<DP, NRC>(rq, smem, tid);
  store_rows<DP, NRC>(rg, smem + 32 * RST, tid);
  store_tr<DV, NTC>(rqt, smem + 2 * 32 * RST, tid);
  store_tr<DV, NTC>(rgt, smem + 2 * 32 * RST + DV * TST, tid);
  if (nstage > 1) {
    load_rows<DP, NRC>(rq, qb, a.ldq, 32, a.Nq, hd, a.d, tid);
    load_rows<DP, NRC>(rg, gb, a.ldo, 32, a.Nq, hd, a.d, tid);
    load_tr<DV, NTC>(rqt, qtb, a.ldqt, 32, a.Nq, a.d, tid);
    load_tr<DV, NTC>(rgt, gtb, a.ldqt, 32, a.Nq, a.d, tid);
  }
  for (int st = 0; st < nstage; ++st) {
    const int q0 = st * 32;
    __syncthreads();  // copy st & 1 is complete; everyone is done reading the other copy (stage st - 1)
    const half_t* Qs = smem + (st & 1) * BUF;
    const half_t* Gs = Qs + 32 * RST;
    const half_t* QTs = Gs + 32 * RST;
    const half_t* GTs = QTs + DV * TST;
    if (st + 1 < nstage) {
      half_t* nx = smem + ((st + 1) & 1) * BUF;
      store_rows<DP, NRC>(rq, nx, tid);
      store_rows<DP, NRC>(rg, nx + 32 * RST, tid);
      store_tr<DV, NTC>(rqt, nx + 2 * 32 * RST, tid);
      store_tr<DV, NTC>(rgt, nx + 2 * 32 * RST + DV * TST, tid);
      if (st + 2 < nstage) {
        load_rows<DP, NRC>(rq, qb, a.ldq, q0 + 64, a.Nq, hd, a.d, tid);
        load_rows<DP, NRC>(rg, gb, a.ldo, q0 + 64, a.Nq, hd, a.d, tid);
        load_tr<DV, NTC>(rqt, qtb, a.ldqt, q0 + 64, a.Nq, a.d, tid);
        load_tr<DV, NTC>(rgt, gtb, a.ldqt, q0 + 64, a.Nq, a.d, tid);
      }
    }
    floatx16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
    // fragments requested ahead of the MFMAs, as in the dQ kernel (d = 40 / 80; at d = 160 they do not fit the register file)
    constexpr bool AHEAD = DS <= 5;
    half8_t qaa[AHEAD ? DS : 1], gaa[AHEAD ? DS : 1], gtf[AHEAD ? DT : 1][2], qtf[AHEAD ? DT : 1][2];
    if constexpr (AHEAD) {
#pragma unroll
      for (int ks = 0; ks < DS; ++ks) {
        qaa[ks] = *reinterpret_cast<const half8_t*>(Qs + r * RST + 16 * ks + 8 * hh);
        gaa[ks] = *reinterpret_cast<const half8_t*>(Gs + r * RST + 16 * ks + 8 * hh);
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          gtf[t][s2] = tr_frag(GTs, 32 * t + r, s2, hh);
          qtf[t][s2] = tr_frag(QTs, 32 * t + r, s2, hh);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < DS; ++ks) {
      const half8_t qa = AHEAD ? qaa[AHEAD ? ks : 0] : *reinterpret_cast<const half8_t*>(Qs + r * RST + 16 * ks + 8 * hh);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, kf[ks], s, 0, 0, 0);
      const half8_t ga = AHEAD ? gaa[AHEAD ? ks : 0] : *reinterpret_cast<const half8_t*>(Gs + r * RST + 16 * ks + 8 * hh);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga, vf[ks], dp, 0, 0, 0);
    }
    half8_t pf[2], zf[2];
    // a stage whose 32 queries all exist needs no per-element test when nothing is causal (the loop is VALU-bound: 32 v_cndmask +
    // compares per stage): lanes of keys past L then carry finite junk in accumulators that are never stored
    auto softmax_block = [&](auto masked_c) {
      constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qq = q0 + 8 * g + 4 * hh;  // queries qq .. qq+3 live in regs 4g .. 4g+3 (nqpad covers the overrun)
        const floatx4 lse = *reinterpret_cast<const floatx4*>(lseb + qq);
        const floatx4 del = *reinterpret_cast<const floatx4*>(delb + qq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const float t = s[i] * a.c + kbias;
          if (MASKED) {
            bool ok = key_ok && qq + e < a.Nq;   // padding rows of lse/delta may hold anything: select, never multiply
            if (a.causal_m > 0) ok = ok && (key / a.causal_m <= qq + e);
            const float p = ok ? __builtin_amdgcn_exp2f(t - lse[e]) : 0.f;
            pf[i >> 3][i & 7] = (half_t)p;
            zf[i >> 3][i & 7] = (half_t)(ok ? p * (dp[i] - del[e]) : 0.f);
          } else {
            const float p = __builtin_amdgcn_exp2f(t - lse[e]);
            pf[i >> 3][i & 7] = (half_t)p;
            zf[i >> 3][i & 7] = (half_t)(p * (dp[i] - del[e]));
          }
        }
      }
    };
    if (a.causal_m > 0 || q0 + 32 > a.Nq) softmax_block(std::true_type{});
    else softmax_block(std::false_type{});
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AHEAD ? gtf[AHEAD ? t : 0][s2] : tr_frag(GTs, 32 * t + r, s2, hh), pf[s2], dv[t], 0, 0, 0);
        dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AHEAD ? qtf[AHEAD ? t : 0][s2] : tr_frag(QTs, 32 * t + r, s2, hh), zf[s2], dk[t], 0, 0, 0);
      }
  }
  if (key < a.L) {
    half_t* okp = a.dk + ((size_t)b * a.L + key) * a.lddk + hd;
    half_t* ovp = a.dv + ((size_t)b * a.L + key) * a.lddv + hd;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * t + 8 * g + 4 * hh;
        if (dd < a.d) {
          const half4_t ok = {(half_t)(dk[t][4 * g] * a.scale), (half_t)(dk[t][4 * g + 1] * a.scale),
                              (half_t)(dk[t][4 * g + 2] * a.scale), (half_t)(dk[t][4 * g + 3] * a.scale)};
          const half4_t ov = {(half_t)dv[t][4 * g], (half_t)dv[t][4 * g + 1], (half_t)dv[t][4 * g + 2], (half_t)dv[t][4 * g + 3]};
          *reinterpret_cast<half4_t*>(okp + dd) = ok;
          *reinterpret_cast<half4_t*>(ovp + dd) = ov;
        }
      }
  }
}

template <int DS>
int launch_bwd(const BwdArgs& a, hipStream_t s) {
  constexpr int DP = 16 * DS, DT = (DS + 1) / 2, DV = 32 * DT, RST = DP + 8;
  constexpr size_t lds_dq = 2 * (size_t)(2 * 32 * RST + DV * TST) * sizeof(half_t);          // two copies of the stage tiles
  constexpr size_t lds_dkv = 2 * (size_t)(2 * 32 * RST + 2 * DV * TST) * sizeof(half_t);
  static_assert(lds_dkv <= 160 * 1024, "LDS budget");
  static_assert(lds_dq <= 160 * 1024, "LDS budget");
  static bool attr_dkv = false, attr_dq = false;
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<DS>), lds_dkv, attr_dkv, "af_attention_bwd(dkv)") ||
      !af_allow_dyn_lds(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DS>), lds_dq, attr_dq, "af_attention_bwd(dq)"))
    return af_check_launch("af_attention_bwd");
  hipLaunchKernelGGL(attn_bwd_dq_kernel<DS>, dim3((a.Nq + 127) / 128, a.heads, a.B), dim3(256), lds_dq, s, a);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<DS>, dim3((a.L + 127) / 128, a.heads, a.B), dim3(256), lds_dkv, s, a);
  return af_check_launch("af_attention_bwd");
}

}  // namespace

extern "C" int64_t af_attention_bwd_scratch_bytes(int B, int Nq, int L, int heads, int d) {
  if (B <= 0 || Nq <= 0 || L <= 0 || heads <= 0 || d <= 0) return 0;
  const int64_t C = (int64_t)heads * d, nq8 = (Nq + 7) / 8 * 8, l8 = (L + 7) / 8 * 8, nq64 = (Nq + 63) / 64 * 64 + 64;
  return (2 * B * C * nq8 + B * C * l8) * 2 + B * heads * nq64 * 4 + 256;
}

extern "C" int af_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const void* lse2,
                                int ld_lse, const void* keybias, int causal_m, void* dq, void* dk, void* dv, void* scratch, int64_t scratch_bytes, int B,
                                int Nq, int L, int heads, int d, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk,
                                int lddv, int ldb, float scale, void* stream) {
  AF_REQUIRE(q && k && v && o && dout && lse2 && dq && dk && dv && scratch, "af_attention_bwd: null pointer");
  AF_REQUIRE(B > 0 && Nq > 0 && L > 0 && heads > 0 && d > 0 && d % 8 == 0, "af_attention_bwd: bad sizes");
  AF_SUPPORTED(d <= 160, "af_attention_bwd: head dim > 160");
  const int C = heads * d;
  AF_REQUIRE(ldq >= C && ldk >= C && ldv >= C && ldo >= C && lddo >= C && lddq >= C && lddk >= C && lddv >= C,
             "af_attention_bwd: row strides must be >= heads*d");
  AF_REQUIRE((ldq | ldk | ldv | ldo | lddo) % 8 == 0 && (lddq | lddk | lddv) % 4 == 0, "af_attention_bwd: misaligned row strides");
  AF_REQUIRE(ld_lse >= (Nq + 31) / 32 * 32 && ld_lse % 4 == 0, "af_attention_bwd: ld_lse must cover Nq rounded up to 32 and be a multiple of 4");
  AF_REQUIRE(scratch_bytes >= af_attention_bwd_scratch_bytes(B, Nq, L, heads, d), "af_attention_bwd: scratch too small");
  if (keybias) AF_REQUIRE(ldb >= (L + 31) / 32 * 32 && ldb % 4 == 0, "af_attention_bwd: ldb must cover L rounded up to 32");
  const int nq8 = (Nq + 7) / 8 * 8, l8 = (L + 7) / 8 * 8, nq64 = (Nq + 63) / 64 * 64 + 64;
  half_t* qt = (half_t*)scratch;
  half_t* dot = qt + (size_t)B * C * nq8;
  half_t* kt = dot + (size_t)B * C * nq8;
  float* delta = (float*)(((uintptr_t)(kt + (size_t)B * C * l8) + 255) & ~(uintptr_t)255);
  hipStream_t s = (hipStream_t)stream;
  AfLaunchScope scope(AF_FAM_ATTN, stream);
  const AfTransposeJob jobs[3] = {{(const half_t*)q, qt, Nq, C, ldq, nq8}, {(const half_t*)dout, dot, Nq, C, lddo, nq8},
                                  {(const half_t*)k, kt, L, C, ldk, l8}};
  af_launch_transpose_tokens_multi(jobs, 3, B, s);                                   // Q^T, dO^T, K^T: [B, C, tokens], one launch (af_bwd.hip)
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)(((long)B * heads * Nq + 255) / 256)), dim3(256), 0, s, (const half_t*)o,
                     (const half_t*)dout, delta, B, Nq, heads, d, ldo, lddo, nq64);
  BwdArgs a;
  a.q = (const half_t*)q;
  a.k = (const half_t*)k;
  a.v = (const half_t*)v;
  a.dout = (const half_t*)dout;
  a.qt = qt;
  a.kt = kt;
  a.dot = dot;
  a.lse2 = (const float*)lse2;
  a.delta = delta;
  a.kbias = (const float*)keybias;
  a.dq = (half_t*)dq;
  a.dk = (half_t*)dk;
  a.dv = (half_t*)dv;
  a.B = B;
  a.Nq = Nq;
  a.L = L;
  a.heads = heads;
  a.d = d;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldv = ldv;
  a.ldo = lddo;
  a.ldqt = nq8;
  a.ldkt = l8;
  a.lddq = lddq;
  a.lddk = lddk;
  a.lddv = lddv;
  a.ldb = ldb;
  a.nqpad = nq64;
  a.ld_lse = ld_lse;
  a.causal_m = causal_m;
  a.c = scale * 1.4426950408889634f;
  a.scale = scale;
  const int ds = (d + 15) / 16;
  switch (ds) {
    case 1: return launch_bwd<1>(a, s);
    case 2: return launch_bwd<2>(a, s);
    case 3: return launch_bwd<3>(a, s);
    case 4: return launch_bwd<4>(a, s);
    case 5: return launch_bwd<5>(a, s);
    case 6: return launch_bwd<6>(a, s);
    case 8: return launch_bwd<8>(a, s);
    case 10: return launch_bwd<10>(a, s);
    default: return af_fail(AF_E_UNSUPPORTED, "af_attention_bwd: unsupported head dim");
  }
}
