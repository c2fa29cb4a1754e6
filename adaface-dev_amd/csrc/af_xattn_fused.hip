// af_xattn_fused.hip -- the whole cross-attention block of a C = 320 transformer layer as one launch (af_xattn_fused).
#include <stdlib.h>

#include "af_common.h"

namespace {

__device__ __forceinline__ void glds16(const half_t* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Whole cross-attention block of a transformer layer at C = 320 (8 heads of 40; the 64 x 64 level) in ONE launch:
//     out = x + b_o + W_o concat_h( softmax(q_h K_h^T * scale) V_h ),   q = LN(x) W_q^T          (attention.py:168-222, 242-252)
// K / V^T come from the forward's one batched context projection (77 keys).  Unfused this is three launches per layer (to_q with the
// LayerNorm folded in, the 77-key core, to_out with the residual: 21 + 23 + 21 us at U-Net batch 8), each too short to fill the chip.
// A workgroup owns 128 tokens, and inside it EVERY WAVE OWNS 16 TOKENS FOR THE WHOLE BLOCK -- no data is exchanged between waves:
//   * the wave's 16 x 320 slice of x sits in registers as the ten B-operand fragments of the q projection (its LayerNorm statistics are
//     taken from them);
//   * per head: Q^T [48 x 16] = W_q,h x^T (30 MFMAs; W_q,h rows in LDS, LayerNorm folded as in af_gemm_desc.ln_colsum, softmax scale folded
//     in), S^T [80 keys x 16] = K_h Q^T, softmax over the keys of a token (the token is the lane's column: two shuffles), O^T [48 x 16] =
//     V_h^T P^T, OUT^T [320 x 16] += W_o,h O^T (40 MFMAs, accumulated in registers over the heads) -- each product's accumulator layout
//     (lane = token, four consecutive rows per register quad) IS the next product's B operand once the K index of that product is
//     permuted accordingly (k slot 8 fq + j <-> row 4 fq + j of tile a for j < 4, of tile b for j >= 4), so the A operands (K_h, V_h^T,
//     W_o,h rows in LDS) are read as two 8-byte halves per fragment and nothing goes through LDS between the four products;
//   * LDS holds only weights / keys / values: W_q,h (30 KB), and double-buffered W_o,h (40 KB), K_h (10 KB), V_h^T (12 KB) whose next
//     head's copies stream in under the current head; two workgroup barriers per head.
// MEASURED (profiles/r03at_xattn_fused.txt): parity-green; 63.2 us per layer at U-Net batch 8 against 63.6 us for the three launches, -0.03 ms per
// denoise step (three alternating runs each) -- on par alone, marginally ahead in the step (two launches fewer per layer).  First form 68 us:
// the folded LayerNorm's column sums were loaded from global memory inside the head loop (an L2 round trip on every head's critical path) --
// staged in LDS once: 64.0; W_o fragments requested a pair of output tiles ahead, dependent MFMAs spaced: 63.2.  What bounds it: a wave that owns
// 16 tokens has ONE B tile, so every 1-KB weight / key / value fragment read from LDS feeds exactly one MFMA (the tiled GEMMs reuse a fragment
// over 4 - 5 tiles): 5.8 MB of LDS reads per workgroup, and the four products of a head are a dependent chain inside the wave (Q -> S -> P -> O ->
// OUT) that only the SIMD's second wave can overlap.  With the per-head DMA removed it still takes 59.6 us.
constexpr int XA_C = 320, XA_H = 8, XA_D = 40, XA_BM = 128;
constexpr int XA_WQ = 5 * 48 * 128, XA_WO = XA_C * 128, XA_KB = 80 * 128, XA_VB = 2 * 48 * 128;
constexpr int XA_LDS = XA_WQ + 2 * (XA_WO + XA_KB + XA_VB) + 2 * XA_C * 4;      // 157,696 B + the folded LayerNorm's column sums and shift (2,560 B)

struct XaDev {
  const half_t* x;         // [M][320] un-normalised rows
  const half_t* wq;        // packed [>= 320][kpad_q], LayerNorm-folded (gamma * W_q); rows head-major
  const float* bq;         // [320] W_q beta (the folded LayerNorm's shift) or nullptr
  const float* cs;         // [320] column sums of the packed rows
  const half_t* k;         // [B * L][ldk]: this layer's keys, head h at columns 40 h
  const half_t* vt;        // [B][320][ldv] (batch stride vbs): values transposed; keys L .. ldv-1 may hold ANYTHING (masked after the LDS read)
  const half_t* wo;        // packed [>= 320][kpad_o]
  const float* bo;         // [320] or nullptr
  const half_t* residual;  // [M][320] or nullptr
  half_t* out;             // [M][320]
  const half_t* zeros;
  int M, N, L, kpad_q, kpad_o, ldk, ldv, ln_on;
  long vbs;
  float ln_eps, scale_log2e;
  // chained form (af_xattn_chain, round 6): the self-attention's output projection runs in FRONT of this block, x = ao W_1^T + b_1 + x0, so `x` is an OUTPUT
  // (the tile's rows are stored there and read back by the same lanes as the final residual); nullptr / 0 otherwise
  const half_t* ao;        // [M][320] self-attention output (the A operand of phase 0)
  const half_t* w1;        // packed [>= 320][kpad_1]: attn1.to_out
  const float* b1;         // [320] or nullptr
  const half_t* x0;        // [M][320]: the transformer block's input (phase 0's residual)
  half_t* x1;              // [M][320]: phase 0's result (scratch the caller owns)
  int kpad_1;
};

__global__ __launch_bounds__(512, 1) void af_xattn320_kernel(const XaDev p) {
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  char* WQ = af_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prow = lane >> 3, slot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile_m = blockIdx.x;
  const int bimg = (tile_m * XA_BM) / p.N;                       // 128 | N: the tile's tokens belong to one image
  auto WO = [&](int par) { return af_smem + XA_WQ + par * (XA_WO + XA_KB + XA_VB); };
  auto KB_ = [&](int par) { return WO(par) + XA_WO; };
  auto VB = [&](int par) { return WO(par) + XA_WO + XA_KB; };

  // ---- loaders (1-KB LDS-DMA pieces of 8 rows x 128 B, chunk index XOR-ed with (row >> 1) & 7 on the source side)
  auto issue_wq = [&](int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = wave + NW * j;                              // 30 pieces: K chunk pc / 6, rows 8 (pc % 6) ..
      if (pc < 30) {
        const int c = pc / 6, row = (pc - c * 6) * 8 + prow;
        const int lc = slot ^ ((row >> 1) & 7);
        glds16(row < XA_D ? p.wq + (size_t)(h * XA_D + row) * p.kpad_q + c * 64 + lc * 8 : p.zeros, WQ + c * (48 * 128) + (pc - c * 6) * 1024);
      }
    }
  };
  auto issue_wo_k_v = [&](int h, int par) {
    char* wo = WO(par);
#pragma unroll
    for (int j = 0; j < 5; ++j) {                                // 40 pieces: rows 8 pc ..
      const int pc = wave + NW * j, row = pc * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      glds16(lc < 5 ? p.wo + (size_t)row * p.kpad_o + h * XA_D + lc * 8 : p.zeros, wo + pc * 1024);
    }
    char* kb = KB_(par);
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                // 10 pieces: keys 8 pc ..
      const int pc = wave + NW * j;
      if (pc < 10) {
        const int key = pc * 8 + prow;
        const int lc = slot ^ ((key >> 1) & 7);
        glds16(key < p.L && lc < 5 ? p.k + ((size_t)bimg * p.L + key) * p.ldk + h * XA_D + lc * 8 : p.zeros, kb + pc * 1024);
      }
    }
    char* vb = VB(par);
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                // 12 pieces: key chunk pc / 6 (64 keys), rows 8 (pc % 6) ..
      const int pc = wave + NW * j;
      if (pc < 12) {
        const int kc = pc / 6, row = (pc - kc * 6) * 8 + prow;
        const int lc = slot ^ ((row >> 1) & 7);
        const int key0 = kc * 64 + lc * 8;
        glds16(row < XA_D && key0 < p.L && key0 + 8 <= p.ldv ? p.vt + (size_t)bimg * p.vbs + (size_t)(h * XA_D + row) * p.ldv + key0 : p.zeros, vb + kc * (48 * 128) + (pc - kc * 6) * 1024);
      }
    }
  };
  issue_wq(0);
  issue_wo_k_v(0, 0);
  // the folded LayerNorm's column sums and the projection shift of all 320 q columns: staged in LDS once (a global load per head inside the
  // loop would put an L2 round trip on every head's critical path -- and, next to LDS-DMA loads in flight, a full vmcnt(0) drain)
  float* csb = reinterpret_cast<float*>(af_smem + XA_WQ + 2 * (XA_WO + XA_KB + XA_VB));
  if (tid < XA_C) {
    csb[tid] = p.ln_on ? p.cs[tid] : 0.f;
    csb[XA_C + tid] = p.bq ? p.bq[tid] : 0.f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the ds_writes above are complete before this wave reaches the loop's first barrier

  // ---- this wave's 16 tokens: x^T fragments of the ten 32-wide K steps (lane = token fr, K chunk fq), LayerNorm statistics from them
  const int m = tile_m * XA_BM + wave * 16 + fr;
  half8_t xf[10];
  {
    const half_t* xr = p.x + (size_t)(m < p.M ? m : 0) * XA_C + fq * 8;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) xf[ks] = *reinterpret_cast<const half8_t*>(xr + ks * 32);
  }
  float mean = 0.f, rstd = 1.f;
  if (p.ln_on) {
    const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const half2_t a = {xf[ks][e], xf[ks][e + 1]};
        s = __builtin_amdgcn_fdot2(a, one2, s, false);
        q = __builtin_amdgcn_fdot2(a, a, q, false);
      }
    s += __shfl_xor(s, 16, 64);
    q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    mean = s * (1.0f / XA_C);
    rstd = rsqrtf(fmaxf(q * (1.0f / XA_C) - mean * mean, 0.f) + p.ln_eps);
  }

  floatx4 acc2[20];
#pragma unroll
  for (int nt = 0; nt < 20; ++nt) acc2[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
  // fragment offsets inside a [rows x 128 B] region (rows are 16-aligned per tile: (row >> 1) & 7 == fr >> 1)
  const int rd0 = fr * 128 + (((0 * 4 + fq) ^ (fr >> 1)) * 16);   // 16-byte fragments of the q projection's weights
  const int rd1 = fr * 128 + (((1 * 4 + fq) ^ (fr >> 1)) * 16);
  // two 8-byte halves of a permuted-K fragment: rows' elements 4 fq .. 4 fq + 3 of 16-element group g (g = 0 .. 3 -> elements 16 g ..)
  auto off8 = [&](int g) { return fr * 128 + ((((2 * g) + (fq >> 1)) ^ (fr >> 1)) * 16) + (fq & 1) * 8; };
  const int o8[4] = {off8(0), off8(1), off8(2), off8(3)};
  auto frag8 = [&](const char* base, int ga, int gb) {
    const half4_t lo = *reinterpret_cast<const half4_t*>(base + o8[ga]), hi = *reinterpret_cast<const half4_t*>(base + o8[gb]);
    return half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto pack8 = [&](const floatx4& a, const floatx4& b) {
    return half8_t{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  };
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};
  // V^T pad: the 8-key chunk that straddles L arrives with whatever the caller's buffer holds behind key L - 1, and P = 0 does not silence an Inf / NaN
  // there (0 x NaN = NaN in the MFMA).  Masked keys contribute NOTHING (attention.py:196-202): the halves of a V^T fragment are AND-ed with a per-lane
  // bit mask of the live keys of its 16-key group (this lane reads keys 16 g + 4 fq + e of group g).
  unsigned long long vmask[5];
#pragma unroll
  for (int g = 0; g < 5; ++g) {
    const int live = p.L - (16 * g + 4 * fq);                    // number of live keys among this lane's four
    vmask[g] = live >= 4 ? ~0ull : (live <= 0 ? 0ull : ((1ull << (16 * live)) - 1ull));
  }
  auto vfrag8 = [&](const char* base, int ga, int gb, unsigned long long ma, unsigned long long mb) {
    unsigned long long lo = *reinterpret_cast<const unsigned long long*>(base + o8[ga]) & ma;
    unsigned long long hi = *reinterpret_cast<const unsigned long long*>(base + o8[gb]) & mb;
    const half4_t l4 = __builtin_bit_cast(half4_t, lo), h4 = __builtin_bit_cast(half4_t, hi);
    return half8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  };

  for (int h = 0; h < XA_H; ++h) {
    const int par = h & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // W_q,h and head h's W_o / K / V^T (issued a head ago) have landed
    __builtin_amdgcn_s_barrier();                                // ... for every wave; everyone has left head h - 1
    if (h + 1 < XA_H) issue_wo_k_v(h + 1, par ^ 1);
    // ---- Q^T [48 x 16] = W_q,h x^T
    floatx4 qa[3] = {zf, zf, zf};
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const half8_t wf = *reinterpret_cast<const half8_t*>(WQ + c * (48 * 128) + t * (16 * 128) + (kk ? rd1 : rd0));
          qa[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[2 * c + kk], qa[t], 0, 0, 0);
        }
    __builtin_amdgcn_s_barrier();                                // every wave is done with W_q,h
    if (h + 1 < XA_H) issue_wq(h + 1);
    // folded LayerNorm, projection shift, softmax scale (in log2 units); rows 40 .. 47 of the head are padding: exactly zero
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int n = t * 16 + 4 * fq;                             // head dimension of element 0
      floatx4 cs = zf, bq = zf;
      if (n < XA_D) {
        cs = *reinterpret_cast<const floatx4*>(csb + h * XA_D + n);
        bq = *reinterpret_cast<const floatx4*>(csb + XA_C + h * XA_D + n);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[t][e] = n < XA_D ? (rstd * (qa[t][e] - mean * cs[e]) + bq[e]) * p.scale_log2e : 0.f;
    }
    const half8_t q0 = pack8(qa[0], qa[1]), q1 = pack8(qa[2], zf);
    // ---- S^T [80 keys x 16] = K_h Q^T (K index = head dimension, permuted as the accumulators of Q^T lie)
    const char* kb = KB_(par);
    floatx4 sa[5];
    // (independent accumulators back to back: a dependent MFMA right behind its predecessor waits out the pipeline)
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) sa[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag8(kb + kt * (16 * 128), 0, 1), q0, zf, 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) sa[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag8(kb + kt * (16 * 128), 2, 3), q1, sa[kt], 0, 0, 0);
    // ---- softmax over the keys of token fr: this lane holds keys 16 kt + 4 fq + e
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kt * 16 + 4 * fq + e >= p.L) sa[kt][e] = -3.0e38f;
        mx = fmaxf(mx, sa[kt][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sa[kt][e] = __builtin_amdgcn_exp2f(sa[kt][e] - mx);
        sum += sa[kt][e];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
    // ---- O^T [48 x 16] = V_h^T P^T (K index = key, permuted as the accumulators of S^T lie: K step s = key tiles 2 s, 2 s + 1)
    const char* vb = VB(par);
    const half8_t p0 = pack8(sa[0], sa[1]), p1 = pack8(sa[2], sa[3]), p2 = pack8(sa[4], zf);
    floatx4 oa[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) oa[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag8(vb + t * (16 * 128), 0, 1, vmask[0], vmask[1]), p0, zf, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) oa[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag8(vb + t * (16 * 128), 2, 3, vmask[2], vmask[3]), p1, oa[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) oa[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag8(vb + 48 * 128 + t * (16 * 128), 0, 1, vmask[4], 0ull), p2, oa[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) oa[t][e] *= inv;
    const half8_t o0 = pack8(oa[0], oa[1]), o1 = pack8(oa[2], zf);
    // ---- OUT^T [320 x 16] += W_o,h O^T   (fragments of the next two output tiles are requested before this pair's MFMAs)
    const char* wo = WO(par);
    half8_t wa[2][2], wb[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      wa[u][0] = frag8(wo + u * (16 * 128), 0, 1);
      wa[u][1] = frag8(wo + u * (16 * 128), 2, 3);
    }
#pragma unroll
    for (int nt = 0; nt < 20; nt += 2) {
      if (nt + 2 < 20) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          wb[u][0] = frag8(wo + (nt + 2 + u) * (16 * 128), 0, 1);
          wb[u][1] = frag8(wo + (nt + 2 + u) * (16 * 128), 2, 3);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) acc2[nt + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u][0], o0, acc2[nt + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc2[nt + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u][1], o1, acc2[nt + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        wa[u][0] = wb[u][0];
        wa[u][1] = wb[u][1];
      }
    }
  }
  // ---- epilogue: output bias, residual; this lane holds channels 16 nt + 4 fq + e of token fr
  if (m < p.M) {
#pragma unroll
    for (int nt = 0; nt < 20; ++nt) {
      const int c = nt * 16 + 4 * fq;
      floatx4 v = acc2[nt];
      if (p.bo) {
        const floatx4 b = *reinterpret_cast<const floatx4*>(p.bo + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b[e];
      }
      if (p.residual) {
        const half4_t r = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * XA_C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
      }
      const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(p.out + (size_t)m * XA_C + c) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TILED form of the same block (round 4): the wave-owns-16-tokens kernel above reads a 1-KB fragment from LDS for every single MFMA of all four
// products (5.8 MB per workgroup).  Here the two projections are ordinary tiled GEMMs over a token tile that stays in LDS, and the core runs
// out of registers:
//   * R = the workgroup's 128 x 320 fp16 tile (80 KB, five 64-channel blocks of 128-byte rows, chunk index XOR (row >> 1) & 7): first x, then the
//     attention output O in place (wave h overwrites columns 40 h .. of its own rows only after every wave has finished reading x);
//   * phase A -- Q_h^T [48 x 128] = W_q,h x^T: WAVE h OWNS HEAD h for all 128 tokens.  W_q streams through a two-slot ring of 64-wide K stages
//     (320 rows x 128 B = 40 KB, the whole-line GEMM's stage); a wave reads its head's 48 rows (3 A fragments) and the tile's 8 token fragments
//     per K half: 24 MFMAs per 11 fragment reads.  LayerNorm statistics from the token fragments (folded norm2, as af_gemm_desc.ln_colsum);
//   * phase B -- per 64-token half: S^T = K_h Q_h^T, softmax over the lane's token column, O_h^T = V_h^T P^T with the accumulators as the next
//     product's B operand (k index permuted as in the kernel above; the 8-wide tails of d = 40 and of the 80 keys on 16x16x16 MFMAs).  K_h / V_h^T
//     fragments are this wave's alone: loaded ONCE from global memory into 60 registers, masked to the live keys there -- no LDS traffic at all;
//   * phase C -- out [128 x 320] = O W_o^T + b_o + residual as a 2 x 4 wave GEMM (64 x 80 per wave) on the same ring; W_o's first two stages
//     arrive during phase B.
// LDS 160 KB = R + ring exactly; 1.6 MB of LDS reads per workgroup instead of 5.8.
constexpr int XT_R = XA_BM * XA_C * 2;                 // 81,920
constexpr int XT_STAGE = XA_C * 128;                   // 40,960: 320 weight rows x 64 K
constexpr int XT_LDS = XT_R + 2 * XT_STAGE;            // 163,840

// a 16-deep k tail (the 8 head dimensions 32 .. 39, the keys 64 .. 79) as a 32-deep MFMA whose upper k half is zero on both sides.
// NOT v_mfma_f32_16x16x16_f16 (-DAF_XATTN_K16 builds it): accumulating onto the result of a 16x16x32 MFMA with the 16-deep form gave results that
// changed from run to run on the same inputs (hipcc 7.2, gfx950: tools/probes/r04g_xattn_debug.py localised it -- the q projection exact, single
// (token tile, head) cells of the attention output wrong at random); with the zero-padded 32-deep form every cell is exact.
// ROOT CAUSE (round 5, tools/probes/r05a_mfma_k16_probe.hip -> profiles/r05a_mfma_k16_probe.txt): a stand-alone chain of the two opcodes on one
// accumulator, no LDS and no barriers, differs in 199 of 200 launches whenever hipcc issues them back to back -- the 4-pass opcode reads its SrcC
// before the 8-pass one in front of it has written it, and hipcc's gfx950 hazard recogniser inserts no wait state for that pair.  A compiler hazard
// bug, not a race in this kernel: one opcode throughout (hardware-forwarded accumulate chain) is the fix.
__device__ __forceinline__ floatx4 mfma_k16(const half4_t& a, const half4_t& b, const floatx4& c) {
#ifdef AF_XATTN_K16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
#else
  const half8_t a8 = {a[0], a[1], a[2], a[3], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  const half8_t b8 = {b[0], b[1], b[2], b[3], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
#endif
}

// PRE (round 6): the self-attention's to_out + residual as phase 0 in front (attention.py:242-252: x = attn1(norm1(x)) + x, then attn2 ...): the tile R first
// holds the self-attention output, a 2 x 4 wave GEMM over it on the weight ring (W_1, five stages) gives x1 = ao W_1^T + b_1 + x0, which is stored to
// memory (the final residual: every lane reads back exactly the elements it stored, phase C has the same wave -> tile mapping) AND into R in the swizzled
// fp16 layout phase A reads -- one launch of `1,32768,320,320` (25 us, a 21 MB write and two 21 MB reads) less per transformer block of the 64 x 64 level.
template <int NT, bool PRE = false>
__global__ __launch_bounds__(512, 1) void af_xattn320t_kernel(const XaDev p) {
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  char* R = af_smem;
  char* RING = af_smem + XT_R;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prow = lane >> 3, slot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile_m = blockIdx.x;
  const int m0 = tile_m * XA_BM;
  const int bimg = m0 / p.N;
  const int h = wave;                                   // phases A / B: this wave's head
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};

  // ---- loaders: 1-KB LDS-DMA pieces of 8 rows x 128 B, chunk index XOR (row >> 1) & 7 on the source side
  auto issue_x = [&]() {                                // 80 pieces: block c (64 channels) x 16 row groups
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int pc = wave + NW * j, c = pc >> 4, row = (pc & 15) * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      const int m = m0 + row;
      glds16(m < p.M ? p.x + (size_t)m * XA_C + c * 64 + lc * 8 : p.zeros, R + c * (XA_BM * 128) + (pc & 15) * 1024);
    }
  };
  auto issue_w = [&](const half_t* w, int kpad, int st, int sl) {   // stage st (64 K) of a packed [>= 320][kpad] weight: 40 pieces
    char* dst = RING + sl * XT_STAGE;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pc = wave + NW * j, row = pc * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      glds16(w + (size_t)row * kpad + st * 64 + lc * 8, dst + pc * 1024);
    }
  };
  // fragment offsets inside a [rows x 128 B] block whose 16-row groups start at EVEN rows ((row >> 1) & 7 == ((row0 >> 1) + (fr >> 1)) & 7)
  auto frag_off = [&](int row0, int kk) { return (row0 + fr) * 128 + (((kk * 4 + fq) ^ (((row0 + fr) >> 1) & 7)) * 16); };
  if constexpr (PRE) {
    // ================= phase 0: x1 [128 x 320] = ao W_1^T + b_1 + x0, 2 x 4 waves of 64 x 80 (the layout of phase C)
#pragma unroll
    for (int j = 0; j < 10; ++j) {                      // the self-attention output tile -> R (as issue_x)
      const int pc = wave + NW * j, c = pc >> 4, row = (pc & 15) * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      const int m = m0 + row;
      glds16(m < p.M ? p.ao + (size_t)m * XA_C + c * 64 + lc * 8 : p.zeros, R + c * (XA_BM * 128) + (pc & 15) * 1024);
    }
    issue_w(p.w1, p.kpad_1, 0, 0);
    issue_w(p.w1, p.kpad_1, 1, 1);
    const int wm0 = wave & 1, wn0 = wave >> 1;
    floatx4 a0[5][4];
#pragma unroll
    for (int tn = 0; tn < 5; ++tn)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) a0[tn][tm] = zf;
#pragma nounroll
    for (int st = 0; st < 5; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (st >= 1 && st + 1 < 5) issue_w(p.w1, p.kpad_1, st + 1, (st + 1) & 1);
      const char* Ws = RING + (st & 1) * XT_STAGE;
      const char* As = R + st * (XA_BM * 128);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8_t wf[5], xf[4];
#pragma unroll
        for (int tn = 0; tn < 5; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + frag_off(wn0 * 80 + tn * 16, kk));
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(As + frag_off(wm0 * 64 + tm * 16, kk));
#pragma unroll
        for (int tn = 0; tn < 5; ++tn)
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) a0[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], a0[tn][tm], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();                       // every wave has read the last of ao and of W_1: R may take x1, the ring W_q
    issue_w(p.wq, p.kpad_q, 0, 0);                      // ... whose first two stages arrive under the epilogue below
    issue_w(p.wq, p.kpad_q, 1, 1);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const int row = wm0 * 64 + tm * 16 + fr, m = m0 + row;
#pragma unroll
      for (int tn = 0; tn < 5; ++tn) {
        const int c = wn0 * 80 + tn * 16 + 4 * fq;
        floatx4 v = a0[tn][tm];
        half4_t o = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if (m < p.M) {
          if (p.b1) {
            const floatx4 b = *reinterpret_cast<const floatx4*>(p.b1 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          const half4_t rr = *reinterpret_cast<const half4_t*>(p.x0 + (size_t)m * XA_C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
          o = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<half4_t*>(p.x1 + (size_t)m * XA_C + c) = o;
        }
        *reinterpret_cast<half4_t*>(R + (c >> 6) * (XA_BM * 128) + row * 128 + ((((c & 63) >> 3) ^ ((row >> 1) & 7)) * 16) + ((c >> 2) & 1) * 8) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's part of x1 is in R before the barrier at the top of phase A
  } else {
    issue_x();
    issue_w(p.wq, p.kpad_q, 0, 0);
    issue_w(p.wq, p.kpad_q, 1, 1);
  }

  // ================= phase A: Q_h^T [48 x 128] = W_q,h x^T
  floatx4 qa[3][8];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) qa[t][tt] = zf;
  float ls[8], lq[8];
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) ls[tt] = lq[tt] = 0.f;
  const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
  for (int st = 0; st < 5; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // x (first pass) and stage st have landed (everything this wave issued)
    __builtin_amdgcn_s_barrier();                         // ... for every wave; the other slot is free
    if (st >= 1 && st + 1 < 5) issue_w(p.wq, p.kpad_q, st + 1, (st + 1) & 1);
    const char* Ws = RING + (st & 1) * XT_STAGE;
    const char* Xs = R + st * (XA_BM * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8_t wf[3], xf[8];
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[t] = *reinterpret_cast<const half8_t*>(Ws + frag_off(h * XA_D + t * 16, kk));
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) xf[tt] = *reinterpret_cast<const half8_t*>(Xs + frag_off(tt * 16, kk));
      if (p.ln_on) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const half2_t a = {xf[tt][e], xf[tt][e + 1]};
            ls[tt] = __builtin_amdgcn_fdot2(a, one2, ls[tt], false);
            lq[tt] = __builtin_amdgcn_fdot2(a, a, lq[tt], false);
          }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) qa[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], xf[tt], qa[t][tt], 0, 0, 0);
    }
  }
  // folded LayerNorm: column sums / shift of this lane's q rows (head dimension 16 t + 4 fq + e)
  floatx4 csv[3], bqv[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int n = t * 16 + 4 * fq;
    csv[t] = (p.ln_on && n < XA_D) ? *reinterpret_cast<const floatx4*>(p.cs + h * XA_D + n) : zf;
    bqv[t] = (p.bq && n < XA_D) ? *reinterpret_cast<const floatx4*>(p.bq + h * XA_D + n) : zf;
  }

  // ---- this head's K / V^T fragments, straight from global memory (each wave's own: nothing to share through LDS), masked to the live keys;
  // requested here, behind phase A's main loop (60 registers that loop has no room for), they arrive under the projection's epilogue arithmetic
  half8_t kA32[5], vA32[3][2];
  half4_t kA16[5], vA16[3];
  typedef unsigned long long u64;
  auto ld8 = [](const half_t* ptr) { return *reinterpret_cast<const u64*>(ptr); };
  {
    // every load is issued unconditionally from an in-range address and masked afterwards by a per-lane bit mask (a select in front of a load
    // becomes an exec-masked branch with its own s_waitcnt: serialised L2 round trips).  K_h here, V_h^T behind the projection's epilogue (the
    // q accumulators leave no room for both).
    u64 klo[5], khi[5], kt8[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
      const int key = kt * 16 + fr;
      const half_t* kp = p.k + ((size_t)bimg * p.L + (key < p.L ? key : 0)) * p.ldk + h * XA_D;
      klo[kt] = ld8(kp + 4 * fq);
      khi[kt] = ld8(kp + 16 + 4 * fq);
      kt8[kt] = ld8(kp + 32 + 4 * (fq & 1));           // d 32 .. 39 (fq < 2; the other lanes read a valid address and are masked)
    }
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
      const u64 km = (kt * 16 + fr) < p.L ? ~0ull : 0ull;
      const half4_t lo = __builtin_bit_cast(half4_t, klo[kt] & km), hi = __builtin_bit_cast(half4_t, khi[kt] & km);
      kA32[kt] = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      kA16[kt] = __builtin_bit_cast(half4_t, kt8[kt] & (fq < 2 ? km : 0ull));
    }
  }
  __builtin_amdgcn_s_barrier();                           // every wave is done with x and with the W_q ring: R may take O, the ring W_o
  issue_w(p.wo, p.kpad_o, 0, 0);
  issue_w(p.wo, p.kpad_o, 1, 1);
  // folded LayerNorm, shift, softmax scale (log2 units); padding rows of the head exactly zero; packed as the B operands of S^T
  half8_t q0[8];
  half4_t q1[8];
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    float mean = 0.f, rstd = 1.f;
    if (p.ln_on) {
      float s = ls[tt], q = lq[tt];
      s += __shfl_xor(s, 16, 64);
      q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      mean = s * (1.0f / XA_C);
      rstd = rsqrtf(fmaxf(q * (1.0f / XA_C) - mean * mean, 0.f) + p.ln_eps);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int n = t * 16 + 4 * fq;
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[t][tt][e] = n < XA_D ? (rstd * (qa[t][tt][e] - mean * csv[t][e]) + bqv[t][e]) * p.scale_log2e : 0.f;
    }
    q0[tt] = half8_t{(half_t)qa[0][tt][0], (half_t)qa[0][tt][1], (half_t)qa[0][tt][2], (half_t)qa[0][tt][3],
                     (half_t)qa[1][tt][0], (half_t)qa[1][tt][1], (half_t)qa[1][tt][2], (half_t)qa[1][tt][3]};
    q1[tt] = half4_t{(half_t)qa[2][tt][0], (half_t)qa[2][tt][1], (half_t)qa[2][tt][2], (half_t)qa[2][tt][3]};
  }

  {
    auto live_mask = [&](int first) {                  // keys first .. first + 3: bits of those < L
      const int live = p.L - first;
      return live >= 4 ? ~0ull : (live <= 0 ? 0ull : ((1ull << (16 * live)) - 1ull));
    };
    u64 vlo[3][2], vhi[3][2], vt8[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int row = t * 16 + fr;                     // head dimension; rows 40 .. 47 are padding: zero
      const half_t* vp = p.vt + (size_t)bimg * p.vbs + (size_t)(h * XA_D + (row < XA_D ? row : 0)) * p.ldv;
      auto at = [&](int key0) { return vp + (key0 + 4 <= p.ldv ? key0 : 0); };     // key0 % 4 == 0 and ldv % 8 == 0: the 8 bytes stay inside the row
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vlo[t][s2] = ld8(at(32 * s2 + 4 * fq));
        vhi[t][s2] = ld8(at(32 * s2 + 16 + 4 * fq));
      }
      vt8[t] = ld8(at(64 + 4 * fq));
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const u64 rm = (t * 16 + fr) < XA_D ? ~0ull : 0ull;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const half4_t lo = __builtin_bit_cast(half4_t, vlo[t][s2] & rm & live_mask(32 * s2 + 4 * fq));
        const half4_t hi = __builtin_bit_cast(half4_t, vhi[t][s2] & rm & live_mask(32 * s2 + 16 + 4 * fq));
        vA32[t][s2] = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      vA16[t] = __builtin_bit_cast(half4_t, vt8[t] & rm & live_mask(64 + 4 * fq));
    }
  }
  // ================= phase B: the 77-key core of head h, NT 16-token tiles at a time (everything in registers: the K / V^T fragments are reused
  // from registers; NT independent chains give the MFMA -> softmax -> MFMA sequence its instruction-level parallelism.  Measured per layer at U-Net
  // batch 8: NT = 1 80.6 us (one dependent chain), NT = 2 46.9, NT = 4 49.6 (208 bytes of spills: 80 accumulators next to 108 fragment registers))
#pragma unroll
  for (int t0 = 0; t0 < 8; t0 += NT) {
    floatx4 sa[5][NT];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int u = 0; u < NT; ++u) sa[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kA32[kt], q0[t0 + u], zf, 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int u = 0; u < NT; ++u) sa[kt][u] = mfma_k16(kA16[kt], q1[t0 + u], sa[kt][u]);
    float inv[NT];
    half8_t p0[NT], p1[NT];
    half4_t p2[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      // softmax over the keys of token (t0 + u, fr): this lane holds keys 16 kt + 4 fq + e
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (kt * 16 + 4 * fq + e >= p.L) sa[kt][u][e] = -3.0e38f;
          mx = fmaxf(mx, sa[kt][u][e]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sa[kt][u][e] = __builtin_amdgcn_exp2f(sa[kt][u][e] - mx);
          sum += sa[kt][u][e];
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      inv[u] = __builtin_amdgcn_rcpf(sum);
      p0[u] = half8_t{(half_t)sa[0][u][0], (half_t)sa[0][u][1], (half_t)sa[0][u][2], (half_t)sa[0][u][3],
                      (half_t)sa[1][u][0], (half_t)sa[1][u][1], (half_t)sa[1][u][2], (half_t)sa[1][u][3]};
      p1[u] = half8_t{(half_t)sa[2][u][0], (half_t)sa[2][u][1], (half_t)sa[2][u][2], (half_t)sa[2][u][3],
                      (half_t)sa[3][u][0], (half_t)sa[3][u][1], (half_t)sa[3][u][2], (half_t)sa[3][u][3]};
      p2[u] = half4_t{(half_t)sa[4][u][0], (half_t)sa[4][u][1], (half_t)sa[4][u][2], (half_t)sa[4][u][3]};
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      floatx4 o[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vA32[t][0], p0[u], zf, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vA32[t][1], p1[u], o[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = mfma_k16(vA16[t], p2[u], o[u]);
      // O_h -> R, columns 40 h + 16 t + 4 fq + e of the token rows: this wave's own columns, 8 bytes per lane
      const int n = t * 16 + 4 * fq;
      if (n < XA_D) {
        const int c = h * XA_D + n;                     // channel: block c >> 6, 16-byte chunk (c & 63) >> 3, 8-byte half (c >> 2) & 1
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const int row = (t0 + u) * 16 + fr;
          const half4_t o4 = {(half_t)(o[u][0] * inv[u]), (half_t)(o[u][1] * inv[u]), (half_t)(o[u][2] * inv[u]), (half_t)(o[u][3] * inv[u])};
          *reinterpret_cast<half4_t*>(R + (c >> 6) * (XA_BM * 128) + row * 128 + ((((c & 63) >> 3) ^ ((row >> 1) & 7)) * 16) + ((c >> 2) & 1) * 8) = o4;
        }
      }
    }
  }

  // ================= phase C: out = O W_o^T + b_o + residual, 2 x 4 waves of 64 x 80
  const int wm = wave & 1, wn = wave >> 1;
  floatx4 acc[5][4];
#pragma unroll
  for (int tn = 0; tn < 5; ++tn)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = zf;
  for (int st = 0; st < 5; ++st) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // W_o stage st has landed; (first pass) this wave's O stores are in LDS
    __builtin_amdgcn_s_barrier();
    if (st >= 1 && st + 1 < 5) issue_w(p.wo, p.kpad_o, st + 1, (st + 1) & 1);
    const char* Ws = RING + (st & 1) * XT_STAGE;
    const char* Os = R + st * (XA_BM * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8_t wf[5], xf[4];
#pragma unroll
      for (int tn = 0; tn < 5; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + frag_off(wn * 80 + tn * 16, kk));
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(Os + frag_off(wm * 64 + tm * 16, kk));
#pragma unroll
      for (int tn = 0; tn < 5; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + wm * 64 + tm * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int tn = 0; tn < 5; ++tn) {
      const int c = wn * 80 + tn * 16 + 4 * fq;
      floatx4 v = acc[tn][tm];
      if (p.bo) {
        const floatx4 b = *reinterpret_cast<const floatx4*>(p.bo + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b[e];
      }
      const half_t* res = PRE ? p.x1 : p.residual;      // PRE: what this very lane stored in phase 0
      if (res) {
        const half4_t rr = *reinterpret_cast<const half4_t*>(res + (size_t)m * XA_C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
      }
      const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(p.out + (size_t)m * XA_C + c) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same block at C = 640 (8 heads of 80; the 32 x 32 level), ONE launch (round 6; the form the round-5 review prescribed: 64-token tiles, an 80 KB
// row tile + the 80 KB two-slot weight ring, 128 workgroups at M = 8192):
//   * R = the workgroup's 64 x 640 fp16 tile (ten 64-channel blocks of [64 rows x 128 B], swizzled as above): x, then O in place;
//   * a 640-row weight goes through the ring of af_xattn320t_kernel as 2 row halves x 10 K stages of [320 rows x 64 K] (40 KB) -- 20 stages per projection;
//   * phase A: row half nh holds heads 4 nh .. 4 nh + 3; wave w computes Q^T [80 x 32] of head 4 nh + (w & 3) for the token half w >> 2, so over the two
//     halves a wave OWNS TWO HEADS x 32 TOKENS (80 accumulator registers); LayerNorm statistics from the token fragments of the first half's ten stages;
//   * phase B: per owned head the K_h / V_h^T fragments (d = 80 = two 32-deep k steps + a 16-deep tail; 80 keys likewise) come from global memory into 100
//     registers, the two 16-token tiles run as two independent chains;
//   * phase C: out [64 x 640] as 2 x 4 waves of 32 x 80 per row half (acc [2][5][2]).
// Every workgroup streams BOTH whole weights (1.6 MB) through its ring -- what the three-launch form's 2-D tiling avoids; measured: see DESIGN.md 8.0.
constexpr int X6_C = 640, X6_D = 80, X6_BM = 64;
constexpr int X6_BLK = X6_BM * 128;                    // 8,192: one 64-channel block of R
constexpr int X6_R = (X6_C / 64) * X6_BLK;             // 81,920
constexpr int X6_LDS = X6_R + 2 * XT_STAGE;            // 163,840

template <int NT>
__global__ __launch_bounds__(512, 1) void af_xattn640t_kernel(const XaDev p) {
  constexpr int NW = 8, NS = 20;                       // NS: stages per projection
  static_assert(NT == 2, "a wave owns two 16-token tiles per head");
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  char* R = af_smem;
  char* RING = af_smem + X6_R;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prow = lane >> 3, slot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * X6_BM;
  const int bimg = m0 / p.N;
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};

  auto issue_x = [&]() {                                // 80 pieces: block c (64 channels) x 8 row groups
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int pc = wave + NW * j, c = pc >> 3, row = (pc & 7) * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      const int m = m0 + row;
      glds16(m < p.M ? p.x + (size_t)m * X6_C + c * 64 + lc * 8 : p.zeros, R + c * X6_BLK + (pc & 7) * 1024);
    }
  };
  auto issue_w = [&](const half_t* w, int kpad, int s, int sl) {   // stage s = (row half s / 10, K stage s % 10) of a packed [>= 640][kpad] weight: 40 pieces
    char* dst = RING + sl * XT_STAGE;
    const int nh = s >= 10 ? 1 : 0, st = s - 10 * nh;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pc = wave + NW * j, row = pc * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      glds16(w + (size_t)(nh * 320 + row) * kpad + st * 64 + lc * 8, dst + pc * 1024);
    }
  };
  auto frag_off = [&](int row0, int kk) { return (row0 + fr) * 128 + (((kk * 4 + fq) ^ (((row0 + fr) >> 1) & 7)) * 16); };

  issue_x();
  issue_w(p.wq, p.kpad_q, 0, 0);
  issue_w(p.wq, p.kpad_q, 1, 1);

  // ================= phase A: Q_h^T [80 x 32] per row half
  const int hh = wave & 3, th = wave >> 2;
  floatx4 qa[2][5][2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) qa[nh][t][tt] = zf;
  float ls[2] = {0.f, 0.f}, lq[2] = {0.f, 0.f};
  const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
#pragma nounroll
    for (int st = 0; st < 10; ++st) {
      const int s = nh * 10 + st;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s >= 1 && s + 1 < NS) issue_w(p.wq, p.kpad_q, s + 1, (s + 1) & 1);
      const char* Ws = RING + (s & 1) * XT_STAGE;
      const char* Xs = R + st * X6_BLK;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8_t wf[5], xf[2];
#pragma unroll
        for (int t = 0; t < 5; ++t) wf[t] = *reinterpret_cast<const half8_t*>(Ws + frag_off(hh * X6_D + t * 16, kk));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) xf[tt] = *reinterpret_cast<const half8_t*>(Xs + frag_off(th * 32 + tt * 16, kk));
        if (nh == 0 && p.ln_on) {
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const half2_t a = {xf[tt][e], xf[tt][e + 1]};
              ls[tt] = __builtin_amdgcn_fdot2(a, one2, ls[tt], false);
              lq[tt] = __builtin_amdgcn_fdot2(a, a, lq[tt], false);
            }
        }
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) qa[nh][t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], xf[tt], qa[nh][t][tt], 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_barrier();                           // every wave is done with x and with the W_q ring: R may take O, the ring W_o
  issue_w(p.wo, p.kpad_o, 0, 0);
  issue_w(p.wo, p.kpad_o, 1, 1);
  // folded LayerNorm, shift, softmax scale (log2 units); packed as the B operands of S^T (k slot 8 fq + j <-> head dimension 16 a + 4 fq + j, j < 4; 16 b + 4 fq + j - 4)
  float mean[2], rstd[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    mean[tt] = 0.f;
    rstd[tt] = 1.f;
    if (p.ln_on) {
      float s = ls[tt], q = lq[tt];
      s += __shfl_xor(s, 16, 64);
      q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      mean[tt] = s * (1.0f / X6_C);
      rstd[tt] = rsqrtf(fmaxf(q * (1.0f / X6_C) - mean[tt] * mean[tt], 0.f) + p.ln_eps);
    }
  }
  half8_t q0[2][2][2];
  half4_t q1[2][2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int h = nh * 4 + hh;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int n = h * X6_D + t * 16 + 4 * fq;
      const floatx4 csv = p.ln_on ? *reinterpret_cast<const floatx4*>(p.cs + n) : zf;
      const floatx4 bqv = p.bq ? *reinterpret_cast<const floatx4*>(p.bq + n) : zf;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) qa[nh][t][tt][e] = (rstd[tt] * (qa[nh][t][tt][e] - mean[tt] * csv[e]) + bqv[e]) * p.scale_log2e;
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        q0[nh][s2][tt] = half8_t{(half_t)qa[nh][2 * s2][tt][0],     (half_t)qa[nh][2 * s2][tt][1],     (half_t)qa[nh][2 * s2][tt][2],     (half_t)qa[nh][2 * s2][tt][3],
                                 (half_t)qa[nh][2 * s2 + 1][tt][0], (half_t)qa[nh][2 * s2 + 1][tt][1], (half_t)qa[nh][2 * s2 + 1][tt][2], (half_t)qa[nh][2 * s2 + 1][tt][3]};
      q1[nh][tt] = half4_t{(half_t)qa[nh][4][tt][0], (half_t)qa[nh][4][tt][1], (half_t)qa[nh][4][tt][2], (half_t)qa[nh][4][tt][3]};
    }
  }

  // ================= phase B: the 77-key core of the wave's two heads x two 16-token tiles
  typedef unsigned long long u64;
  auto ld8 = [](const half_t* ptr) { return *reinterpret_cast<const u64*>(ptr); };
  auto live_mask = [&](int first) {                    // keys first .. first + 3: bits of those < L
    const int live = p.L - first;
    return live >= 4 ? ~0ull : (live <= 0 ? 0ull : ((1ull << (16 * live)) - 1ull));
  };
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int h = nh * 4 + hh;
    half8_t kA32[5][2], vA32[5][2];
    half4_t kA16[5], vA16[5];
    {
      // unconditional loads from in-range addresses, masked afterwards (a select in front of a load becomes an exec-masked branch with its own wait)
      u64 klo[5][2], khi[5][2], kt8[5];
#pragma unroll
      for (int kt = 0; kt < 5; ++kt) {
        const int key = kt * 16 + fr;
        const half_t* kp = p.k + ((size_t)bimg * p.L + (key < p.L ? key : 0)) * p.ldk + h * X6_D;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          klo[kt][s2] = ld8(kp + 32 * s2 + 4 * fq);
          khi[kt][s2] = ld8(kp + 32 * s2 + 16 + 4 * fq);
        }
        kt8[kt] = ld8(kp + 64 + 4 * fq);
      }
      u64 vlo[5][2], vhi[5][2], vt8[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const half_t* vp = p.vt + (size_t)bimg * p.vbs + (size_t)(h * X6_D + t * 16 + fr) * p.ldv;
        auto at = [&](int key0) { return vp + (key0 + 4 <= p.ldv ? key0 : 0); };     // key0 % 4 == 0 and ldv % 8 == 0: the 8 bytes stay inside the row
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          vlo[t][s2] = ld8(at(32 * s2 + 4 * fq));
          vhi[t][s2] = ld8(at(32 * s2 + 16 + 4 * fq));
        }
        vt8[t] = ld8(at(64 + 4 * fq));
      }
#pragma unroll
      for (int kt = 0; kt < 5; ++kt) {
        const u64 km = (kt * 16 + fr) < p.L ? ~0ull : 0ull;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half4_t lo = __builtin_bit_cast(half4_t, klo[kt][s2] & km), hi = __builtin_bit_cast(half4_t, khi[kt][s2] & km);
          kA32[kt][s2] = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        kA16[kt] = __builtin_bit_cast(half4_t, kt8[kt] & km);
      }
#pragma unroll
      for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half4_t lo = __builtin_bit_cast(half4_t, vlo[t][s2] & live_mask(32 * s2 + 4 * fq));
          const half4_t hi = __builtin_bit_cast(half4_t, vhi[t][s2] & live_mask(32 * s2 + 16 + 4 * fq));
          vA32[t][s2] = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        vA16[t] = __builtin_bit_cast(half4_t, vt8[t] & live_mask(64 + 4 * fq));
      }
    }
    floatx4 sa[5][NT];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int u = 0; u < NT; ++u) sa[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kA32[kt][0], q0[nh][0][u], zf, 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int u = 0; u < NT; ++u) sa[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kA32[kt][1], q0[nh][1][u], sa[kt][u], 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int u = 0; u < NT; ++u) sa[kt][u] = mfma_k16(kA16[kt], q1[nh][u], sa[kt][u]);
    float inv[NT];
    half8_t p0[NT], p1[NT];
    half4_t p2[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (kt * 16 + 4 * fq + e >= p.L) sa[kt][u][e] = -3.0e38f;
          mx = fmaxf(mx, sa[kt][u][e]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sa[kt][u][e] = __builtin_amdgcn_exp2f(sa[kt][u][e] - mx);
          sum += sa[kt][u][e];
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      inv[u] = __builtin_amdgcn_rcpf(sum);
      p0[u] = half8_t{(half_t)sa[0][u][0], (half_t)sa[0][u][1], (half_t)sa[0][u][2], (half_t)sa[0][u][3],
                      (half_t)sa[1][u][0], (half_t)sa[1][u][1], (half_t)sa[1][u][2], (half_t)sa[1][u][3]};
      p1[u] = half8_t{(half_t)sa[2][u][0], (half_t)sa[2][u][1], (half_t)sa[2][u][2], (half_t)sa[2][u][3],
                      (half_t)sa[3][u][0], (half_t)sa[3][u][1], (half_t)sa[3][u][2], (half_t)sa[3][u][3]};
      p2[u] = half4_t{(half_t)sa[4][u][0], (half_t)sa[4][u][1], (half_t)sa[4][u][2], (half_t)sa[4][u][3]};
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      floatx4 o[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vA32[t][0], p0[u], zf, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vA32[t][1], p1[u], o[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = mfma_k16(vA16[t], p2[u], o[u]);
      const int c = h * X6_D + t * 16 + 4 * fq;         // channel: block c >> 6, 16-byte chunk (c & 63) >> 3, 8-byte half (c >> 2) & 1
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int row = th * 32 + u * 16 + fr;
        const half4_t o4 = {(half_t)(o[u][0] * inv[u]), (half_t)(o[u][1] * inv[u]), (half_t)(o[u][2] * inv[u]), (half_t)(o[u][3] * inv[u])};
        *reinterpret_cast<half4_t*>(R + (c >> 6) * X6_BLK + row * 128 + ((((c & 63) >> 3) ^ ((row >> 1) & 7)) * 16) + ((c >> 2) & 1) * 8) = o4;
      }
    }
  }

  // ================= phase C: out = O W_o^T + b_o + residual, 2 x 4 waves of 32 x 80 per row half
  const int wm = wave & 1, wn = wave >> 1;
  floatx4 acc[2][5][2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int tn = 0; tn < 5; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) acc[nh][tn][tm] = zf;
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
#pragma nounroll
    for (int st = 0; st < 10; ++st) {
      const int s = nh * 10 + st;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // W_o stage s has landed; (first pass) this wave's O stores are in LDS
      __builtin_amdgcn_s_barrier();
      if (s >= 1 && s + 1 < NS) issue_w(p.wo, p.kpad_o, s + 1, (s + 1) & 1);
      const char* Ws = RING + (s & 1) * XT_STAGE;
      const char* Os = R + st * X6_BLK;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8_t wf[5], xf[2];
#pragma unroll
        for (int tn = 0; tn < 5; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + frag_off(wn * 80 + tn * 16, kk));
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(Os + frag_off(wm * 32 + tm * 16, kk));
#pragma unroll
        for (int tn = 0; tn < 5; ++tn)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[nh][tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[nh][tn][tm], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 32 + tm * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int tn = 0; tn < 5; ++tn) {
        const int c = nh * 320 + wn * 80 + tn * 16 + 4 * fq;
        floatx4 v = acc[nh][tn][tm];
        if (p.bo) {
          const floatx4 b = *reinterpret_cast<const floatx4*>(p.bo + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if (p.residual) {
          const half4_t rr = *reinterpret_cast<const half4_t*>(p.residual + (size_t)m * X6_C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
        }
        const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4_t*>(p.out + (size_t)m * X6_C + c) = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm's normalising pass + the 1x1 convolution behind it at C = 320, ONE launch (af_gn_proj_fused; round 4): the SpatialTransformer's
// `proj_in(norm(x))` (attention.py:283-291).  Unfused: af_groupnorm_apply writes the normalised [M, 320] tensor (42 MB of traffic at U-Net
// batch 8, 14 us) and a 16.6 us GEMM reads it back.  Here a workgroup normalises its 128-token tile straight into the LDS tile R (the layout of
// af_xattn320t_kernel: five 64-channel blocks of 128-byte rows, swizzled) -- statistics from the PARTIAL sums the producing convolution left
// (af_gemm_desc.gn_partials), folded by every workgroup for its batch item -- and runs the projection as a 2 x 4 wave GEMM over R on the two-slot
// weight ring; the first two weight stages stream in under the normalisation.  The normalised tensor never exists in memory.
struct GpDev {
  const half_t* x;          // [M][320]
  const float* partials;    // [B][128][32][2] (sum, M2 about the block's mean) per 128-row block and group
  const float* gamma;
  const float* beta;
  const half_t* w;          // packed [>= 320][kpad]
  const float* bias;        // [320] or nullptr
  half_t* out;              // [M][320]
  const half_t* zeros;
  int M, HW, nblk, kpad;
  float eps;
};

__global__ __launch_bounds__(512, 1) void af_gn_proj320_kernel(const GpDev p) {
  constexpr int NW = 8, G = 32, CPG = XA_C / G;
  extern __shared__ __attribute__((aligned(16))) char af_smem[];
  char* R = af_smem;
  char* RING = af_smem + XT_R;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prow = lane >> 3, slot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * XA_BM;
  const int bimg = m0 / p.HW;
  const floatx4 zf = {0.f, 0.f, 0.f, 0.f};
  auto issue_w = [&](int st, int sl) {
    char* dst = RING + sl * XT_STAGE;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pc = wave + NW * j, row = pc * 8 + prow;
      const int lc = slot ^ ((row >> 1) & 7);
      glds16(p.w + (size_t)row * p.kpad + st * 64 + lc * 8, dst + pc * 1024);
    }
  };
  issue_w(0, 0);
  issue_w(1, 1);
  // ---- this thread's share of the tile: chunk ch (8 channels) of rows r0, r0 + 12, ...; its loads are requested before the statistics fold
  const bool act = tid < 480;
  const int ch = tid % 40, r0 = tid / 40;
  half8_t xv[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const int r = r0 + 12 * k;
    if (act && r < XA_BM && m0 + r < p.M) xv[k] = *reinterpret_cast<const half8_t*>(p.x + (size_t)(m0 + r) * XA_C + ch * 8);
  }
  floatx4 gm[2], bt[2];
  if (act) {
    gm[0] = *reinterpret_cast<const floatx4*>(p.gamma + ch * 8);
    gm[1] = *reinterpret_cast<const floatx4*>(p.gamma + ch * 8 + 4);
    bt[0] = *reinterpret_cast<const floatx4*>(p.beta + ch * 8);
    bt[1] = *reinterpret_cast<const floatx4*>(p.beta + ch * 8 + 4);
  }
  // ---- fold the producer's partial sums of this batch item: 16 lanes of blocks x 32 groups, then 16 -> 1 through LDS (inside R, which is still free)
  float* fold = reinterpret_cast<float*>(R);                 // [16][32][3] (n, sum, M2)
  float* mr = fold + 16 * G * 3;                             // [32][2] (mean, rstd)
  {
    // (sum, M2) partials of 128-row blocks, merged pairwise (af_common.h, GroupNorm partial statistics)
    const int g = tid & 31, kl = tid >> 5;
    GnAcc acc = {0.f, 0.f, 0.f};
    for (int k = kl; k < p.nblk; k += 16) {
      const float* w = p.partials + (((size_t)bimg * 128 + k) * G + g) * 2;
      acc = gn_acc_merge(acc, GnAcc{128.f * CPG, w[0], w[1]});
    }
    fold[(kl * G + g) * 3] = acc.n;
    fold[(kl * G + g) * 3 + 1] = acc.s;
    fold[(kl * G + g) * 3 + 2] = acc.m2;
  }
  __syncthreads();
  if (tid < G) {
    GnAcc acc = {fold[tid * 3], fold[tid * 3 + 1], fold[tid * 3 + 2]};
#pragma unroll
    for (int k = 1; k < 16; ++k) acc = gn_acc_merge(acc, GnAcc{fold[(k * G + tid) * 3], fold[(k * G + tid) * 3 + 1], fold[(k * G + tid) * 3 + 2]});
    const float inv_n = 1.0f / ((float)p.HW * (float)CPG);
    const float mean = acc.s * inv_n;
    const float var = fmaxf(acc.m2 * inv_n, 0.f);
    mr[tid * 2] = mean;
    mr[tid * 2 + 1] = rsqrtf(var + p.eps);
  }
  __syncthreads();
  float sc[8], sh[8];
  if (act) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (ch * 8 + e) / CPG;
      const float k = mr[g * 2 + 1] * gm[e >> 2][e & 3];
      sc[e] = k;
      sh[e] = bt[e >> 2][e & 3] - mr[g * 2] * k;
    }
  }
  __syncthreads();                                           // everyone has read the statistics: R may take the tile
  if (act) {
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const int r = r0 + 12 * k;
      if (r < XA_BM) {
        half8_t o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m0 + r < p.M) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)xv[k][e] * sc[e] + sh[e]);
        }
        *reinterpret_cast<half8_t*>(R + (ch >> 3) * (XA_BM * 128) + r * 128 + (((ch & 7) ^ ((r >> 1) & 7)) * 16)) = o;
      }
    }
  }
  // ---- out = R W^T + b: 2 x 4 waves of 64 x 80 over the five 64-wide stages
  auto frag_off = [&](int row0, int kk) { return (row0 + fr) * 128 + (((kk * 4 + fq) ^ (((row0 + fr) >> 1) & 7)) * 16); };
  const int wm = wave & 1, wn = wave >> 1;
  floatx4 acc[5][4];
#pragma unroll
  for (int tn = 0; tn < 5; ++tn)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = zf;
  for (int st = 0; st < 5; ++st) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // weight stage st has landed; (first pass) this thread's tile rows are in LDS
    __builtin_amdgcn_s_barrier();
    if (st >= 1 && st + 1 < 5) issue_w(st + 1, (st + 1) & 1);
    const char* Ws = RING + (st & 1) * XT_STAGE;
    const char* Xs = R + st * (XA_BM * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8_t wf[5], xf[4];
#pragma unroll
      for (int tn = 0; tn < 5; ++tn) wf[tn] = *reinterpret_cast<const half8_t*>(Ws + frag_off(wn * 80 + tn * 16, kk));
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) xf[tm] = *reinterpret_cast<const half8_t*>(Xs + frag_off(wm * 64 + tm * 16, kk));
#pragma unroll
      for (int tn = 0; tn < 5; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + wm * 64 + tm * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int tn = 0; tn < 5; ++tn) {
      const int c = wn * 80 + tn * 16 + 4 * fq;
      floatx4 v = acc[tn][tm];
      if (p.bias) {
        const floatx4 b = *reinterpret_cast<const floatx4*>(p.bias + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b[e];
      }
      const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(p.out + (size_t)m * XA_C + c) = o;
    }
  }
}

}  // namespace

// Whole cross-attention block at C = 320 (af_xattn320_kernel above): q projection with the LayerNorm folded in (wq / bq / ln_colsum as
// ops.pack_matrix_ln packs them; ln_colsum NULL = x is used as it is), 77-key attention over this layer's slices of the batched K / V^T
// projection, output projection, bias and residual.
extern "C" int af_xattn_fused(const void* x, const void* wq, const void* bq, const void* ln_colsum, float ln_eps, int kpad_q, const void* k, int ldk,
                              const void* vt, int64_t vt_batch_stride, int ldv, const void* wo, const void* bo, int kpad_o, const void* residual,
                              void* out, int B, int N, int L, int C, int heads, float scale, const void* zeros, void* stream) {
  AF_REQUIRE(x && wq && k && vt && wo && out && zeros, "af_xattn_fused: null pointer");
  AF_SUPPORTED((C == XA_C || C == X6_C) && heads == XA_H, "af_xattn_fused: built for C = 320 / 640, 8 heads (the 64 x 64 and 32 x 32 levels of SD-1.5)");
  AF_REQUIRE(B > 0 && N > 0 && N % (C == XA_C ? XA_BM : X6_BM) == 0, "af_xattn_fused: tokens per image must be a positive multiple of 128 (C = 320) / 64 (C = 640)");
  AF_SUPPORTED(L > 0 && L <= 80, "af_xattn_fused: at most 80 keys");
  AF_REQUIRE(kpad_q >= C && kpad_q % 64 == 0 && kpad_o >= C && kpad_o % 64 == 0, "af_xattn_fused: weight row strides must be 64-multiples covering C");
  AF_REQUIRE(ldk >= C && ldk % 8 == 0 && ldv >= L && ldv % 8 == 0 && vt_batch_stride % 8 == 0, "af_xattn_fused: K / V^T strides must keep 16-byte alignment");
  XaDev p{};
  p.x = (const half_t*)x;
  p.wq = (const half_t*)wq;
  p.bq = (const float*)bq;
  p.cs = (const float*)ln_colsum;
  p.ln_on = ln_colsum != nullptr;
  p.k = (const half_t*)k;
  p.vt = (const half_t*)vt;
  p.wo = (const half_t*)wo;
  p.bo = (const float*)bo;
  p.residual = (const half_t*)residual;
  p.out = (half_t*)out;
  p.zeros = (const half_t*)zeros;
  p.M = B * N;
  p.N = N;
  p.L = L;
  p.kpad_q = kpad_q;
  p.kpad_o = kpad_o;
  p.ldk = ldk;
  p.ldv = ldv;
  p.vbs = (long)vt_batch_stride;
  p.ln_eps = ln_eps;
  p.scale_log2e = scale * 1.4426950408889634f;
  if (C == X6_C) {
    AF_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)bo | (uintptr_t)bq | (uintptr_t)ln_colsum) & 15) == 0,
               "af_xattn_fused: x / out / residual / bo / bq / ln_colsum must be 16-byte aligned");
    static bool attr6 = false;
    if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_xattn640t_kernel<2>), X6_LDS, attr6, "af_xattn_fused")) return af_check_launch("af_xattn_fused");
    AfLaunchScope scope6(AF_FAM_XATTN, stream);
    hipLaunchKernelGGL(af_xattn640t_kernel<2>, dim3(p.M / X6_BM), dim3(512), X6_LDS, (hipStream_t)stream, p);
    return af_check_launch("af_xattn_fused(C = 640)");
  }
  static bool attr_set = false;
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_xattn320_kernel), XA_LDS, attr_set, "af_xattn_fused")) return af_check_launch("af_xattn_fused");
  AfLaunchScope scope(AF_FAM_XATTN, stream);
  // AF_XATTN_TILED (default 1): the tiled form (af_xattn320t_kernel); 0 = the wave-owns-16-tokens form.  Re-read per call under AF_GEMM3_ABLATE_DYNAMIC (A/B runs).
  static const int tiled_env = getenv("AF_XATTN_TILED") ? atoi(getenv("AF_XATTN_TILED")) : 1;
  static const bool dyn = getenv("AF_GEMM3_ABLATE_DYNAMIC") != nullptr;
  const int tiled = dyn ? (getenv("AF_XATTN_TILED") ? atoi(getenv("AF_XATTN_TILED")) : 1) : tiled_env;
  if (tiled) {
    static bool attr_t = false;
    if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_xattn320t_kernel<2>), XT_LDS, attr_t, "af_xattn_fused")) return af_check_launch("af_xattn_fused");
    hipLaunchKernelGGL(af_xattn320t_kernel<2>, dim3(p.M / XA_BM), dim3(512), XT_LDS, (hipStream_t)stream, p);
    return af_check_launch("af_xattn_fused(tiled)");
  }
  hipLaunchKernelGGL(af_xattn320_kernel, dim3(p.M / XA_BM), dim3(512), XA_LDS, (hipStream_t)stream, p);
  return af_check_launch("af_xattn_fused");
}

// The chained form: the self-attention's output projection + residual in front of the block (af_xattn320t_kernel<2, true>).  ao = the self-attention
// core's output, w1 / b1 = attn1.to_out, x0 = the transformer block's input; x1 = ao W_1^T + b_1 + x0 is written to the caller's scratch `x1` and is both
// the cross-attention's input and its residual.  The other arguments as af_xattn_fused (which has no residual argument here: it is x1).
extern "C" int af_xattn_chain(const void* ao, const void* w1, const void* b1, int kpad_1, const void* x0, void* x1, const void* wq, const void* bq,
                              const void* ln_colsum, float ln_eps, int kpad_q, const void* k, int ldk, const void* vt, int64_t vt_batch_stride, int ldv,
                              const void* wo, const void* bo, int kpad_o, void* out, int B, int N, int L, int C, int heads, float scale, const void* zeros,
                              void* stream) {
  AF_REQUIRE(ao && w1 && x0 && x1 && wq && k && vt && wo && out && zeros, "af_xattn_chain: null pointer");
  AF_SUPPORTED(C == XA_C && heads == XA_H, "af_xattn_chain: built for C = 320, 8 heads (the 64 x 64 level of SD-1.5)");
  AF_REQUIRE(B > 0 && N > 0 && N % XA_BM == 0, "af_xattn_chain: tokens per image must be a positive multiple of 128");
  AF_SUPPORTED(L > 0 && L <= 80, "af_xattn_chain: at most 80 keys");
  AF_REQUIRE(kpad_1 >= C && kpad_1 % 64 == 0 && kpad_q >= C && kpad_q % 64 == 0 && kpad_o >= C && kpad_o % 64 == 0, "af_xattn_chain: weight row strides must be 64-multiples covering C");
  AF_REQUIRE(ldk >= C && ldk % 8 == 0 && ldv >= L && ldv % 8 == 0 && vt_batch_stride % 8 == 0, "af_xattn_chain: K / V^T strides must keep 16-byte alignment");
  AF_REQUIRE((((uintptr_t)ao | (uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)out | (uintptr_t)b1) & 15) == 0 && x1 != out && x1 != x0 && x1 != ao,
             "af_xattn_chain: ao / x0 / x1 / out / b1 must be 16-byte aligned and x1 a buffer of its own");
  XaDev p{};
  p.ao = (const half_t*)ao;
  p.w1 = (const half_t*)w1;
  p.b1 = (const float*)b1;
  p.kpad_1 = kpad_1;
  p.x0 = (const half_t*)x0;
  p.x1 = (half_t*)x1;
  p.x = (const half_t*)x1;
  p.wq = (const half_t*)wq;
  p.bq = (const float*)bq;
  p.cs = (const float*)ln_colsum;
  p.ln_on = ln_colsum != nullptr;
  p.k = (const half_t*)k;
  p.vt = (const half_t*)vt;
  p.wo = (const half_t*)wo;
  p.bo = (const float*)bo;
  p.residual = (const half_t*)x1;
  p.out = (half_t*)out;
  p.zeros = (const half_t*)zeros;
  p.M = B * N;
  p.N = N;
  p.L = L;
  p.kpad_q = kpad_q;
  p.kpad_o = kpad_o;
  p.ldk = ldk;
  p.ldv = ldv;
  p.vbs = (long)vt_batch_stride;
  p.ln_eps = ln_eps;
  p.scale_log2e = scale * 1.4426950408889634f;
  static bool attr_t = false;
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_xattn320t_kernel<2, true>), XT_LDS, attr_t, "af_xattn_chain")) return af_check_launch("af_xattn_chain");
  AfLaunchScope scope(AF_FAM_XATTN, stream);
  hipLaunchKernelGGL((af_xattn320t_kernel<2, true>), dim3(p.M / XA_BM), dim3(512), XT_LDS, (hipStream_t)stream, p);
  return af_check_launch("af_xattn_chain");
}

// GroupNorm(32) of a single-source tensor whose partial statistics exist (af_gemm_desc.gn_partials) + the 1x1 convolution / Linear behind it at
// C = 320, one launch (af_gn_proj320_kernel above): out [M][320] = GN(x) W^T + bias.  HW = rows per batch item (a multiple of 128), nblk = HW / 128.
extern "C" int af_gn_proj_fused(const void* x, const void* partials, int nblk, const void* gamma, const void* beta, float eps, const void* w, const void* bias,
                                int kpad, void* out, int B, int HW, int C, int groups, const void* zeros, void* stream) {
  AF_REQUIRE(x && partials && gamma && beta && w && out && zeros, "af_gn_proj_fused: null pointer");
  AF_SUPPORTED(C == XA_C && groups == 32, "af_gn_proj_fused: built for C = 320 in 32 groups (the 64 x 64 level of SD-1.5)");
  AF_REQUIRE(B > 0 && HW > 0 && HW % XA_BM == 0 && nblk == HW / XA_BM && nblk <= 128, "af_gn_proj_fused: HW must be a multiple of 128 (at most 128 blocks) and nblk = HW / 128");
  AF_REQUIRE(kpad >= C && kpad % 64 == 0, "af_gn_proj_fused: the weight row stride must be a 64-multiple covering C");
  AF_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w) & 15) == 0, "af_gn_proj_fused: x / out / gamma / beta / w must be 16-byte aligned");
  GpDev p;
  p.x = (const half_t*)x;
  p.partials = (const float*)partials;
  p.gamma = (const float*)gamma;
  p.beta = (const float*)beta;
  p.w = (const half_t*)w;
  p.bias = (const float*)bias;
  p.out = (half_t*)out;
  p.zeros = (const half_t*)zeros;
  p.M = B * HW;
  p.HW = HW;
  p.nblk = nblk;
  p.kpad = kpad;
  p.eps = eps;
  static bool attr_set = false;
  if (!af_allow_dyn_lds(reinterpret_cast<const void*>(&af_gn_proj320_kernel), XT_LDS, attr_set, "af_gn_proj_fused")) return af_check_launch("af_gn_proj_fused");
  AfLaunchScope scope(AF_FAM_GEMM, stream);
  hipLaunchKernelGGL(af_gn_proj320_kernel, dim3(p.M / XA_BM), dim3(512), XT_LDS, (hipStream_t)stream, p);
  return af_check_launch("af_gn_proj_fused");
}
