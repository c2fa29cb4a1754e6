/*
 * adaface_hip.h -- C ABI of libadaface_hip.so: the MI355X (gfx950) kernels of AdaFace's
 * denoising hot path (SD-1.5 U-Net epsilon-prediction).
 *
 * The reference (askerlee/AdaFace-dev) has no FFI layer: its seams are Python call
 * signatures that bottom out in torch ops.  Each entry point below names the reference
 * call site(s) (file:line relative to the reference tree) whose arithmetic it replaces;
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the library
 *     allocates nothing and keeps no global mutable state (re-entrant: forward thread and
 *     autograd thread may call concurrently);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     no call synchronises, so every call is hipGraph-capturable;
 *   - activations are fp16 ("h"), NHWC / token-major [rows, channels]; accumulation,
 *     statistics and softmax are fp32;
 *   - return value: 0 = ok, <0 = AF_E_* ; af_last_error() gives the message (thread-local).
 */
#ifndef ADAFACE_HIP_H
#define ADAFACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AF_OK 0
#define AF_E_BADARG (-1)      /* null pointer, negative size, misaligned dimension   */
#define AF_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for        */
#define AF_E_HIP (-3)         /* launch failed; message holds hipGetErrorString      */

const char* af_last_error(void);
int af_version(void);
/* number of HIP devices visible, or AF_E_HIP (used by smoke tests; does not create a context) */
int af_device_count(void);

/* ---- profiling hook (bench.py roofline leg): when enabled, every launch of the kernel
 * family `family` is bracketed by hipEvents on its own stream; af_prof_read synchronises
 * those events and returns the launch count and summed milliseconds since the last reset. */
#define AF_FAM_GEMM 0   /* conv3x3 implicit-GEMM + linear/conv1x1 (one kernel template) */
#define AF_FAM_ATTN 1
#define AF_FAM_GNORM 2
#define AF_FAM_LNORM 3
#define AF_FAM_ELEM 4
#define AF_FAM_XATTN 5  /* af_attention launches with fewer keys than queries: the U-Net's cross-attention cores */
#define AF_FAM_COUNT 6
int af_prof_enable(int on);
int af_prof_reset(void);
int af_prof_read(int family, int* launches, double* total_ms);

/* ---- GEMM / implicit-GEMM convolution -------------------------------------------------
 * out[M,N] = epilogue( sum_k A[m,k] * Wt[n,k] )          (Wt is [Npad, Kpad], K contiguous)
 * Replaces: nn.Conv2d 3x3 / 1x1 (ldm/modules/diffusionmodules/util.py:214-224, called at
 * openaimodel.py:202-242,108,152,523,689; attention.py:269,280) and nn.Linear
 * (attention.py:34,54,156-162; openaimodel.py:511-515,221).
 *
 * A operand (activations, fp16):
 *   taps == 1 : plain rows.  k <  k1 reads a1[m*lda1 + k], k >= k1 reads a2[m*lda2 + (k-k1)]
 *               (a2 may be NULL when k1 == K): fuses torch.cat([h, skip], 1) (openaimodel.py:918).
 *   taps == 9 : 3x3, pad 1, implicit im2col over an NHWC image [B, H, W, c1 (+c2)]:
 *               k = tap*(c1+c2) + c ; stride 1|2 (Downsample, openaimodel.py:150-161);
 *               upsample=1 reads the nearest-x2 upsampled image without materialising it
 *               (Upsample, openaimodel.py:117-119); upsample=2 reads the ZERO-INSERTED x2 image of size
 *               Ho x Wo (data at even positions): with flipped/transposed weights this is the input
 *               gradient of the stride-2 convolution.  M = B*Ho*Wo.
 * Epilogue (fp32): + bias[n] ; + rowbias[(m / rows_per_batch)*ld_rowbias + n] (ResBlock time
 *   embedding add, openaimodel.py:265-274) ; act ; + residual[m*N + n] ; -> fp16.
 *   act: AF_ACT_NONE | AF_ACT_SILU | AF_ACT_GEGLU (Wt rows interleaved [16 x | 16 gate],
 *   out has N/2 columns: a * gelu_erf(g), attention.py:36-38).
 *   out_mode AF_OUT_SPLIT_T: columns n >= split_col are written TRANSPOSED to out2
 *   ([B, N - split_col, ld_out2] with token index contiguous): V^T for the attention kernel.
 *   out_mode AF_OUT_F32: out is fp32 [M][ld_out] -- the accumulator is stored without the fp16 cast (weight gradients: sums over
 *   thousands of tokens overflow fp16 long before they lose precision).  Standard epilogue only, N % 4 == 0, no splitk_fused.
 *   With split-K the reduce pass writes fp32; unsplit, tiles 1 / 2 do (tiles 3 .. 10 fall back to tile 1).
 */
#define AF_ACT_NONE 0
#define AF_ACT_SILU 1
#define AF_ACT_GEGLU 2
#define AF_ACT_QUICKGELU 3 /* x * sigmoid(1.702 x): CLIP text MLP (transformers QuickGELUActivation) */
#define AF_OUT_NORMAL 0
#define AF_OUT_SPLIT_T 1
#define AF_OUT_F32 2

typedef struct af_gemm_desc {
  const void* a1;       /* fp16 */
  const void* a2;       /* fp16 or NULL */
  const void* wt;       /* fp16 [Npad][Kpad], Npad % 128 == 0, Kpad % 64 == 0, zero padded */
  const void* bias;     /* fp32 [N] or NULL */
  const void* rowbias;  /* fp16 [B][ld_rowbias] or NULL */
  const void* residual; /* fp16 [M][N] or NULL */
  void* out;            /* fp16 [M][ld_out] (fp32 with AF_OUT_F32) */
  void* out2;           /* fp16, AF_OUT_SPLIT_T only */
  int32_t M, N, K;      /* logical sizes (K = taps*(c1+c2) for taps == 9) */
  int32_t kpad;         /* row stride of wt in elements */
  int32_t taps;         /* 1 or 9 */
  int32_t c1, c2;       /* channels of a1 / a2 (taps==1: k1 = c1, k2 = c2) */
  int32_t lda1, lda2;   /* taps==1 row strides in elements */
  int32_t B, H, W;      /* taps==9 input image (before upsample) */
  int32_t Ho, Wo;       /* taps==9 output image */
  int32_t stride;       /* 1 | 2 */
  int32_t upsample;     /* 0 | 1 (nearest x2) | 2 (zero-insert x2) */
  int32_t rows_per_batch;
  int32_t ld_rowbias;
  int32_t act;
  int32_t out_mode;
  int32_t ld_out;       /* elements; 0 -> N (or N/2 for GEGLU) */
  int32_t split_col;    /* AF_OUT_SPLIT_T */
  int32_t ld_out2;      /* AF_OUT_SPLIT_T: tokens per batch item rounded up to 8 */
  int32_t tile;         /* 0 = auto (1 or 2), 1 = 128x128, 2 = 64x64 (register-staged), 3 = 128x128 and 4 = 128x320 LDS-DMA
                           pipelined ring (standard epilogue, channel counts % 32 == 0, no upsample, tile 4: N % 320
                           == 0; otherwise falls back to 1) */
                        /* 5 = 256x256 and 6 = 256x320 ring tiles (8 waves as 4 x 2): plain or GEGLU 1x1 GEMMs only, N % 256 / N % 320 == 0,
                           no split-K */
                        /* 7 .. 10 = whole-line kernel (64-wide K stages, LDS-DMA pieces of 8 rows x one 128-byte line, two slots; channel
                           counts and K padding multiples of 64): 7 = 128x320 (GEGLU 128x256, transposed-V split), 8 = 128x128 (4 waves; also GEGLU,
                           transposed-V split), 9 / 10 = GEGLU 256x320 / 256x256; tiles 7 and 8 also take upsample = 1 (nearest x2).  Anything outside a tile's scope falls back to tile 1 */
                        /* 11 = whole-line kernel, 128x160 tile of four waves (N % 160 == 0; standard epilogue, taps 1 / 9, nearest x2): 74 KB of
                           LDS, so TWO workgroups share a CU and one's epilogue / barrier waits run under the other's MFMAs: the faster
                           form when the 128x320 grid has under ~1.5 workgroups per CU (batch 1 - 4 passes, the 32x32 level) */
                        /* 12 / 13 = the 128x128 / 128x160 whole-line tiles with a FOUR-slot ring: three K stages in flight ahead of the MFMAs
                           instead of one (128 / 147 KB of LDS, one workgroup per CU): for grids of at most ~one workgroup per CU, where
                           nothing else covers a stage's L2 / HBM latency (the 16x16 level: weights read once, 160 tiles) */
                        /* 14 = halo-resident 3x3 kernel: 256 output pixels (whole image rows) x 160 channels per workgroup; per 64 input
                           channels the rows' (R+2) x (W+2) halo is loaded into LDS once and the nine taps read it at shifted addresses
                           (stride 1, pad 1, c1 % 64 == 0, c2 % 64 == 0 -- one or two channel-concatenated sources --, no K tail, N % 160 == 0,
                           Wo in {8, 16, 32, 64}, a tile = 256 / Wo whole rows of one image or whole images (the 8 x 8 level), upsample 0 / 1; split-K over 64-channel chunks of c1 + c2).  Main loop: ping-pong between the two
                           waves of a SIMD (one group issues a stage's 40 MFMAs while the other reads its fragments and issues its LDS-DMA
                           pieces; 160 KB of LDS).  Round 6: every per-stage address (fragment reads per tap, LDS-DMA pieces) is a register or a scalar +
                           immediate computed outside the loop, out-of-image halo lanes are EXEC-masked instead of reading a zero page (~10 instead of
                           ~80 vector instructions per 40-MFMA stage).  Also round 6, for the VAE's channel counts and image sizes (model.py:136-175, 536-567): where N is
                           a 128-multiple but no 160-multiple the tile is 256 x 128 (no K tail), and on images wider than 64 pixels (Wo % 16 == 0, Ho % 16 == 0; never
                           split) a tile is a 16 x 16-pixel PATCH of one image (18 x 18 halo) instead of whole rows.  af_gemm_halo_variant tells which form a
                           descriptor gets.  Outside that scope it falls back to tile 1 */
                        /* 19 = tile 14 with its round-5 main loop (addresses recomputed per stage): the A/B arm; bit-identical results */
                        /* 15 = whole-line kernel, 256 x 128 tile of eight waves (4 x 2; N % 128 == 0; standard epilogue, taps 1 / 9, nearest x2): 85 instead of
                           64 FLOP per operand byte for narrow outputs over many rows (the VAE decoder's 128- / 256-channel convolutions at 256^2 / 512^2) */
                        /* 16 / 17 = whole-line kernel, 64 x 128 / 128 x 64 tile of four waves (1 x 4 / 2 x 2; N % 128 / 64 == 0; standard and transposed-V-split epilogues, folded
                           LayerNorm, taps 1 / 9, no nearest x2, no K tail): 25 KB stages, so THREE workgroups share a CU -- a short-K GEMM's K step is a
                           memory round trip, and what hides it is the other workgroups' MFMAs, not a deeper ring */
                        /* 18 = SMALL plain GEMMs with their MFMA fragments straight from global memory (both operands are K-contiguous: a fragment is one 16-byte
                           load per lane), 32 x 64 outputs per workgroup, no LDS, no barrier, several K steps of fragments in flight, never split: the training
                           legs' GEMMs of a few hundred rows, which on tile 2 are chains of K / 64 dependent stages (+ a split-K reduce launch).  taps 1, one
                           source, standard epilogue, fp16 or fp32 output, K % 8 == 0, 16-byte aligned rows; outside that scope it falls back to tile 2 */
  int32_t splits;       /* split-K factor (<=1: none).  >1 needs the standard epilogue and a workspace:
                           each split writes an fp32 partial [M][N], a second launch reduces + applies the epilogue */
  void* workspace;      /* fp32, >= splits*M*N*4 bytes when splits > 1 */
  int64_t workspace_bytes;
  const void* zeros;    /* >= 16 bytes of zeros, 16-byte aligned: source of halo / out-of-range lanes (tile 3) */
  int32_t tap_shift;    /* 3x3 only: 0 = padding 1 on every side; 1 = taps shifted by +1 pixel, i.e. padding (0, 1, 0, 1) as the VAE
                           encoder's Downsample pads before its stride-2 conv (ldm/modules/diffusionmodules/model.py:73-77) */
  int32_t splitk_fused; /* split-K only: 1 = reduce inside the GEMM launch (tiles 3 .. 10, at most AF_SPLITK_MAX_TILES output tiles; otherwise the
                           two-launch form is used).  The LAST AF_SPLITK_COUNTER_BYTES of the workspace are then per-tile arrival counters:
                           the caller zeroes them once when it allocates the workspace, every launch leaves them zero.  The partial slabs
                           must fit in the first workspace_bytes - AF_SPLITK_COUNTER_BYTES bytes.  Results are bit-identical to the
                           two-launch form (slices are summed in slice order by the last-arriving workgroup of each tile) */
  const void* ln_colsum; /* fp32 [Npad] or NULL.  Non-NULL = LayerNorm folded into this GEMM (BasicTransformerBlock.norm1/2/3 in front of
                           attn1 q|k|v, attn2.to_q and the GEGLU projection, attention.py:242-252): a1 holds the UN-normalised rows, wt holds
                           fp16(gamma * W), bias holds b + W beta, ln_colsum[n] = sum_k wt[n][k] (of the fp16 values), and the epilogue
                           computes rstd_m * (acc - mean_m * ln_colsum[n]) + bias[n] with the row statistics (mean, biased variance over the
                           K = c1 columns, eps = ln_eps) accumulated from the A fragments inside the main loop.  Whole-line tiles (7 .. 13)
                           only, taps == 1, c2 == 0, no split-K; anything else is AF_E_UNSUPPORTED */
  float ln_eps;
  /* K-concatenated 1x1 tail of a 3x3 convolution (taps == 9 only; all NULL / 0 otherwise): behind the nine tap blocks K continues with c3 (+ c4)
   * PLAIN columns read from a3 [M][lda3] (| a4 [M][lda4]) at the OUTPUT pixel's row, K = 9 (c1 + c2) + c3 + c4 -- out = conv3x3(a1 | a2) + conv1x1(a3 | a4)
   * in one launch: the ResBlock's out_layers convolution and its channel-changing skip_connection (openaimodel.py:256-276; wt = [3x3 weights in
   * (ky, kx, cin) order | 1x1 weights], bias = the two biases added up).  stride 1, no upsample, no tap_shift, c3 / c4 multiples of 64, whole-line
   * tiles 7 .. 13 only (AF_E_UNSUPPORTED otherwise: the caller keeps the two-launch form for such shapes). */
  const void* a3;
  const void* a4;
  int32_t c3, c4, lda3, lda4;
  /* GroupNorm statistics of the OUTPUT from the producing GEMM's epilogue (all 0 / NULL otherwise): gn_partials != NULL makes the launch also write,
   * per (batch item b, 128-row tile blk of that item, group g of gn_cpg output channels), the partial (sum, M2) of the fp16 values it stores -- M2 = the sum of
   * squares about THAT BLOCK'S OWN MEAN; the consumer merges blocks pairwise (M2_a + M2_b + (mean_b - mean_a)^2 n_a n_b / (n_a + n_b)), so no
   * E[x^2] - mean^2 difference is ever formed and groups with |mean| >> sigma keep their variance (torch's fp32 group_norm, util.py:195-212, is the
   * reference) -- as fp32 [B][128][32][2]: exactly the partial workspace of af_groupnorm, so af_groupnorm_apply can normalise the tensor without a
   * statistics pass of its own (GroupNorm32 behind a convolution: openaimodel.py:202-233, 256-276; attention.py:283-291).  Needs the standard
   * epilogue on a whole-line tile whose width is a multiple of gn_cpg (tile 7: 128 x 320, tile 11 / 13: 128 x 160), gn_cpg even, N % gn_cpg == 0,
   * N / gn_cpg <= 32, rows_per_batch % 128 == 0 (or M % 128 == 0 when rows_per_batch is 0), rows_per_batch / 128 <= 128, fp16 output, no split-K;
   * anything else is AF_E_UNSUPPORTED (callers check af_gemm_gn_stats_ok first). */
  void* gn_partials;
  int32_t gn_cpg;
  /* Split-K reduction left to the CONSUMER (round 6; NULL otherwise).  A split launch normally ends with a chip-wide reduce pass (fp32 slabs ->
   * bias / row bias / activation / residual -> fp16).  Where the output's first reader is a GroupNorm (every ResBlock convolution:
   * openaimodel.py:256-276, util.py:195-212) that pass and the GroupNorm's own read of its result are one launch too many: with defer_reduce != NULL
   * a launch that WOULD run the separate reduce pass skips it, leaves its `*defer_reduce` = splits slabs [splits][M][N] fp32 at the start of
   * `workspace`, and writes nothing to `out`; af_groupnorm_splitk (or af_splitk_reduce) then finishes the tensor.  *defer_reduce = 0 when the launch
   * stored `out` itself (unsplit, or reduced in-launch).  Standard epilogue without activation only (AF_E_BADARG otherwise). */
  int32_t* defer_reduce;
} af_gemm_desc;
#define AF_SPLITK_MAX_TILES 4096
#define AF_SPLITK_COUNTER_BYTES (AF_SPLITK_MAX_TILES * 4)

int af_gemm(const af_gemm_desc* d, void* stream);

/* The reduce pass of a split-K launch on its own (what af_gemm runs after a split launch unless af_gemm_desc.defer_reduce held it back):
 * out[m][n] = fp16( sum_sp slabs[sp][m][n] + bias[n] + rowbias[m / rows_per_batch][n] + residual[m][n] ), summed in slice order.  bias fp32 [N],
 * rowbias fp16 rows of ld_rowbias, residual fp16 [M][N] -- each may be NULL.  N % 4 == 0. */
int af_splitk_reduce(const void* slabs, int splits, const void* bias, const void* rowbias, int ld_rowbias, int rows_per_batch, const void* residual,
                     void* out, int M, int N, void* stream);

/* ---- fused feed-forward of a transformer block at C = 320 ------------------------------------------------
 * Replaces, in ONE launch, BasicTransformerBlock's  x + ff(norm3(x))  (attention.py:31-58 FeedForward / GEGLU, :242-252) at the
 * 64 x 64 level of SD-1.5:  out = residual + b2 + W2 (v * gelu_erf(g)),  [v | g] = W1 LN(x) + b1, without the [M, inner]
 * intermediate ever leaving the compute unit.  x / residual / out: fp16 [M, C] (C = 320); w1: fp16 packed GEGLU-interleaved
 * [2 * inner][kpad1] with the LayerNorm's gamma folded in, b1 fp32 [2 * inner] (interleaved, + W1 beta), ln_colsum fp32 [2 * inner]
 * column sums of the packed rows (NULL: no LayerNorm in front); w2: fp16 packed [>= C rows][kpad2], b2 fp32 [C] (may be NULL);
 * zeros as af_gemm_desc.zeros.  AF_E_UNSUPPORTED for any other C.                                                              */
int af_ff_fused(const void* x, const void* w1, const void* b1, const void* ln_colsum, float ln_eps, int kpad1, const void* w2, const void* b2,
                int kpad2, const void* residual, void* out, int M, int C, int inner, const void* zeros, void* stream);
/* The same feed-forward with the SpatialTransformer's proj_out + residual behind it (round 6; attention.py:287-304: x = proj_out(blocks(...)) + x_in), one launch:
 *     x3  = residual + b2 + W2 (v * gelu_erf(g))        (af_ff_fused's result: here it never leaves the compute unit)
 *     out = x_in + bp + Wp x3
 * wp: fp16 packed [>= C rows][kpad_p], bp fp32 [C] (may be NULL), x_in fp16 [M, C].  gn_partials (may be NULL): the GroupNorm partial statistics of `out` as
 * af_gemm_desc.gn_partials leaves them, [M / rows_per_batch][128][32][2] floats, groups of gn_cpg channels, rows_per_batch a multiple of 128 (at most 128 blocks).
 * C = 320 only (AF_E_UNSUPPORTED otherwise); out / residual / x_in / b2 / bp 16-byte aligned. */
int af_ff_chain(const void* x, const void* w1, const void* b1, const void* ln_colsum, float ln_eps, int kpad1, const void* w2, const void* b2, int kpad2,
                const void* residual, const void* wp, const void* bp, int kpad_p, const void* x_in, void* out, void* gn_partials, int gn_cpg,
                int rows_per_batch, int M, int C, int inner, const void* zeros, void* stream);

/* Whole cross-attention block of a transformer layer at C = 320, 8 heads (the 64 x 64 level of SD-1.5) in ONE launch (replaces
 * attention.py:168-222 + the norm2 of :242-252): out = residual + bo + Wo . concat_h(softmax(q_h K_h^T scale) V_h), q = LN(x) Wq^T.
 * wq / bq / ln_colsum: the q projection packed with the LayerNorm folded in (as af_gemm_desc.ln_colsum takes it; ln_colsum NULL = x is
 * used as it is, bq NULL = no shift).  k [B * L][ldk] and vt [B][C][ldv] (batch stride vt_batch_stride; the row pad, keys L .. ldv-1, may hold ANYTHING: masked keys contribute nothing) are this
 * layer's slices of the context projection (head h at columns / rows 40 h ..).  N (tokens per image) % 128 == 0, L <= 80.
 * Round 6: C = 640 (8 heads of 80, the 32 x 32 level; N % 64 == 0) is taken too -- 64-token workgroups that stream both 640 x 640 weights; measured slower than
 * three launches there, so the module keeps it behind AF_FUSE_XATTN640.  Any other C: AF_E_UNSUPPORTED. */
int af_xattn_fused(const void* x, const void* wq, const void* bq, const void* ln_colsum, float ln_eps, int kpad_q, const void* k, int ldk,
                   const void* vt, int64_t vt_batch_stride, int ldv, const void* wo, const void* bo, int kpad_o, const void* residual,
                   void* out, int B, int N, int L, int C, int heads, float scale, const void* zeros, void* stream);

/* The same block with the SELF-attention's output projection chained in front (round 6; attention.py:242-252: x = attn1(norm1(x)) + x; x = attn2(norm2(x),
 * context) + x): x1 = ao W1^T + b1 + x0 is formed by the launch itself -- ao [B * N][C] the self-attention core's output, w1 / b1 attn1.to_out packed as for
 * af_gemm, x0 the transformer block's input -- written to the caller's scratch x1 [B * N][C] (a buffer of its own) and used as the cross-attention's
 * input AND residual; out = x1 + bo + Wo . attention(LN(x1) Wq^T, K, V).  One `[B * N, 320] x [320, 320]` launch and three passes over that tensor
 * less per transformer block.  The other arguments as af_xattn_fused. */
int af_xattn_chain(const void* ao, const void* w1, const void* b1, int kpad_1, const void* x0, void* x1, const void* wq, const void* bq,
                   const void* ln_colsum, float ln_eps, int kpad_q, const void* k, int ldk, const void* vt, int64_t vt_batch_stride, int ldv,
                   const void* wo, const void* bo, int kpad_o, void* out, int B, int N, int L, int C, int heads, float scale, const void* zeros,
                   void* stream);

/* PADDED LEADING DIMENSIONS -- the contract for every entry point that takes an ld larger than the logical extent (k / q rows wider than heads*d,
 * vt rows of ldv >= L keys, keybias rows of ldb >= L, out2 rows of ld_out2 >= rows_per_batch): a consumer NEVER lets bytes outside the logical
 * extent reach a result -- they may be uninitialised memory (NaN, Inf) -- and a producer that owns a padded output row (af_gemm's AF_OUT_SPLIT_T
 * out2, af_transpose_tokens) writes zeros into the pad.  Packed weights (wt) are different: the caller zero-pads them to [npad][kpad] when packing,
 * and the kernels rely on that.  tests/conftest.py runs every GPU test with NaN-filled torch.empty buffers to hold this.                        */

/* GroupNorm(32) of a single-source tensor whose partial statistics already exist (af_gemm_desc.gn_partials) + the 1x1 convolution behind it at C = 320
 * in ONE launch: out [B*HW][320] = GroupNorm(x) W^T + bias -- the SpatialTransformer's proj_in(norm(x)) (attention.py:283-291); the normalised
 * tensor never exists in memory.  partials fp32 [B][128][32][2], nblk = HW / 128 valid blocks per batch item; w packed as for af_gemm
 * ([>= 320][kpad]).  HW % 128 == 0.  AF_E_UNSUPPORTED for any other C / group count. */
int af_gn_proj_fused(const void* x, const void* partials, int nblk, const void* gamma, const void* beta, float eps, const void* w, const void* bias,
                     int kpad, void* out, int B, int HW, int C, int groups, const void* zeros, void* stream);

/* ---- GroupNorm(32) [+ SiLU], NHWC -----------------------------------------------------
 * Replaces GroupNorm32 + nn.SiLU (util.py:195-212; openaimodel.py:202-233,686-690; eps 1e-5)
 * and Normalize (attention.py:70-71; eps 1e-6).  x = concat(x1[.., c1], x2[.., c2]) along
 * channels (x2 may be NULL).  Two launches: partial sums -> normalise.
 * workspace: fp32, at least af_groupnorm_ws_floats(B) floats.  x1 / x2 / y / gamma / beta must
 * be 16-byte aligned (vector accesses); AF_E_BADARG otherwise.                              */
int af_groupnorm_ws_floats(int B);
/* GroupNorm(32) [+ SiLU] of a single-source tensor whose partial statistics already exist (af_gemm_desc.gn_partials of the launch that produced x):
 * partials fp32 [B][128][32][2] with nblk valid blocks per batch item; one launch (the normalising pass of af_groupnorm).  stats (optional) as
 * af_groupnorm_stats. */
int af_groupnorm_apply(const void* x, int C, const void* gamma, const void* beta, void* y, void* stats, int B, int HW, int groups, float eps,
                       int silu, const void* partials, int nblk, void* stream);
/* GroupNorm(32) [+ SiLU] fed by the fp32 slabs of a split-K launch (af_gemm_desc.defer_reduce) in ONE launch: the value at (row, c) is
 * fp16( sum_sp slabs[sp][row][c] + bias[c] + rowbias[b][c] + residual[row][c] ) -- the same arithmetic in the same order as af_splitk_reduce, so the
 * stored tensor x_out is bit-identical to the two-launch form -- which is written to x_out (the convolution's output: later readers need it) AND
 * normalised into y with the one-launch in-register forms of af_groupnorm (a workgroup per (batch item, group)); stats as af_groupnorm_stats.
 * Replaces af_splitk_reduce_kernel + gn_small / gn_pair behind every split ResBlock convolution (openaimodel.py:256-276).  Scope =
 * af_groupnorm_splitk_ok(B, HW, C, groups) == 1 (the in-register forms: HW * C / groups small enough); AF_E_UNSUPPORTED otherwise. */
int af_groupnorm_splitk_ok(int B, int HW, int C, int groups);
int af_groupnorm_splitk(const void* slabs, int splits, const void* bias, const void* rowbias, int ld_rowbias, const void* residual, void* x_out, int C,
                        const void* gamma, const void* beta, void* y, void* stats, int B, int HW, int groups, float eps, int silu, void* stream);
/* 1 when af_gemm with this (tile, splits) takes gn_partials for an output of N channels in groups of cpg and rows_per_batch rows per batch item */
int af_gemm_gn_stats_ok(int tile, int splits, int taps, int act, int out_mode, int N, int cpg, int rows_per_batch);
/* Which form of the halo-resident 3x3 kernel (tile 14) this descriptor would run: 0 = outside its scope (af_gemm falls back to a tap-by-tap tile),
 * 1 = 256 x 160 tiles on whole image rows, 2 = 256 x 128 tiles on whole image rows, 3 = 256 x 128 tiles on 16 x 16-pixel patches.  Reads the geometry
 * fields only (taps, c1 .. c4, N, B, H, W, Ho, Wo, stride, upsample, tap_shift, act, out_mode, kpad, M); no launch. */
int af_gemm_halo_variant(const af_gemm_desc* d);
int af_groupnorm(const void* x1, const void* x2, int c1, int c2, const void* gamma, const void* beta,
                 void* y, int B, int HW, int groups, float eps, int silu, void* workspace, void* stream);

/* ---- LayerNorm over the last dim (nn.LayerNorm, attention.py:232-234; eps 1e-5) ------- */
/* same, and also writes stats fp32 [B, groups, 2] = (mean, rstd) for af_groupnorm_bwd */
int af_groupnorm_stats(const void* x1, const void* x2, int c1, int c2, const void* gamma, const void* beta,
                       void* y, void* stats, int B, int HW, int groups, float eps, int silu, void* workspace,
                       void* stream);

int af_layernorm(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                 void* stream);

/* ---- fused attention core: softmax(q k^T * scale + keybias) v --------------------------
 * Replaces attention.py:180-204 (CrossAttention.forward between the projections) and the
 * explicit SDPA of adaface/diffusers_attn_lora_capture.py:79-139 without materialising
 * the [b*h, N, L] score tensor.
 *   q  [B, Nq, ldq]   k [B, L, ldk]   o [B, Nq, ldo]   (row strides in elements, >= heads*d:
 *   q and k may be column slices of one fused projection output)
 *   vt [B, heads*d, ldv] (V transposed, key index contiguous, ldv % 8 == 0)
 *   keybias: fp32 [B, ldb] added to every score row (0 = keep, -FLT_MAX = masked key as
 *   masked_fill_(~mask, -finfo.max) attention.py:188-194); NULL = none.  Keys >= L are
 *   always excluded.  d % 8 == 0, d <= 160.                                                */
int af_attention(const void* q, const void* k, const void* vt, void* o, const void* keybias, int B, int Nq,
                 int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb, float scale, void* stream);
/* same, and also writes lse2 fp32 [B, heads, Nq] = log2(sum_j exp(score_ij)) (base-2 log-sum-exp of the
 * scaled+biased scores), which af_attention_bwd consumes. */
int af_attention_lse(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse, const void* keybias,
                     int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb, float scale,
                     void* stream);

/* general form: causal_m > 0 adds CLIP's causal mask with `causal_m` keys per token (key j visible to query i
 * iff j / causal_m <= i): adaface/arc2face_models.py:145-231 (CLIPAttentionMKV, K/V widened x m). */
int af_attention_ex(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse, const void* keybias,
                    int causal_m, int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb,
                    float scale, void* stream);
/* same with an explicit element stride between the batch items of vt (>= heads*d*ldv): lets vt be a row slice of a larger
 * [B, sum_C, ldv] tensor -- the U-Net computes the K / V^T projections of ALL its cross-attention layers in one GEMM */
int af_attention_strided(const void* q, const void* k, const void* vt, void* o, void* lse2, int ld_lse, const void* keybias, int causal_m,
                         int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldo, int ldv, int ldb, int64_t vt_batch_stride,
                         float scale, void* stream);

/* ---- attention backward (flash-style, recomputes P from q, k and lse2) -------------------
 * Input gradients of af_attention for dO = `dout`: dq [B,Nq,lddq], dk [B,L,lddk], dv [B,L,lddv].
 * q/k/v/o/dout are ROW-major token tensors (v is NOT transposed here) with row strides ld*;
 * lse2 fp32 [B, heads, ld_lse] from af_attention_lse, ld_lse >= roundup(Nq, 32), % 4 == 0.
 * scratch: >= af_attention_bwd_scratch_bytes(...) bytes (Q^T, K^T, dO^T copies and delta).
 * Reference: autograd of attention.py:180-204 (the reference stores the full score tensor).   */
int64_t af_attention_bwd_scratch_bytes(int B, int Nq, int L, int heads, int d);
int af_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const void* lse2,
                     int ld_lse, const void* keybias, int causal_m, void* dq, void* dk, void* dv, void* scratch,
                     int64_t scratch_bytes, int B, int Nq, int L, int heads, int d, int ldq, int ldk, int ldv,
                     int ldo, int lddo, int lddq, int lddk, int lddv, int ldb, float scale, void* stream);

/* explicit score / probability capture for one cross-attention layer (attention.py:207-220):
 * score[b,h,i,j] = q.k*scale, prob = softmax_j(score); fp32 outputs [B,heads,Nq,L].         */
int af_attention_scores(const void* q, const void* k, void* score, void* prob, int B, int Nq, int L, int heads,
                        int d, float scale, void* stream);

/* ---- explicit cross-attention of the capture / score-rewrite path (adaface/diffusers_attn_lora_capture.py:79-139, 309-315) ----
 * The captured cross-attention layers (22-24) materialise their scores so that they can be rewritten between the product and the
 * softmax (SC/MC mixing, subject-token normalisation) and captured WITH gradients.  q [B*Nq, ldq], k / v [B*L, ldk / ldv] fp16
 * row-major (head h = columns h*d ..), score / prob / dscore fp32 [B, heads, Nq, L] contiguous, L <= 128, d % 8 == 0.
 *   af_xattn_scores          score = scale * q k^T
 *   af_xattn_softmax_pv      prob = softmax_L(score);  o[B*Nq, ldo] = prob v                        (fp16)
 *   af_xattn_softmax_pv_bwd  dscore = prob * (dP - sum_L prob dP),  dP = dout v^T (+ dprob_ext, the gradient arriving on a captured prob; may be NULL)
 *   af_xattn_rowmix          out[B*Nq, ldout] = alpha * w x      (w fp32 [B,heads,Nq,L], x fp16 [B*L, ldx]):   dq = scale * dscore k
 *   af_xattn_colmix          out[B*L, ldout]  = alpha * w^T x    (x fp16 [B*Nq, ldx]):   dv = prob^T dout,  dk = scale * dscore^T q;
 *                            per (batch item, head) a GEMM over the queries on v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: w is not rounded);
 *                            deterministic two-pass reduction over 16 query chunks through a caller-owned fp32 workspace          */
#define AF_XATTN_COLMIX_CHUNKS 16
int af_xattn_scores(const void* q, int ldq, const void* k, int ldk, void* score, int B, int Nq, int L, int heads, int d, float scale, void* stream);
int af_xattn_softmax_pv(const void* score, const void* v, int ldv, void* prob, void* o, int ldo, int B, int Nq, int L, int heads, int d, void* stream);
int af_xattn_softmax_pv_bwd(const void* prob, const void* v, int ldv, const void* dout, int lddo, const void* dprob_ext, void* dscore, int B, int Nq,
                            int L, int heads, int d, void* stream);
int af_xattn_rowmix(const void* w, const void* x, int ldx, void* out, int ldout, float alpha, int B, int Nq, int L, int heads, int d, void* stream);
int64_t af_xattn_colmix_ws_bytes(int B, int L, int heads, int d);
int af_xattn_colmix(const void* w, const void* x, int ldx, void* out, int ldout, float alpha, void* workspace, int64_t workspace_bytes, int B, int Nq,
                    int L, int heads, int d, void* stream);

/* ---- small element-wise kernels --------------------------------------------------------
 * timestep embedding [cos | sin] (util.py:154-174) -> fp16 [B, dim]                        */
int af_timestep_embedding(const void* timesteps_i64, void* out, int B, int dim, float max_period, void* stream);
/* NCHW fp32 -> NHWC fp16 with channel padding to cpad (zeros) and back (NHWC fp16 -> NCHW fp32) */
int af_nchw_f32_to_nhwc_f16(const void* x, void* y, int B, int C, int HW, int cpad, void* stream);
int af_nhwc_f16_to_nchw_f32(const void* x, void* y, int B, int C, int HW, int cstride, void* stream);
/* classifier-free guidance + DDIM update (ldm/models/diffusion/ddim.py:253-255, 279-301, sigma = 0):
 * eps = e_u + g (e_c - e_u); pred_x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);
 * x_prev = sqrt(a_prev) pred_x0 + sqrt(1-a_prev) eps.   eps2 = [e_c ; e_u] fp32 [2n];
 * e_u == NULL (n_uncond = 0) means no guidance.  x, x_prev, pred_x0: fp32 [n].              */
int af_cfg_ddim_step(const void* eps2, const void* x, void* x_prev, void* pred_x0, int64_t n, int has_uncond,
                     float guidance, float a_t, float a_prev, void* stream);
/* q_sample (ldm/models/diffusion/ddpm.py:395-398): x_t = sa[b] x0 + sb[b] noise, fp32, per-sample scalars */
int af_q_sample(const void* x0, const void* noise, const void* sa, const void* sb, void* xt, int B, int64_t per,
                void* stream);

/* y = x * sigmoid(x), fp16 (nn.SiLU on the time embedding, openaimodel.py:219-220) */
int af_silu_f16(const void* x, void* y, int64_t n, void* stream);

/* ---- backward-pass kernels (activation gradients only: U-Net base weights are frozen, ddpm.py:4131) ----
 * GroupNorm(+SiLU) input gradient; x = concat(x1, x2) as in the forward, stats from af_groupnorm_stats,
 * dy/add [B,HW,c1+c2] (add optional, summed into dx), dx written to dx1 [B,HW,c1] / dx2 [B,HW,c2].      */
int af_groupnorm_bwd(const void* x1, const void* x2, int c1, int c2, const void* gamma, const void* beta,
                     const void* stats, const void* dy, const void* add, void* dx1, void* dx2, int B, int HW,
                     int groups, int silu, void* workspace, void* stream);
/* LayerNorm parameter gradients (the trainable LayerNorms of the CLIP encoders, subj_basis_generator.py:841-853): dgamma[c] =
 * sum_r dy[r,c] * (x[r,c] - mean_r) * rstd_r, dbeta[c] = sum_r dy[r,c]; x, dy fp16 [rows, C], outputs fp32 [C]; row_stats: caller-owned
 * fp32 scratch [rows][2] (mean, rstd: written by the first of the two launches); rows <= 2048, C <= 1536 */
int af_layernorm_param_grads(const void* x, const void* dy, void* dgamma, void* dbeta, void* row_stats, int rows, int C, float eps,
                             void* stream);
/* LayerNorm input gradient (+ optional `add`) */
int af_layernorm_bwd(const void* x, const void* gamma, const void* dy, const void* add, void* dx, int rows, int C,
                     float eps, void* stream);
/* un-fused GEGLU on the interleaved pre-activation hp [M, 2*inner] (training keeps hp): forward and input gradient */
int af_geglu_fwd(const void* hp, void* out, int64_t M, int inner, void* stream);
int af_geglu_bwd(const void* hp, const void* dout, void* dhp, int64_t M, int inner, void* stream);
/* adjoint of nearest-x2 upsampling: y [B,H,W,C] = 2x2 block sums of x [B,2H,2W,C] */
int af_sumpool2x2(const void* x, void* y, int B, int H, int W, int C, void* stream);
int af_add_f16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* out = a + alpha * b: the decoder's skip-connection gradients joining the encoder's, scaled by res_hidden_states_gradscale
 * (adaface/diffusers_attn_lora_capture.py:23-42 ScaleGrad, :382-396) */
int af_axpy_f16(const void* a, const void* b, float alpha, void* out, int64_t n, void* stream);
/* x [B,N,ldx] (C columns) -> y [B,C,ldy] with the token index contiguous (zero padded to ldy) */
int af_transpose_tokens(const void* x, void* y, int B, int N, int C, int ldx, int ldy, void* stream);
/* two such transposes of the same [B, N] token grid in one launch (a weight gradient's operands dy^T and x^T): x1 [B*N, ldx1] -> y1 [B, C1, ldy],
 * x2 [B*N, ldx2] -> y2 [B, C2, ldy] */
int af_transpose_tokens_pair(const void* x1, void* y1, int C1, int ldx1, const void* x2, void* y2, int C2, int ldx2, int B, int N, int ldy,
                             void* stream);
/* cautious AdamW (ldm/c_adamw.py:65-123) over a flat fp32 buffer; seg_offsets int64 [nseg+1] delimit the
 * parameter tensors (the caution mask is renormalised per tensor); counts: uint32 [nseg] scratch.          */
int af_cadamw_step(void* p, const void* g, void* m, void* v, const void* seg_offsets, int nseg, void* counts, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, int correct_bias, void* stream);

/* ---- parameter-gradient helpers of the (trainable) CLIP text encoder -------------------------------
 * out fp32 [C] (+)= sum_rows a[r,c] * (b ? b[r,c] : 1): bias gradients (b = NULL) and LayerNorm gamma gradients
 * (a = dy, b = x_hat).  accumulate != 0 adds to `out`.                                                      */
int af_colsum(const void* a, const void* b, void* out, int rows, int C, int accumulate, void* stream);
/* same for tall inputs (rows >> C): `chunks` row ranges are folded by separate workgroups into partial (fp32 [chunks, C],
 * caller-owned) and then reduced in a fixed order -- deterministic, no atomics */
int af_colsum_tall(const void* a, const void* b, void* partial, void* out, int rows, int C, int chunks, int accumulate, void* stream);
/* quick-GELU x*sigmoid(1.702x) (CLIP MLP) and its input gradient, fp16 element-wise */
int af_quickgelu_fwd(const void* x, void* y, int64_t n, void* stream);
int af_quickgelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
/* y = a * s (fp32), used to unscale the loss-scaled gradient arena before the optimizer step */
int af_scale_f32(void* a, float s, int64_t n, void* stream);
/* a = clamp(a, lo, hi): gradient clipping by VALUE over the flat gradient arena (Lightning `gradient_clip_algorithm: 'value'`,
 * `gradient_clip_val: 0.01` in configs/stable-diffusion/v1-distill-arc2face-ada.yaml:150-152; ddpm.py:496-497) */
int af_clamp_f32(void* a, float lo, float hi, int64_t n, void* stream);

/* ---- VAE decoder (ldm/modules/diffusionmodules/model.py:151-243 AttnBlock): row softmax of an explicit fp16 score matrix
 * [rows, L], L % 8 == 0, L <= 4096 (single-head 512-dim attention runs as af_gemm -> af_softmax_rows -> af_gemm) */
int af_softmax_rows(const void* x, void* y, int64_t rows, int L, void* stream);
/* its backward, ds = p * (dp - rowsum(p * dp)), for the decode-with-grad path of the ArcFace alignment loss (ddpm.py:2511-2535
 * differentiates through decode_first_stage_with_grad, ddpm.py:899-908) */
int af_softmax_rows_bwd(const void* p, const void* dp, void* ds, int64_t rows, int L, void* stream);
/* masked VAE-encoder attention (model.py:191-209): p fp16 [N, N] (post-softmax) *= ((cls[i] & cls[j]) != 0); cls uint8 [N]: bit 0 fg*aug != 0, bit 1 (1-fg)*aug != 0 */
int af_mask_pairs(void* p, const void* cls, int N, void* stream);
/* read every 128-byte line of [ptr, ptr + bytes) once (no writes): cache warm-up of packed weights ahead of the GEMM that streams them,
   meant for a side stream */
int af_prefetch(const void* ptr, int64_t bytes, void* stream);
/* the same with the grid capped at max_workgroups (256 threads each, striding over the lines): a prefetcher that runs BESIDE the GEMMs of a
   step on a second stream must not take the chip from them (ops.WeightPrefetcher) */
int af_prefetch_ex(const void* ptr, int64_t bytes, int max_workgroups, void* stream);

/* ---- ArcFace ResNetFace-18 IR-SE face encoder (reference evaluation/arcface_resnet.py:62-97, 139-154, 157-217) ----
 * NHWC fp16 activations, C % 8 == 0.  Convolutions / FCs are af_gemm calls with eval-mode BatchNorm folded on the host.
 * y = prelu(x * scale[c] + shift[c]); scale/shift fp32 [C] or both NULL; slope fp32 [1] device pointer or NULL (no PReLU) */
int af_affine_prelu(const void* x, const void* scale, const void* shift, const void* slope, void* y, int64_t rows, int C, void* stream);
/* x [B, 2Ho, 2Wo, C] -> y [B, Ho, Wo, C] (nn.MaxPool2d(2, 2), arcface_resnet.py:165) */
int af_maxpool2x2(const void* x, void* y, int B, int Ho, int Wo, int C, void* stream);
/* x [B, HW, C] -> out fp16 [B, C] (AdaptiveAvgPool2d(1), arcface_resnet.py:142) */
int af_global_avgpool(const void* x, void* out, int B, int HW, int C, void* stream);
/* y = prelu(x * sigmoid(se_logits[b, c]) + residual); se_logits fp16 [B, C] or NULL (use_se = False) (arcface_resnet.py:88-95,153) */
int af_se_residual_prelu(const void* x, const void* se_logits, const void* residual, const void* slope, void* y, int B, int HW, int C,
                         void* stream);
/* Input-gradient kernels of the same layers.  The encoder is frozen (arcface_wrapper.py:65-76) but calc_arcface_align_loss
 * (arcface_wrapper.py:89-166) back-propagates through it into the decoded image, so d/dx is needed and parameter gradients are not.
 * dx = dy * prelu'(x * scale + shift) * scale; x may be NULL when slope is NULL (pure affine) */
int af_affine_prelu_bwd(const void* x, const void* scale, const void* shift, const void* slope, const void* dy, void* dx, int64_t rows,
                        int C, void* stream);
/* x [B, 2Ho, 2Wo, C] (the forward input), dy [B, Ho, Wo, C] -> dx [B, 2Ho, 2Wo, C]: dy to the first maximum of each window */
int af_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int B, int Ho, int Wo, int C, void* stream);
/* dgl fp16 [B, C] = sigmoid'(se_logits) * mean_hw(dpre * x), dpre = dy * prelu'(x * sigmoid(se_logits) + residual): the gradient of
 * the SE logits ALREADY divided by HW (the squeeze's 1/HW, applied early so the fp16 value stays in range; the chain between is linear) */
int af_se_gate_grad(const void* x, const void* se_logits, const void* residual, const void* slope, const void* dy, void* dgl, int B, int HW,
                    int C, void* stream);
/* dx = dpre * sigmoid(se_logits) + dpool[b, c];  dres = dpre.  dpool fp16 [B, C] (the squeeze-branch gradient, 1/HW included) or NULL */
int af_se_residual_prelu_bwd(const void* x, const void* se_logits, const void* residual, const void* slope, const void* dy,
                             const void* dpool, void* dx, void* dres, int B, int HW, int C, void* stream);

/* ---- trainable DoRA adapters on the U-Net's up_blocks.3 convolutions (adaface/diffusers_attn_lora_capture.py:541-591; peft
 * DoraConv2dLayer.forward: y = base(x) + (s - 1) * conv(xd, W) + s * scaling * B(A(xd)), xd = dropout(x)) ---------------------
 * out = y0 + u[c] * c2 + v[c] * lb   (fp16 [rows, C]; u = s - 1, v = s * scaling, fp32 [C]) */
int af_dora_combine(const void* y0, const void* c2, const void* lb, const void* u, const void* v, void* out, int64_t rows, int C,
                    void* stream);
/* out = a * b, fp16 (dropout: b holds 0 or 1 / (1 - p)) */
int af_mul_f16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* x [B,H,W,C] fp16 -> out [B*Ho*Wo, 9*C] with column order (ky, kx, c), zero halo, pad 1: the explicit operand of a 3x3 conv's
 * weight gradient (a GEMM over pixels) */
int af_im2col3x3(const void* x, void* out, int B, int H, int W, int C, int stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAFACE_HIP_H */
