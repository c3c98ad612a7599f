/*
 * spann3r_hip.h -- C-ABI of libspann3r_hip.so: the MI355X (gfx950) hot-path kernels of
 * Spann3R's per-frame forward.
 *
 * The reference is Python on PyTorch; its ONLY native FFI is the `curope` extension
 * (`rope_2d`, croco/models/curope/curope.cpp:49-69).  Everything else on the hot path is an
 * ATen op reached through nn.Module.forward.  This header therefore declares
 *   (1) `sp3_rope_2d`   -- the drop-in for the curope FFI, and
 *   (2) one entry point per ATen-level operator the reference's hot path calls, each citing the
 *       reference call site(s) it replaces.
 * A maintainer binds these with ctypes; spann3r_amd/lib.py is that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`
 *   - `stream` is a hipStream_t passed as void* (use the caller's CURRENT stream; the reference's
 *     curope launches on stream 0, kernels.cu:102 -- a latent bug we do not reproduce)
 *   - return value: 0 on success, non-zero error code otherwise; `sp3_last_error()` gives the text
 *     (the Python binding raises RuntimeError, matching TORCH_CHECK -> RuntimeError)
 *   - activations are fp32 in memory; `wdtype` selects the MFMA arithmetic:
 *       SP3_F32  : v_mfma_f32_16x16x4_f32  (exact fp32; the <=1e-3 parity mode)
 *       SP3_BF16 : v_mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulate; the bench mode)
 *   - no function allocates, frees or synchronises: all of them are hipGraph-capturable
 */
#ifndef SPANN3R_HIP_H
#define SPANN3R_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SP3_F32 = 0, SP3_BF16 = 1 };
enum { SP3_ACT_NONE = 0, SP3_ACT_GELU = 1, SP3_ACT_RELU = 2 };
enum { SP3_EPI_PLAIN = 0, SP3_EPI_ROPE_VT = 1, SP3_EPI_PIXSHUF = 2, SP3_EPI_PARTIAL = 3 };
enum { SP3_LOAD_PLAIN = 0, SP3_LOAD_CONV3X3 = 1, SP3_LOAD_SOFTMAX = 2 };

const char* sp3_last_error(void);
int sp3_version(void);

/* ------------------------------------------------------------------------------------------
 * sp3_gemm : C = epilogue( alpha * A[M,K] . W[N,K]^T )
 * Replaces every nn.Linear / 1x1 Conv2d / 3x3 Conv2d / ConvTranspose2d(k==stride) / einsum on the
 * path: croco/models/blocks.py:74-77,97,110,154-156,167 ; dust3r/model.py:190-191 ;
 * spann3r/model.py:154,174,301,309 ; croco/models/dpt_block.py:33-75,95-113,180-188,318-324,
 * 356-410 ; dust3r/patch_embed.py:24 (after sp3_im2col_patch).
 * A is fp32 [M, lda]; W is [N, K] in `wdtype`; bias fp32 [N].
 * loader CONV3X3: A is an NHWC fp32 map [batchless B*H*W, Cin]; row m = output pixel, K = 9*Cin
 *                 ordered (tap, ci); pad 1; stride conv_stride.
 * epilogue ROPE_VT (fused qkv / kv / q projection of an attention layer): columns
 *   [0, rope_cols) get bias then 2-D RoPE (tables + per-row int32 (y,x) positions) and are stored
 *   row-major to C (ldc) in `wdtype`; columns >= rope_cols are V: stored transposed per head to
 *   vt[((b*heads + h)*64 + d) * vt_ld + n] in `wdtype` (head_dim is 64 on this path).
 * epilogue PIXSHUF: N = ks*ks*Cout, ConvTranspose2d(k=s=ks) scatter into NHWC [B, ks*H, ks*W, Cout].
 * epilogue PARTIAL: K is split over `splitk` workgroups (grid.z); slice z stores its raw fp32 partial sums to
 *   C + z*M*ldc (no bias/activation); sp3_reduce_ln finishes the op (bias + residual + LayerNorm(s)).
 * a_bf16: A (and A2) hold bf16 instead of fp32 (bf16 mode keeps GEMM inputs in bf16: same numerics as converting
 *   on load, half the L2 traffic).  Requires wdtype == SP3_BF16.
 */
typedef struct sp3_gemm_desc {
  const float* A;
  const float* A2;        /* optional second A source for k >= K1 (torch.cat(..., dim=-1) without the copy:
                             spann3r/model.py:300); K1 % 64 == 0; A2 row stride lda2 */
  const void* W;
  void* C;
  const float* bias;
  const float* res1;      /* optional residuals, added after activation: C += res1 (+ res2) */
  const float* res2;
  int32_t M, N, K;
  int32_t batch;          /* grid.y; strides below in elements */
  int32_t K1;             /* k < K1 reads A, k >= K1 reads A2 (set K1 = K when A2 is null) */
  int64_t lda, lda2, ldw, ldc, ldr1, ldr2;   /* ldw: W row stride (>= K) */
  int64_t strideA, strideW, strideC;
  float alpha;
  int32_t wdtype;         /* SP3_F32 | SP3_BF16 : dtype of W and of the MFMA */
  int32_t act;            /* SP3_ACT_* applied after bias */
  int32_t out_bf16;       /* plain epilogue: store C as bf16 instead of fp32 */
  int32_t relu_in;        /* apply ReLU to A on load (ResidualConvUnit pre-activation) */
  int32_t loader;         /* SP3_LOAD_* */
  int32_t conv_H, conv_W, conv_C, conv_OH, conv_OW, conv_stride;
  int32_t epi;            /* SP3_EPI_* */
  const float* rope_cos;  /* [max_pos, 16] */
  const float* rope_sin;
  const int32_t* pos;     /* [M, 2] (y, x) */
  int32_t rope_cols;
  void* vt;
  int32_t tokens;         /* rows per image (N tokens): b = m / tokens, n = m % tokens */
  int32_t heads;
  int64_t vt_ld;
  int32_t ps_k, ps_H, ps_W, ps_C;
  int32_t tile;           /* -1 auto (sp3_gemm_plan tells which); 30..: lean small-M instances (sp3_gemm_plan); 0: 32x32 (K over 4 waves); 1: 64x64; 2: 64x128; 3: 64x64 (K over 4 waves);
                             5 / 6: 128x128 / 128x64 with both operands staged through LDS (bf16 fragment-order A and W,
                             K % 64 == 0) */
  int32_t a_bf16;
  int32_t splitk;         /* >= 1; > 1 only with SP3_EPI_PARTIAL */
  int32_t a_packed;       /* A is in MFMA-fragment order [ceil(M/16)][ceil(K/KB)][2 halves][64 lanes][CH/2] (see w_packed); written
                             that way by the producing kernel (out_packed options), plain loader only */
  /* --- LayerNorm folded into the GEMM (DESIGN.md "LN fold"): for y = LayerNorm(x; gamma, beta) . W^T + b,
   *   y = rstd * (x . (gamma (.) W)^T) - rstd * mean * s + (b + beta . W^T),  s_n = sum_k (gamma (.) W)_nk.
   * CONSUMER side: A is the raw stream x (packed), W is gamma (.) W, bias is the folded bias, ln_s = s, and the row
   * statistics come from ln_stats[M][ln_nt][2] = per-32-column partial (sum, sum of squares) of x written by the
   * PRODUCER of x; mean/rstd are finished per output tile.  PRODUCER side (plain epilogue): stats_out receives those
   * partials of the rows this launch finalises (N % 32 == 0) and c2 a second, fragment-order copy of C in the MFMA
   * dtype (the consumers' A operand).  Replaces the norm1/norm2/norm3/norm_y/value_norm launches
   * (croco/models/blocks.py:128-129,187-190, spann3r/model.py:308). */
  const float* ln_stats;
  const float* ln_s;
  int32_t ln_nt;          /* number of 32-column groups of x = C_x / 32 */
  int32_t ln_C;           /* C_x */
  float ln_eps;
  float* stats_out;       /* [M][N/32][2] or null */
  void* c2;               /* packed copy of C (dims M x N) in the MFMA dtype, or null */
  int32_t qkv_packed;     /* ROPE_VT epilogue, bf16: q/k stored in fragment order (rows = b*vt_ld + n, K-dim = rope_cols) and
                             V in PV-operand order [(b,h)][vt_ld/32][4][64][8] -- the layouts sp3_attention_packed reads */
  int32_t out_packed;     /* plain epilogue: store C in fragment order (it is the next GEMM's packed A; dims M x N) */
  int32_t w_packed;       /* W is in MFMA-fragment order [ceil(N/16)][ceil(K/KB)][2 halves][64 lanes][CH/2], zero padded; KB/CH =
                             64/16 (bf16) or 32/8 (fp32); lane = 16*g + r holds row 16*nb + r, k = kb*KB + g*CH + e: e < CH/2 in the first 1 KB
                             half of the 2 KB block, the rest in the second (every wave access is one contiguous KB).
                             2 (fp32 W, product mode f16x3 only): same blocks, but a lane's 32 bytes hold the fp16 planes of its 8 k --
                             first half h = fp16(w), second half l = fp16((w - h) * 2^11), the split the kernel would take per use */
  /* --- grouped launches (batch > 1, grid.y): `batch` independent problems of one shape in one launch, e.g. the two
   *   sides of a DUSt3R decoder layer (different weights, dust3r/model.py:196-198).  A / W / C advance by strideA /
   *   strideW / strideC ELEMENTS per batch index (C in its own dtype, any epilogue, also the packed layouts), res1 /
   *   res2 are [batch][M][ld], and the operands below advance by these BYTE offsets.  Offsets may be negative (the
   *   cross-attention k/v projection of side z reads the tokens and statistics of side 1-z).  rope tables / pos are
   *   shared. */
  int64_t sb_A2, sb_bias, sb_ln_stats, sb_ln_s, sb_stats_out, sb_c2, sb_vt;
  /* --- diagnostics: if non-null, the first 8 workgroups of a role-loop launch (tiles 13-15) store shader-clock stamps of
   *   their own timeline: trace[wg*64 + 0] = entry, [1] = loop done, [2] = epilogue done, [8 + i/4] = loader passed the
   *   barrier of k-block i (i % 4 == 0), [32 + i/4] = consumer wave 0 finished k-block i.  tools/trace_gemm.py prints it. */
  int64_t* trace;
  /* --- the spatial-memory read as two launches (spann3r/model.py:159-183: softmax over the bank, probabilities below
   *   attn_thresh dropped, renormalised, times V):
   *   launch 1, the score GEMM (plain loader/epilogue): sm_stats_out[M][ceil(N/32)][2] receives, per row and 32-column
   *     group of the finished C, (max, sum exp(x - max)) -- the softmax statistics leave with the scores.
   *   launch 2, loader SP3_LOAD_SOFTMAX (32x32 tile): A is the fp32 score matrix [M, lda] (K = bank tokens, any multiple
   *     of 4); each lane merges sm_stats[M][sm_nt][2] of its rows to (m, Z) and turns the scores it loads into
   *     p = exp(a - m) / Z, p < sm_thresh -> 0, on their way into the MFMA; C = (p . W^T) / sum(kept p) (+ bias, res);
   *     sm_zout[M][4] receives (sum(kept p), m, 1/Z, -) per row.  No probability matrix exists in memory.
   *     sp3_colsum_softmax finishes the column sums (mem_attn) from the scores and sm_zout. */
  float* sm_stats_out;
  const float* sm_stats;
  int32_t sm_nt;
  float sm_thresh;
  float* sm_zout;
  /* --- fp32 operands (wdtype SP3_F32, A fp32), register-ring tiles 0-3: products through three bf16 MFMAs per k-block
   *   (hi.hi + hi.lo + lo.hi of a bf16 split of both operands: 16 mantissa bits per product, fp32 accumulate) instead of the
   *   fp32 MFMA -- 2-3x the matrix rate, the "f32x3" precision mode of the model.  Ignored for bf16 operands. */
  int32_t f32x3;
  /* --- dtype of the residual maps res1 / res2 (their pointer type says fp32): 1 = they hold bf16.  Only the lean small-map
   *   convolutions (tiles 40 / 41) read bf16 residuals, and only next to a bf16 map and output (res_bf16 == out_bf16 == a_bf16);
   *   sp3_gemm_plan answers -1 for any other combination and the general tiles refuse res_bf16 != 0, so a descriptor whose
   *   residual dtype differs from its output dtype is never routed to a kernel that would reinterpret the bytes. */
  int32_t res_bf16;
  /* --- the growing extent of the spatial-memory read as DEVICE state (round 6): if non-null, the score GEMM of the read (tile 43:
   *   sm_stats_out set) takes its column count N from dyn_n[0] and its P.V GEMM (tile 44: loader SP3_LOAD_SOFTMAX) its contraction
   *   length K (and sm_nt = ceil(K / 32)) -- the bank's current token count, spann3r/model.py:80-95 grows it by a frame per step.
   *   The descriptor's N / K then only bound what the launch can serve (grid size, argument checks; dyn_n[0] <= that bound,
   *   dyn_n[0] % 4 == 0), so ONE captured hipGraph serves every bank length up to it.  Lean instances only: the general tiles
   *   refuse a descriptor with dyn_n set. */
  const int32_t* dyn_n;
} sp3_gemm_desc;
int sp3_gemm(const sp3_gemm_desc* desc_host, void* stream);
/* Two differently shaped groups of problems in ONE launch (grid.y = a.batch + b.batch), e.g. a decoder layer's self-attention
 * q/k/v projection (N = 2304) and its cross-attention k/v projection (N = 1536): both read the previous layer's tokens
 * (dust3r/model.py:196-198, croco/models/blocks.py:187-189), so the second launch boundary buys nothing.  Both descriptors
 * must resolve to the same kernel instance (dtypes, loader, tile; b.tile < 0 takes a's); no split-K. */
int sp3_gemm2(const sp3_gemm_desc* a, const sp3_gemm_desc* b, void* stream);
/* The tile sp3_gemm runs `desc` on when desc->tile < 0 (desc->tile itself is ignored), or < 0 for an invalid descriptor.  Tiles
 * 30.. are the LEAN small-M instances (csrc/gemm_sm.hip): the per-frame step's 196-row weight-streaming Linears
 * (croco/models/blocks.py:73-79,94-112,149-169 at batch 1) on bf16 fragment-order operands, with shape, tile and epilogue fixed at
 * compile time -- 30 / 31: q/k/v projections (K = 1024 / 768, ROPE_VT epilogue with qkv_packed), 32 / 33: fc1 + GELU into
 * fragment order (K = 1024 / 768), 34..38: output projections onto the fp32 residual stream (K = 1024, 4096, 768, 3072, 1792).
 * They serve M <= 256, batch <= 2, bias set, alpha = 1, no split-K / second residual / split A; anything else runs on the general
 * tiles above.  40 / 41: loader CONV3X3 on maps of <= 256 / <= 2048 output pixels (NHWC map, residuals and output all fp32 or all bf16 --
 * a_bf16 = out_bf16 = res_bf16 --, bf16 fragment-order weights, K = 9 Cin a multiple of 64, plain epilogue with bias / ReLU / two residuals): the DPT heads' small-map convolutions
 * (croco/models/dpt_block.py:33-75,95-113) in ONE launch instead of split-K partials + sp3_reduce_ln.  SP3_LEAN_GEMM=0 in the environment switches them off (A/B runs). */
int sp3_gemm_plan(const sp3_gemm_desc* desc_host);

/* ------------------------------------------------------------------------------------------
 * sp3_layernorm : nn.LayerNorm over the last dim (croco/models/blocks.py:128-129,187-190 eps 1e-6;
 * dust3r/model.py:153,204; spann3r/model.py:154,174,308 -- norm_q/k/v eps 1e-5).
 * x fp32 [rows, ldx] -> out [rows, ldo] (fp32, or bf16 if out_bf16). C % 4 == 0, C <= 4096.
 */
int sp3_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                  void* out, int64_t ldo, int out_bf16, int rows, int C, void* stream);
/* Same, output in fragment order (see sp3_gemm_desc.a_packed). */
int sp3_layernorm_packed(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                         void* out, int out_bf16, int rows, int C, void* stream);
/* fp32 row-major output (as sp3_layernorm) AND a bf16 fragment-order copy for GEMM consumers in the same launch: rows are taken
 * in groups of group_rows (one image / one decoder side) and group g starts at packed row g * group_rows_pad (a multiple of 16),
 * so every group is an a_packed operand of its own.  enc_norm / dec_norm (dust3r/model.py:153,204), whose outputs are both
 * API-visible (fp32) and the A operands of decoder_embed / the key MLPs (dust3r/model.py:190-191, spann3r/model.py:299-303). */
int sp3_layernorm_dual(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* out, int64_t ldo,
                       void* out_packed_bf16, int rows, int C, int group_rows, int group_rows_pad, void* stream);
/* Same, but stores the result TRANSPOSED: out[c * ldo + row] (used to append LN_v(value) columns
 * to the [1024, capacity] V^T bank of the spatial memory). */
int sp3_layernorm_t(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                    void* out, int64_t ldo, int out_bf16, int rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * sp3_reduce_ln : finishes a split-K GEMM and fuses what the reference does next on the residual stream:
 *   x = sum_s partial[s] + bias (+ res);  x_out = x (fp32, optional);
 *   out1 = LayerNorm(x; g1,b1) (optional), out2 = LayerNorm(x; g2,b2) (optional)  -- fp32 or bf16.
 * Replaces `x = x + drop_path(proj/fc2(...))` followed by the next norm1/norm2/norm3/norm_y/enc_norm/dec_norm
 * (croco/models/blocks.py:128-129,187-190; dust3r/model.py:153,204).  One wave per row; C % 4 == 0, C <= 4096.
 */
typedef struct sp3_reduce_ln_desc {
  const float* partial;   /* [splits][rows][C] fp32, split stride in elements */
  int64_t split_stride;
  const float* bias;      /* [C] or null */
  const float* res;       /* [rows, ldres] or null */
  int64_t ldres;
  float* x_out;           /* [rows, ldx] or null (may alias res) */
  int64_t ldx;
  const float* g1; const float* b1; void* out1; int64_t ld1; int32_t out1_bf16;
  const float* g2; const float* b2; void* out2; int64_t ld2; int32_t out2_bf16;
  float eps;
  int32_t splits, rows, C;
  int32_t out1_packed, out2_packed;   /* store the LayerNorm output in fragment order (GEMM a_packed operand) */
  int32_t act;            /* SP3_ACT_NONE | SP3_ACT_RELU, applied to sum + bias BEFORE the residuals (sp3_gemm's order) */
  const float* res2;      /* second residual [rows, ldres2] or null: x = act(sum + bias) + res + res2 */
  int64_t ldres2;
} sp3_reduce_ln_desc;
int sp3_reduce_ln(const sp3_reduce_ln_desc* desc_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * sp3_rope_2d : drop-in for curope.rope_2d (curope.cpp:49-69, kernels.cu:17-108).
 * tokens [B,N,H,D] modified in place; element (b,n,h,d) at tokens + b*sB + n*sN + h*sH + d
 * (the reference only requires stride(3)==1 && stride(2)==D, kernels.cu:91; sH generalises it);
 * positions int64 [B,N,2] contiguous; D % 4 == 0; dtype SP3_F32 | SP3_BF16; fwd = +1 / -1.
 */
int sp3_rope_2d(void* tokens, int dtype, int B, int N, int H, int D, int64_t sB, int64_t sN, int64_t sH,
                const int64_t* positions, float base, float fwd, void* stream);

/* ------------------------------------------------------------------------------------------
 * sp3_attention : softmax(q k^T * scale) v per head, head_dim 64
 * (croco/models/blocks.py:105-109 and 162-166).  q [B,Nq,heads,64] / k [B,Nk,heads,64] with row
 * strides ldq/ldk and batch strides sq/sk (elements, `dtype`), already RoPE'd; vt as written by
 * the ROPE_VT epilogue; out fp32 (or bf16 if out_bf16) [B*Nq, ldo] with head h at columns [64h, 64h+64).
 * dtype: SP3_F32 | SP3_BF16 | 2 = fp32 operands with every product through three bf16 MFMAs of a (hi, lo) split (the
 * "f32x3" precision of the model: 16 mantissa bits per product, fp32 softmax and accumulation).
 */
int sp3_attention(const void* q, int64_t sq, int64_t ldq, const void* k, int64_t sk, int64_t ldk,
                  const void* vt, int64_t vt_ld, void* out, int64_t ldo, int out_bf16,
                  int B, int heads, int Nq, int Nk, float scale, int dtype, void* stream);
/* out_packed != 0: out is stored in fragment order as a [B*Nq, heads*64] GEMM operand (ldo ignored) */
int sp3_attention_ex(const void* q, int64_t sq, int64_t ldq, const void* k, int64_t sk, int64_t ldk,
                     const void* vt, int64_t vt_ld, void* out, int64_t ldo, int out_bf16, int out_packed,
                     int B, int heads, int Nq, int Nk, float scale, int dtype, void* stream);

/* bf16 attention on the fragment-order q/k and PV-order V written by the qkv_packed ROPE_VT epilogue: every operand load
 * is one contiguous wave read, the next key tile is prefetched during the softmax.  qp/kp: packed [B*npad_q|npad_k, q_cols|k_cols]
 * matrices, head h of q at columns q_col0 + 64h (k: k_col0 + 64h); vtp as described at qkv_packed (vt_ld = npad_k).
 * out: fp32/bf16 row-major [B*Nq, ldo] or fragment order (out_packed).  o_group > 0: every o_group consecutive images form
 * one problem of a grouped launch whose output rows start at a multiple of o_group_rows (image b -> row
 * (b / o_group) * o_group_rows + (b % o_group) * Nq), so each problem's packed rows stay 16-aligned; 0: rows b * Nq. */
int sp3_attention_packed(const void* qp, int q_cols, int q_col0, int npad_q, const void* kp, int k_cols, int k_col0, int npad_k,
                         const void* vtp, void* out, int64_t ldo, int out_bf16, int out_packed,
                         int B, int heads, int Nq, int Nk, float scale, int o_group, int o_group_rows, void* stream);

/* sp3_attention_packed with the QUERY PROJECTION inside the launch (round 6): a decoder layer's cross-attention
 * (croco/models/blocks.py:149-169) reads q = RoPE2D(projq(norm2(x))) of its own side; instead of a 196 x 768 x 768 GEMM launch in front
 * of the attention launch, every attention workgroup (16 query rows x one head) computes its [16 x 64] q tile from the fragment-order
 * bf16 copy of x, the producer's LayerNorm partials (folded norm2: ln_s = column sums of gamma (.) Wq, bias = bq + Wq beta) and the
 * fragment-order weight.  Images come in groups (the two decoder sides of a grouped launch): image b belongs to group b / o_group;
 * its rows are (b % o_group) * Nq + token inside the group's x / statistics / positions; the *_group_stride fields step from group
 * to group (elements of bf16, floats).  k / v / out arguments as sp3_attention_packed.  D = 768 (12 heads), Nq < 512. */
typedef struct sp3_attn_qproj_desc {
  const void* x_packed; int64_t x_group_stride;
  const float* ln_stats; int64_t stats_group_stride;   /* [rows][D / 32][2] */
  const void* w_packed; int64_t w_group_stride;         /* [D][D] fragment order */
  const float* ln_s; const float* bias; int64_t vec_group_stride;
  const int32_t* pos; const float* rope_cos; const float* rope_sin;   /* [rows of one group][2] (y, x); tables [max_pos][16] */
  float ln_eps; int32_t D;
  const void* kp; int32_t k_cols, k_col0, npad_k; const void* vtp;
  void* out; int64_t ldo; int32_t out_bf16, out_packed;
  int32_t B, heads, Nq, Nk; float scale; int32_t o_group, o_group_rows;
} sp3_attn_qproj_desc;
int sp3_attention_packed_qproj(const sp3_attn_qproj_desc* desc_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spatial-memory kernels (spann3r/model.py:97-210).
 * The bank (spann3r_amd/model.py SpatialMemory) keeps, next to the reference's mem_k / mem_v / mem_attn / mem_count, the
 * two operands of a read already normalised and in sp3_gemm's fragment order (w_packed): K' = gamma_q (.) LN_k(mem_k)
 * [cap, 1024] and LN_v(mem_v)^T [1024, cap], plus s_bank / b_bank [cap] that fold LN_q into the S GEMM.
 * sp3_bank_write   : one launch per stored frame (:80-95 plus the bank-side LayerNorms of :154,174): raw copies, K', V^T,
 *   s_bank[t] = alpha * sum_c K'[t,c] (of the rounded operand), b_bank[t] = alpha * sum_c beta_q[c] LN_k(k)[t,c].
 * sp3_softmax_thresh: rows of S fp32 [rows, ld]: softmax over the first M columns; if thresh > 0, p < thresh -> 0 and
 *   renormalise (:160,170-172).  Outputs (either may be null): P fp32 [rows, ld] (columns [M, Mpad) written as 0) and
 *   P_packed = the probabilities as sp3_gemm's a_packed operand [rows, K] with K = M rounded up to one k-block (64 bf16 /
 *   32 fp32, zero filled), stride_packed elements per batch entry.  batch in grid.y via strideS.
 * sp3_colsum_accum : mem_attn[j] += sum_r P[r, j]  (:180-181) from the row-major fp32 P;
 * sp3_colsum_packed: the same from the fragment-order copy (one writer per column, fixed order: deterministic).
 * sp3_pack_stats   : fragment-order copy + per-32-column (sum, sum of squares) partials of a row-major fp32 matrix (what
 *   a producer GEMM's c2 / stats_out write), for callers that hand memory_read a plain tensor.
 * sp3_cos_sim      : score[t] = mean_p cos(k[p,:], wm[t,p,:]) for t < T (:102-112); k fp32 [P,C],
 *   wm fp32 [T,P,C] contiguous; scratch fp32 [T*P] holds the per-patch cosines (one wave per (t,p) pair, then one
 *   block per t averages them). Deterministic reduction order.
 * sp3_mem_append   : count[0..M) += 1; count[M..M+P) = 0; attn[M..M+P) = 0  (:84-90).
 * sp3_prune_select : w = attn/count, w[count < protect] = 1e8; sel[0..top_k) = indices of the top_k
 *   weights, sorted by weight descending, ties by index ascending (:187-193). M <= 16384.
 * sp3_gather_rows  : dst[i, :] = src[sel[i], :] (row gather of a [*, C] matrix, elem_size bytes/elem).
 * sp3_gather_cols  : dst[c, i] = src[c, sel[i]] for c < C (a row-major V^T), zero-fills [n_sel, n_fill).
 * sp3_gather_packed_rows / _cols : the same two gathers on fragment-order matrices ([*, C] rows = tokens; [C, cap] with
 *   k = token; the column gather zero-fills tokens [n_sel, n_fill)).
 * sp3_gather_1d    : dst[i] = src[sel[i]] fp32.
 */
typedef struct sp3_bank_write_desc {
  const float* feat_k;    /* [P, C] fp32: the frame's memory key   (feat_k1) */
  const float* feat_v;    /* [P, C] fp32: the frame's memory value (cur_v + feat_k1) */
  float* k_raw;           /* [cap, C] fp32 = reference mem_k; rows [M, M+P) are written */
  float* v_raw;           /* [cap, C] fp32 = reference mem_v */
  void* k_hat;            /* fragment-order [cap, C] in wdtype: gamma_q (.) LN_k(mem_k) */
  void* v_hat_t;          /* fragment-order [C, cap] in wdtype: LN_v(mem_v)^T */
  float* s_bank;          /* [cap] */
  float* b_bank;          /* [cap] */
  const float *gamma_k, *beta_k, *gamma_v, *beta_v, *gamma_q, *beta_q;   /* norm_k / norm_v / norm_q, fp32 [C] */
  float eps;              /* 1e-5 (spann3r/model.py:245-247) */
  float alpha;            /* 1 / sqrt(C): the S GEMM's scale, folded into s_bank / b_bank */
  int32_t M, P, C, cap;   /* cap % 64 == 0, C % 256 == 0 */
  int32_t wdtype;         /* SP3_F32 | SP3_BF16 */
  const int32_t* state;   /* round 6: if non-null, the first row M is read from state[0] on the device (sp3_bank_state_set) and the M
                           * field is ignored: one captured launch serves every fill level; rows >= cap are not written */
} sp3_bank_write_desc;
int sp3_bank_write(const sp3_bank_write_desc* desc_host, void* stream);
int sp3_softmax_thresh(const float* S, float* P, int64_t ld, int64_t strideS, int rows, int M, int Mpad,
                       float thresh, int batch, void* P_packed, int64_t stride_packed, int packed_bf16, void* stream);
/* the packed-only form of sp3_softmax_thresh for long banks, two streaming launches (row statistics; then one contiguous kilobyte of
 * fragment-order probabilities per wave store): same arithmetic with v_exp_f32; rowstat_ws: batch * rows * 4 floats */
int sp3_softmax_pack(const float* S, int64_t ld, int64_t strideS, int rows, int M, float thresh, int batch, void* P_packed,
                     int64_t stride_packed, int packed_bf16, float* rowstat_ws, void* stream);
int sp3_colsum_accum(const float* P, int64_t ld, int rows, int M, float* mem_attn, void* stream);
int sp3_colsum_packed(const void* P_packed, int packed_bf16, int rows, int M, float* mem_attn, void* stream);
/* the column sums of the two-launch read (sp3_gemm, loader SP3_LOAD_SOFTMAX): mem_attn[j] += sum_r [p >= thresh] p / Z'_r,
 * p = exp(S[r, j] - m_r) / Z_r with rowz[r] = (Z', m, 1/Z, -) = that launch's sm_zout; fixed summation order.  append_P > 0
 * also does sp3_mem_append(mem_count, mem_attn, M, append_P) (the frame that was read against is committed in the same launch). */
int sp3_colsum_softmax(const float* S, int64_t ld, int rows, int M, const float* rowz, float thresh, float* mem_attn,
                       float* mem_count, int append_P, void* stream);
int sp3_pack_stats(const float* x, int64_t ldx, int rows, int C, void* packed, int packed_bf16, float* stats, void* stream);
int sp3_gather_packed_rows(const void* src, void* dst, const int32_t* sel, int n_sel, int C, int elem_size, void* stream);
int sp3_gather_packed_cols(const void* src, void* dst, const int32_t* sel, int n_sel, int n_fill, int C, int cap,
                           int elem_size, void* stream);
int sp3_cos_sim(const float* k, const float* wm, int T, int P, int C, float* scratch, float* score, void* stream);
/* Device-resident fill state of a bank (round 6): state[0] = M (tokens stored), state[1] = wm (frames of working memory,
 * spann3r/model.py:120-143).  The host owns the policy (commit / skip / prune are its decisions, as in the reference) and pushes the
 * values after every change; the kernels of a step read them from the device, so the step's hipGraph does not depend on them.
 * sp3_cos_sim_state = sp3_cos_sim against the last state[1] frames of k_raw [cap, C] (rows [M - wm P, M)), wm <= Tmax: launched
 * for Tmax frames, surplus workgroups exit; score[t] for t >= wm is left untouched. */
int sp3_bank_state_set(int32_t* state, int M, int wm, void* stream);
/* The long-bank read without a score matrix (round 6; spann3r/model.py:159-183 at attn_thresh = 0, > 256 query rows).  Stage 1 =
 * sp3_gemm with sm_stats_out AND a fragment-order bf16 output (lean tile 45): C = p~ = exp(s - m_g) per 64-key group g of a row,
 * [M rows][ldc = bank capacity], sm_stats_out[ldc / 64][M rounded up to 256] = (m_g, sum_g p~) as float2.  sp3_prob_merge turns the
 * statistics into scale[g][row] = exp(m_g - m_row) / Z_row (same [ldc / 64][rows_pad] layout, fp32).  Stage 2 = sp3_gemm with loader
 * SP3_LOAD_SOFTMAX in its probability form (lean tile 46: A = p~ a_packed bf16, sm_stats = scale, PARTIAL epilogue, splitk a multiple of
 * 8, ldw = the capacity): partial[s] = sum over the groups of slice s of scale[g] * (p~_g . V_hat_g); sp3_reduce_ln adds the slices and
 * q.  sp3_colsum_prob: mem_attn[key] += sum_rows p~[row, key] * scale[key / 64][row] (:180-181).  M = bank tokens (N of stage 1, K of
 * stage 2), or dyn_n[0] if dyn_n is set (then M only sizes the launch). */
int sp3_prob_merge(const float* stats, float* scale, int rows, int M, int cap, const int32_t* dyn_n, void* stream);
int sp3_colsum_prob(const void* P_packed, const float* scale, int rows, int M, int cap, const int32_t* dyn_n, float* mem_attn, void* stream);
int sp3_cos_sim_state(const float* k, const float* k_raw, int Tmax, int P, int C, const int32_t* state, float* scratch, float* score,
                      void* stream);
int sp3_mem_append(float* count, float* attn, int M, int P, void* stream);
int sp3_prune_select(const float* attn, const float* count, int M, float protect, int top_k,
                     int32_t* sel, void* stream);
int sp3_gather_rows(const void* src, void* dst, const int32_t* sel, int n_sel, int C, int elem_size, void* stream);
int sp3_gather_cols(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* sel, int n_sel,
                    int n_fill, int C, int elem_size, void* stream);
int sp3_gather_1d(const float* src, float* dst, const int32_t* sel, int n_sel, void* stream);

/* ------------------------------------------------------------------------------------------
 * DPT-head helpers (all maps NHWC fp32).
 * sp3_im2col_patch : non-overlapping p x p patches -> rows [B*(H/p)*(W/p), C*p*p], k = (c, py, px)
 *   so the Conv2d weight [E, C, p, p] is used unchanged as W[E, C*p*p] (dust3r/patch_embed.py:24,
 *   spann3r/model.py:317). Source element (b,c,y,x) at img + b*sb + c*sc + y*sy + x*sx.
 * sp3_upsample2x   : F.interpolate(scale_factor=2, mode='bilinear', align_corners=True)
 *   (croco/models/dpt_block.py:214-216, 253-258) on [B,H,W,C] -> [B,2H,2W,C]; optional crop of the
 *   output to [outH, outW] (dust3r/heads/dpt_head.py:57).
 * sp3_head_final   : head.4 1x1 conv (128->4) + postprocess (dust3r/heads/postprocess.py:22-58):
 *   pts3d = xyz/max(|xyz|,1e-8) * expm1(|xyz|), conf = 1 + exp(c).  feat [pixels, C] fp32,
 *   w fp32 [4, C], b fp32 [4] -> pts [pixels,3], conf [pixels], raw [pixels,4] (optional).
 */
/* sp3_conv3x3_tile : Conv2d(3x3, stride 1, padding 1) on an NHWC map [B,H,W,Cin] -> [B,H,W,Cout] as an LDS-tiled
 *   implicit GEMM (bf16 MFMA, fp32 accumulation): out = act(conv(relu_in ? relu(x) : x) + bias) + res1 + res2.
 *   x fp32 or bf16 (in_bf16); w_packed = the [Cout, 9*Cin] weight (k = (ky*3+kx)*Cin + ci) as bf16 in sp3_gemm's
 *   w_packed fragment order; bias fp32[Cout] / res1 / res2 fp32 NHWC maps of the output shape, all optional;
 *   out fp32 or bf16 (out_bf16).  Cin, Cout multiples of 64; act is SP3_ACT_NONE or SP3_ACT_RELU.
 *   Replaces the Conv2d calls of ResidualConvUnit_custom (croco/models/dpt_block.py:120-142), scratch.layer_rn
 *   (:180-188) and the head convs (:318-324) in bf16 mode; fp32 mode and stride 2 use sp3_gemm's LOAD_CONV3X3.  out_bf16: bit 0 = the output map is bf16, bit 1 = the residual maps res1 / res2 are bf16 too (else fp32).
 *   Bits 2-3 choose the workgroup's tile: 0 = by size, 1 = 8 x 8 pixels x 64 channels, 2 = 8 rows x 16 pixels x 64 channels, 3 = 8 x 16
 *   pixels x 32 channels (2 and 3 need Cin % 128 == 0: the halo is staged 128 channels at a time; half the weight stream per flop).
 *   By size: the 64-channel wide tile once its grid has >= 256 workgroups, the 32-channel one from 80 (batch-1 56 x 56 maps:
 *   112 -> 224 workgroups), else 8 x 8.  All tiles give the same sums up to fp32 addition order.
 */
int sp3_conv3x3_tile(const void* x, int in_bf16, const void* w_packed, const float* bias, const float* res1,
                     const float* res2, void* out, int out_bf16, int B, int H, int W, int Cin, int Cout,
                     int relu_in, int act, void* stream);
int sp3_im2col_patch(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int C, int H, int W,
                     int p, void* out, int out_bf16, int out_packed, void* stream);
int sp3_upsample2x(const float* in, float* out, int B, int H, int W, int C, int outH, int outW, void* stream);
/* the same on bf16 NHWC maps (bf16 mode of the DPT heads keeps its feature maps in bf16; interpolation arithmetic in fp32) */
int sp3_upsample2x_bf16(const void* in, void* out, int B, int H, int W, int C, int outH, int outW, void* stream);
int sp3_head_final(const float* feat, const float* w, const float* b, int64_t pixels, int C, float* pts,
                   float* conf, float* raw, void* stream);
/* the same with the feature map stored as bf16 (outputs stay fp32: the API's pointmaps / confidences) */
int sp3_head_final_bf16(const void* feat, const float* w, const float* b, int64_t pixels, int C, float* pts, float* conf,
                        float* raw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline: one decoded RGB frame (uint8 HWC, row stride src_row_stride bytes) -> the normalised
 * float image the model consumes.  Replaces PIL Image.crop + Image.resize(LANCZOS) (dust3r/datasets/utils/cropping.py:54-111),
 * ImgNorm (dust3r/utils/image.py:23) and transpose_to_landscape (dust3r/datasets/base/base_stereo_view_dataset.py:215-220):
 *   crop window [crop_t, +H1) x [crop_l, +W1)  ->  Pillow's 8-bit LANCZOS resample to W2 x H2 (horizontal pass into `tmp`
 *   [H1, W2, 3] uint8, then vertical; bounds [n, 2] = (first tap, taps) and coef [n, ksize] = 22-bit fixed-point weights per
 *   output index, computed by the host: spann3r_amd/preprocess.py)  ->  crop [crop2_t, +outH) x [crop2_l, +outW)  ->
 *   (u/255 - 0.5)/0.5  ->  out fp32 [3, outH, outW], or [3, outW, outH] if `transpose` (portrait rectified to landscape).
 * Bit-exact with Pillow for the uint8 stage. */
int sp3_preprocess_image(const uint8_t* src, int64_t src_row_stride, int crop_l, int crop_t, int H1, int W1,
                         const int32_t* hbounds, const int32_t* hcoef, int hksize, int W2,
                         const int32_t* vbounds, const int32_t* vcoef, int vksize, int H2,
                         int crop2_l, int crop2_t, int outW, int outH, int transpose, uint8_t* tmp, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training criterion: ConfLoss_t(Regr3D_t(L21, norm_mode='avg_dis', fix_first), alpha).compute_frame_loss
 * (spann3r/loss.py:20-84,129-285; dust3r/losses.py:52-56) forward and backward on stacked buffers:
 *   P [E, B, HW, 3] predicted pointmaps and Cf [E, B, HW] confidences, one slab per loss entry in the reference's order
 *   L0, L1, R1, L2, R2, ..., R_{n-1} (E = 2 (n-1); L_i = preds_all[i][0], R_i = preds_all[i-1][1]);
 *   G [n, B, HW, 3] ground-truth points (world), V [n, B, HW] validity (bytes), pose0 [B, 16] camera pose of view 1 (affine).
 * forward: out7 (device) = factor loss, loss, conf_loss_1, conf_loss2, conf_mean, pts3d_1, pts3d_2 (the reference's details);
 *   ent_out (device, nullable) [E, 2] = (mean error, conf loss) per entry; ws = device scratch of sp3_conf_loss_ws_bytes(n, B)
 *   bytes that the backward reads.  backward: dP, dC = gradients of grad_scale2[0] * loss + grad_scale2[1] * factor loss
 *   (grad_scale2: 2 floats on the device).  Sums in double, fixed order: deterministic. */
int64_t sp3_conf_loss_ws_bytes(int n, int B);
int sp3_conf_loss_forward(const float* P, const float* Cf, const float* G, const uint8_t* V, const float* pose0, int n, int B, int HW,
                          float alpha, int fix_first, void* ws, float* out7, float* ent_out, void* stream);
int sp3_conf_loss_backward(const float* P, const float* Cf, const float* G, const uint8_t* V, const float* pose0, int n, int B, int HW,
                           float alpha, int fix_first, const void* ws, const float* grad_scale2, float* dP, float* dC, void* stream);
/* Regr3D_t_ScaleShiftInv(L21, gt_scale) forward (spann3r/loss.py:292-368; the validation criterion of spann3r/training.py:39,152):
 * same stacked buffers as sp3_conf_loss_forward (no confidences); the joint median-depth shift and median-centre / median-norm
 * scale (torch.nanmedian: lower middle element) are radix selections on the device.  out (device, 6 + E floats): loss (the SUM over
 * the E entries of the mean Euclidean error), factor loss, mean gt_shift_z, pred_shift_z, gt_scale, pred_scale, then the mean error of
 * every entry.  ws: sp3_ssi_loss_ws_bytes(n, B, HW) bytes of device scratch. */
int64_t sp3_ssi_loss_ws_bytes(int n, int B, int HW);
int sp3_ssi_loss_forward(const float* P, const float* G, const uint8_t* V, const float* pose0, int n, int B, int HW, int fix_first,
                         int gt_scale, void* ws, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pieces of the BACKWARD of the spatial-memory read in its training form (attn_thresh = 0, mem_dropout; forward:
 * spann3r/model.py:145-183), between the four sp3_gemm launches of spann3r_amd/train.py:
 * sp3_transpose     : dst[c * ld_dst + r] = src[r * ld_src + c]  (sp3_gemm computes A[M,K] . W[N,K]^T only)
 * sp3_mul           : out = a (.) b  (the dropout mask on the probabilities, :168)
 * sp3_softmax_bwd   : dS = alpha * A (.) (dA - rowsum(dA (.) A)), dA = dAd (.) mask (mask nullable), rows x T, row stride ld
 * sp3_layernorm_bwd : dx = LayerNorm backward of dy w.r.t. x (+ dx_add, nullable: the residual path), dgamma / dbeta [C]
 *                     (accumulate != 0: added to their current contents); scratch: ceil(rows / 4) * 2 * C floats.  Column sums in a
 *                     fixed order: deterministic. */
int sp3_transpose(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols, void* stream);
int sp3_transpose_batched(const float* src, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst, int64_t stride_dst, int rows,
                          int cols, int batch, void* stream);       /* `batch` matrices, element strides between them */
/* same, and columns [rows, pad_to) of every destination row are written as zeros (pad_to <= ld_dst, pad_to - rows < 32): the
 * contraction pad of the next GEMM without a separate fill launch */
int sp3_transpose_pad(const float* src, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst, int64_t stride_dst, int rows,
                      int cols, int batch, int pad_to, void* stream);
/* exact-erf GELU (nn.GELU, croco/models/blocks.py:73-79) of the train-mode blocks and its backward dx = dy * gelu'(x) */
int sp3_gelu(const float* x, float* y, int64_t n, void* stream);
int sp3_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* train-mode DPT head pieces (croco/models/dpt_block.py:120-218, dust3r/heads/postprocess.py:10-58), NHWC fp32:
 * sp3_im2col3x3 / sp3_col2im3x3 : the 3x3, pad 1, stride 1|2 convolution as a GEMM over col [B*OH*OW, 9*C] (k = (ky*3+kx)*C + c) and the
 *   adjoint gather that turns d col back into d x;  sp3_relu / sp3_relu_bwd;  sp3_upsample2x_bwd : adjoint of sp3_upsample2x;
 * sp3_postprocess(_bwd) : raw [M, 4] -> pts3d [M, 3] = xyz / |xyz| * expm1(|xyz|), conf [M] = 1 + exp(c), and its backward. */
int sp3_im2col3x3(const float* x, float* col, int B, int H, int W, int C, int stride, void* stream);
int sp3_col2im3x3(const float* dcol, float* dx, int B, int H, int W, int C, int stride, void* stream);
int sp3_relu(const float* x, float* y, int64_t n, void* stream);
int sp3_relu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int sp3_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, int outH, int outW, void* stream);
int sp3_postprocess(const float* raw, float* pts, float* conf, int64_t M, void* stream);
int sp3_postprocess_bwd(const float* raw, const float* dpts, const float* dconf, float* draw, int64_t M, void* stream);
/* torch.optim.AdamW update of one parameter tensor (spann3r/training.py:327; decoupled weight decay, bias correction for `step` >= 1);
 * the gradient is multiplied by grad_scale first (1 / accum_iter, or the clip coefficient of clip_grad_norm_, croco/utils/misc.py:274). */
int sp3_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
              int step, float grad_scale, void* stream);
int sp3_mul(const float* a, const float* b, float* out, int64_t n, void* stream);
/* bf16 training step (spann3r/training.py:170-259 under bf16 autocast): fp32 row-major [rows, cols] (row stride ld) -> the GEMM's bf16
 * fragment order, as [rows, cols] (dst, nullable) and / or as its transpose [cols, rows] (dstT, nullable) in one pass; pads are written
 * as zeros.  With both forms every product of a Linear's backward (dX = dY . W, dW = dY^T . X) is an sp3_gemm A . W^T launch on packed
 * bf16 operands: the ATen calls they replace are the matmuls autograd derives for nn.Linear (croco/models/blocks.py:73-112). */
int sp3_pack_bf16(const float* src, int64_t ld, int rows, int cols, void* dst, void* dstT, void* stream);
/* sp3_pack_bf16 of the 3x3 / pad 1 im2col matrix [(b, oy, ox), (ky*3 + kx)*C + c] of an NHWC fp32 map x [B, H, W, C] (stride 1 or 2; the layout
 * of sp3_im2col3x3), gathered inside the pack launch: the operand of a 3x3 convolution's bf16 GEMMs (croco/models/dpt_block.py:120-218) without
 * writing and re-reading the 9x larger column matrix. */
int sp3_pack_bf16_conv3x3(const float* x, int B, int H, int W, int C, int stride, int act, void* dst, void* dstT, void* stream);
/* sp3_pack_bf16 with the producer's activation applied to every element as it is loaded (act: 0 none, 1 exact-erf GELU = sp3_gelu, 2 ReLU =
 * sp3_relu; also the `act` of sp3_pack_bf16_conv3x3): Mlp's act(fc1(x)) -> fc2 (croco/models/blocks.py:73-79) and the ReLU -> convolution pairs of
 * the DPT head (dpt_block.py:120-142) without the activation launch and its fp32 output */
int sp3_pack_bf16_act(const float* src, int64_t ld, int rows, int cols, void* dst, void* dstT, int act, void* stream);
/* sp3_pack_bf16, and colsum[j] (+)= sum_r src[r, j] (the bias gradient of the Linear whose dY is being packed; rows <= 8192): the pack
 * launch leaves the column sums of its 64-row tiles in partial_ws (ceil(rows / 64) * ceil(cols / 64) * 64 floats), a second small launch
 * adds them in a fixed order: deterministic, and no second pass over src. */
int sp3_pack_bf16_colsum(const float* src, int64_t ld, int rows, int cols, void* dst, void* dstT, float* colsum, int accumulate,
                         float* partial_ws, void* stream);
/* torch.nn.utils.clip_grad_norm_ (croco/utils/misc.py:262-288, called with clip_grad = 1.0 by spann3r/training.py:227-228) on flat
 * gradient buckets, without a host round trip: sp3_sumsq_partial writes sp3_sumsq_blocks(n) partial sums of squares of one bucket
 * (fixed order: deterministic), sp3_clip_coef reduces `count` partials of all buckets to out[0] = extra_scale * min(1, max_norm /
 * (norm + 1e-6)) and out[1] = norm (of the gradients times |extra_scale|; max_norm <= 0: no clipping). */
int64_t sp3_sumsq_blocks(int64_t n);
/* out[j] (+)= sum_r x[r, j] for a tall fp32 matrix (the bias gradient of a Linear / convolution: autograd's dY.sum(0)), two deterministic
 * stages; scratch: sp3_colsum_rows_ws(rows, N) floats */
int64_t sp3_colsum_rows_ws(int rows, int N);
int sp3_colsum_rows(const float* x, int64_t ld, int rows, int N, float* out, int accumulate, float* scratch, void* stream);
int sp3_sumsq_partial(const float* g, int64_t n, double* partial, void* stream);
int sp3_clip_coef(const double* partial, int count, float max_norm, float extra_scale, float* out, int* step_counter, void* stream);
   /* step_counter (nullable): incremented by one -- the optimizer's step count kept on the device so that a captured step replays */
/* torch.optim.AdamW (spann3r/training.py:327) over one flat bucket of n elements (n % 1024 == 0): p, g, m, v share the element layout;
 * chunk_table holds (weight_decay, lr_scale) per 1024-element chunk (lr_scale < 0: chunk untouched -- parameters without a gradient
 * on any rank); the gradient is multiplied by grad_scale * grad_scale_dev[0] (device scalar, nullable: sp3_clip_coef's out[0]). */
int sp3_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* chunk_table, float lr, float beta1, float beta2,
                   float eps, int step, const float* grad_scale_dev, float grad_scale, const int* step_dev, const float* lr_dev, void* stream);
   /* step_dev / lr_dev (nullable): the step count / learning rate are read from device memory instead (hipGraph replay of the step) */
int sp3_softmax_bwd(const float* A, const float* dAd, const float* mask, float* dS, int64_t ld, int rows, int T, float alpha, void* stream);
/* same, and dS[:, T .. Tpad) = 0 (Tpad <= ld) */
int sp3_softmax_bwd_pad(const float* A, const float* dAd, const float* mask, float* dS, int64_t ld, int rows, int T, int Tpad, float alpha,
                        void* stream);
/* Multi-head attention in training (croco/models/blocks.py:100-108 Attention.forward, :160-166 CrossAttention.forward): the
 * reshape / permute(0, 2, 1, 3) / RoPE2D between the projections and the per-head products, and their backward, for up to three
 * tensors (q, k, v / dq, dk, dv) in ONE launch.  A part reads element (b, n, h, d) at src + b*s_b + n*s_n + h*s_h + d, rotates the
 * token by RoPE2D (pos != NULL: int64 [B*N, 2] (y, x) positions; fwd = +1: croco/models/pos_embed.py:96-157 forward, fwd = -1: its
 * transpose = the backward), and writes it to dst + b*d_b + n*d_n + h*d_h + d (dst nullable) and to the per-head transpose
 * dstT [B*H][hd][ldT] (nullable; pad columns zero), which is the W operand of the A . W^T GEMMs contracted over tokens.
 * hd <= 64, hd % 4 == 0; every stride a multiple of 4 elements, pointers 16-byte aligned. */
typedef struct sp3_head_part {
  const float* src;
  int64_t s_b, s_n, s_h;
  float* dst;
  int64_t d_b, d_n, d_h;
  float* dstT;
  const int64_t* pos;
  int32_t N;
  float fwd;
  int64_t ldT;          /* row length of dstT; columns [N, ldT) are zeroed; 0: N rounded up to 8; at most N rounded up to 64 */
} sp3_head_part;
int sp3_head_shuffle(const sp3_head_part* parts, int nparts, int B, int H, int hd, float rope_base, const float* cos_tab,
                     const float* sin_tab, int tab_len, void* stream);
   /* cos_tab / sin_tab (nullable): cos / sin of position * rope_base^(-f / (hd/4)) as [tab_len][hd / 4] fp32; positions outside the table
    * (or no table) are computed in the kernel */
/* Attention of the train-mode blocks (croco/models/blocks.py:100-108, :160-166: softmax(q k^T * scale) v per head, head_dim 64) as
 * flash-style kernels with a backward -- what torch's autograd derives for those lines, without materialising the attention matrix.
 *   forward : sp3_attention's kernel on fp32 operands (q, k already rotated; vt = per-head V^T [B*heads][64][vt_ld], zero padded to a
 *             multiple of 64 keys), also leaving lse[b*heads + h][query] = (max_j s_j, 1 / sum_j exp(s_j - max)) for the backward
 *             (two floats: one folded log-sum-exp would round a large maximum into the recomputed probabilities);
 *   backward: dq, dk, dv from (q, k, v, d out, lse) with the probabilities recomputed; two launches (16 queries / 16 keys per
 *             workgroup); the per-head transposes qT, doT [B*heads][64][ldTq] and kT [..][ldTk] (sp3_head_shuffle) feed the products
 *             contracted over tokens; lse is the forward's [B*heads][Nq][2]; D is a [B*heads][Nq] scratch: rowsum(P (.) dP) of the recomputed, rounded P and dP, so that the
 *             rows of dS sum to zero as in the unfused softmax backward (rowsum(d out (.) out) leaks a common-mode term in fp32).
 * bf16_products: 0 = exact fp32 MFMA, 1 = operands rounded to bf16 into one bf16 MFMA per product (fp32 accumulation, bf16 training).
 * Element (b, n, h, d) of q / k / v / dout / dq / dk / dv is at p + b*s + n*ld + h*64 + d (strides in elements, multiples of 4). */
int sp3_attention_train_fwd(const float* q, int64_t sq, int64_t ldq, const float* k, int64_t sk, int64_t ldk, const float* vt, int64_t vt_ld,
                            float* out, int64_t ldo, float* lse, int B, int heads, int Nq, int Nk, float scale, int bf16_products, void* stream);
typedef struct sp3_attn_bwd_desc {
  const float *q, *k, *v, *dout;
  int64_t sq, ldq, sk, ldk, sv, ldv, sdo, lddo;
  const float *qT, *kT, *doT;
  int64_t ldTq, ldTk;
  const float* lse;
  float* D;
  float *dq, *dk, *dv;
  int64_t sdq, lddq, sdk, lddk, sdv, lddv;
  int32_t B, heads, Nq, Nk;
  float scale;
  int32_t bf16_products;
} sp3_attn_bwd_desc;
int sp3_attention_train_bwd(const sp3_attn_bwd_desc* d, void* stream);
int sp3_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t ldy, const float* dx_add, int64_t ld_add,
                      float* dx, int64_t ld_dx, float* dgamma, float* dbeta, int accumulate, float* scratch, int rows, int C, float eps,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-forward geometry (demo.py:147-215).
 * sp3_focal_weiszfeld : estimate_focal_knowing_depth(pts3d, pp, focal_mode='weiszfeld') (dust3r/post_process.py:38-60):
 *   pts3d fp32 [B, H, W, 3], principal point (ppx, ppy); closed-form L2 start, `iters` (reference: 10) rounds of inverse-distance
 *   re-weighting, clipped to [focal_min, focal_max] (already multiplied by the reference's focal_base); focal [B] on the device.
 * sp3_conf_filter : the confident part of a cloud, pixel order kept (demo.py:205-211: conf_sig = (conf - 1) / conf > thresh,
 *   boolean indexing of points and colours): conf [n], pts / rgb [n, 3] (rgb nullable) -> out_pts / out_rgb [total, 3];
 *   scratch: ceil(n / 1024) ints; total: device int64. */
int sp3_focal_weiszfeld(const float* pts3d, int B, int H, int W, float ppx, float ppy, int iters, float focal_min, float focal_max,
                        float* focal, void* stream);
int sp3_conf_filter(const float* conf, const float* pts, const float* rgb, int64_t n, float thresh, int* scratch, int64_t* total,
                    float* out_pts, float* out_rgb, void* stream);

/* Camera poses from pointmaps (demo.py:170-186 calls cv2.solvePnPRansac(points, pixel grid, K, 0) per frame).  OpenCV's pipeline
 * (calib3d/src/solvepnp.cpp, ptsetreg.cpp, epnp.cpp, calibration.cpp; restated for the tests in oracle/pnp_oracle.py) is run by
 * spann3r_amd/postprocess.py::estimate_poses: the 5-point EPnP hypotheses and the 12x12 / 6x6 solves on the host, everything that
 * is O(H*W) here, one workgroup per frame (and hypothesis), double-precision sums in a fixed order.
 * Poses are Rt[12] = row-major R (9) | t (3), world -> camera.  cv_mode != 0 selects OpenCV's consensus test: projection in double,
 * no depth-sign test, (float)err^2 <= (float)thresh^2; cv_mode == 0 is the stricter test of earlier rounds (err < thresh, z > 0).
 * sp3_pnp_score    : counts[F][n_hyp] = consensus of hypothesis Rt[F][n_hyp][12] over the finite points of frame f.
 * sp3_pnp_dlt_accum: out41[F][41] = the four symmetric 4x4 blocks (10 unique entries each: S, Sx, Sy, Sr) of the calibrated DLT
 *   normal matrix [[S, 0, -Sx], [0, S, -Sy], [-Sx, -Sy, Sr]] (cvFindExtrinsicCameraParams2's L^T L) over the finite points (Rt == null)
 *   or over the consensus set of Rt[F][12], and the number of points used; norm4[F][4] = a centroid (3) and scale applied to the
 *   object points first ((0, 0, 0, 1) reproduces OpenCV's un-normalised system).
 * sp3_pnp_gn_accum : out29[F][29] = normal equations of the reprojection error (pixels) at pose Rt[F][12] (DOUBLE: the pose under
 *   refinement): H (21 unique, row-major
 *   upper triangle, parameters (omega, delta) of Xc' = Xc + omega x Xc + delta), g (6), squared error, point count -- over the
 *   consensus set of Rt_mask[F][12] if given (a FIXED set, as solvePnPRansac refines it), else over the inliers of Rt itself. */
int sp3_pnp_dlt_accum(const float* pts, int F, int H, int W, float focal, float cx, float cy, const float* norm4, const float* Rt,
                      float thresh, int cv_mode, double* out41, void* stream);
int sp3_pnp_gn_accum(const float* pts, int F, int H, int W, float focal, float cx, float cy, const double* Rt, const float* Rt_mask,
                     float thresh, double* out29, void* stream);
int sp3_pnp_score(const float* pts, int F, int H, int W, float focal, float cx, float cy, const float* Rt, int n_hyp, float thresh,
                  int cv_mode, int* counts, void* stream);

/* small utilities */
/* n (1..8) contiguous device-to-device copies in one launch; every copy 16-byte aligned and a multiple of 16 bytes.
 * (The sequence loop's per-frame bookkeeping -- torch .clone() / slice assignments around spann3r/model.py:523-531.) */
int sp3_copy_multi(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream);
int sp3_copy2d_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int cols, void* stream);
int sp3_fill_f32(float* p, float v, int64_t n, void* stream);
int sp3_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* Measurement aid (bench.py): a one-wave kernel that spins `cycles` shader clocks and stores its own duration in ticks of
 * the constant 100 MHz counter: a kernel of known length to calibrate the cost of a HIP-event bracket against. */
int sp3_spin(int64_t cycles, int64_t* ticks_100mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif
