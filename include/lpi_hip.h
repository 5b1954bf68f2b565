/*
 * lpi_hip.h — C ABI of liblpi_hip.so: the MI355X (gfx950) kernels behind the LPI retrieval hot path.
 *
 * The reference (Kelvin-ywc/LPI, retrieval/) has no FFI of its own: its hot path is stock PyTorch modules
 * (SURVEY.md section 2d).  Each entry point below replaces the ATen arithmetic under one reference call site;
 * the citation after "replaces:" is the reference file:line (relative to /root/reference/retrieval/).
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (torch); no hidden allocation, no host sync;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     0 on success, a negative LPI_E* code on a rejected argument, or a positive hipError_t;
 *   - `dtype` selects the storage type of activations/weights fed to the matrix cores:
 *       LPI_F32  : f32 in, f32 accumulate (v_mfma_f32_16x16x4_f32)     — parity mode
 *       LPI_BF16 : bf16 in, f32 accumulate (v_mfma_f32_16x16x32_bf16) — throughput mode
 *     the residual stream, LayerNorm statistics, losses and prompt factors are always f32;
 *   - token rows are batch-major: row(b, l) = b * L + l (the reference permutes to [L, B, d] for
 *     nn.MultiheadAttention, models/clip/model.py:253; the arithmetic is layout independent);
 *   - matrices handed to lpi_gemm_nt are padded by the caller: M % 128 == 0, N % 128 == 0,
 *     K % (128 / sizeof(element)) == 0, all leading dimensions 16-byte aligned.
 */
#ifndef LPI_HIP_H
#define LPI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPI_F32 0
#define LPI_BF16 1
#define LPI_F16 2          /* STORAGE type of the residual stream in bf16 mode (the reference's own activation type: it runs fp16 end to
                            * end, model.py:371-392); never an MFMA operand type.  Accepted where a parameter is named x_dtype, and as
                            * lpi_gemm_nt's c_dtype together with a residual of the same type. */

#define LPI_EINVAL (-22)   /* bad shape / alignment / null pointer */
#define LPI_ENOSYS (-38)   /* combination not built */

/* epilogue selectors of lpi_gemm_nt */
#define LPI_EPI_NONE 0        /* C = alpha*acc (+bias) (+residual)                                     */
#define LPI_EPI_QUICKGELU 1   /* u = acc+bias; C = u*sigmoid(1.702u) (model.py:163-165); aux (if given) = d/du[u*sigmoid(1.702u)] — the
                               * DERIVATIVE, evaluated here from the same sigmoid and saved for the backward                          */
#define LPI_EPI_DQUICKGELU 2  /* C = acc * aux, aux = the derivative the forward saved          (backward of the activation)          */
#define LPI_EPI_LN 3          /* LayerNorm FOLDED into the GEMM: A is the LayerNorm's INPUT x, B = gamma o W (columns scaled), and
                               * C = rstd[row] * (acc - mean[row] * c1[col]) + bias[col] (alpha must be 1: LPI_EINVAL otherwise), c1[n] = sum_k B[n,k], bias = W beta + b — i.e.
                               * LN(x) W^T + b without ever writing LN(x) (model.py:172-177: ln_1 -> attn in_proj, ln_2 -> c_fc).  `residual`
                               * carries the LN operand block (f32): mean[ldr] | rstd[ldr] | c1[N] (ldr >= M, a multiple of 4). */
#define LPI_EPI_LN_QUICKGELU 4 /* ... followed by the QuickGELU epilogue (aux as for LPI_EPI_QUICKGELU).  Both: bf16 / f16 operands, shapes the
                               * persistent 256x256 kernel takes (LPI_ENOSYS otherwise: the caller runs LayerNorm + GEMM). */
#define LPI_EPI_RES_ROWSTATS 5 /* LPI_EPI_NONE on the fp16 residual stream (c_dtype LPI_F16, residual required), and the ROW STATISTICS of what it
                               * stores come out with it: `aux` is an f32 buffer of N/128 slots x 2 x ldaux (ldaux >= M, a multiple of 4) and
                               * aux[(2 j) ldaux + m] = sum, aux[(2 j + 1) ldaux + m] = sum of squares of the 128 stored (fp16-rounded) values
                               * C[m, 128 j .. 128 j + 127], summed in a fixed order.  lpi_ln_stats_finalize turns them into the mean / rstd the
                               * LayerNorm-fold epilogues of the NEXT GEMM read (model.py:172-177: x + attn(..) -> ln_2, x + mlp(..) -> the next
                               * block's ln_1), so no separate pass over the stream is needed.  Same shapes as the LN-fold epilogues. */

int lpi_version(void);   /* the C-ABI version: changes with every change of a signature or of an argument's meaning (bindings check it) */
/* number of kernels launched by this library since load (tests use it to prove the HIP path ran) */
uint64_t lpi_launch_count(void);

/* tuning knobs (speed only, never results):
 *   key 0 / 1  minimum number of 256x256 tiles for which lpi_gemm_nt uses the phased 256x256 kernel instead of the 128x128 one,
 *              for bf16 / f32 operands (defaults 1 / 1500; INT_MAX disables it);
 *   key 2      >= 0 (default 0): 2-byte-operand launches of the 256x256 kernel use its PERSISTENT form (a workgroup per CU walks its
 *              tiles; store-only epilogues: the next tile's first K-tile lands under the epilogue; epilogues that load a 2-byte
 *              residual / u tile: that tile comes to LDS by LDS-DMA; same results bit for bit); 2: store-only epilogues only;
 *              -1: one tile per workgroup everywhere;
 *   key 3      != 0 forces the two-pass attention backward where the fused single-pass kernel would be used (bf16, all four head
 *              matrices resident in LDS);
 *   key 4      row-tile group size of the 256x256 kernel's XCD-aware tile order (default 0 = 8; measured flat from 4 to 16);
 *   key 5      bf16 launches with at least 16 but fewer than this many 256x256 tiles use 256x128 tiles instead (twice the
 *              workgroups for launches that would leave the chip half empty; default 160, 0 disables it; same results bit for bit);
 *   key 6      != 0 (default): a 256x256 launch of 256k + rem tiles with rem <= 128 runs those rem tiles as 2*rem tiles of 256x128
 *              inside the same launch — one round of half tiles instead of a half-empty round (bf16; same results bit for bit);
 *   key 7      attention backward generation for 2-byte operands: 0 (default) the streamed single-pass kernel of attention4.hip where it
 *              applies and is faster (non-causal, 160 < L <= 224: the vision tower), the one-head-per-workgroup kernels of attention.hip
 *              elsewhere; 1 the kernels of attention.hip everywhere; 5 forces attention4.hip at every L it takes (same products, delta
 *              summed in another order: equal to rounding, not bit for bit).
 *   key 8      != 0: lpi_gemm_nt_grouped never groups (issues its problems one after the other; A/B switch, same bits).
 *   key 11     > 0: the streamed attention backward launches at most this many workgroups (tests: several heads per workgroup at small B H).
 *   key 12     A/B switches of the streamed attention backward (bit 0: K / V of the next head as one burst instead of spread over the head).
 *   key 13     != 0: the attention forward keeps padded (160-byte) K / V image rows where it would use the swizzled unpadded ones (Lp = 288; a TEST hook, same bits).
 *   key 14     1: the generic epilogue of the persistent GEMM everywhere (no half-width staging for store-only 2-byte outputs): a TEST hook, the bit-for-bit
 *              reference of the half-width staging (same bits).
 *   key 15     tile order of the persistent 256x256 GEMM for weights that do not fit an XCD's L2 next to the activation stream (round 5): 0 (default) = the
 *              N-tiles (an even number) of a weight above 3 MB are cut into two SLICES and the tiles run slice-major, so that each XCD keeps one slice
 *              resident instead of re-reading the whole weight every round; 2 / 3: that many slices wherever N divides; -1: off.  Same bits.
 *   key 9      > 0: minimum number of waves (<= 8) of a workgroup of the one-head attention kernels on RAGGED batches (the text tower: thousands of
 *              workgroups of a few dozen rows, bound by the latency of their staging loads; the waves beyond the row blocks only help to stage).  Same bits.
 *   key 10     reserved (0).  Returns LPI_EINVAL for a key outside 0..15. */
int lpi_set_tuning(int key, int value);
int lpi_get_tuning(int key);   /* current value of a knob (>= 0), LPI_EINVAL for a key outside 0..15 */

/* ---- a4: nn.Linear / in_proj / out_proj / c_fc / c_proj / conv1-as-matmul and every dgrad --------------
 * C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
 * replaces: models/clip/model.py:175-177 (c_fc, c_proj), :172,185 (nn.MultiheadAttention in/out proj),
 *           :215,228 (conv1 as per-patch matmul), :257 (x @ proj); prompt_learner.py:61 (@ text_projection);
 *           slinet.py:139 (logit_scale * I @ T^T); and their autograd dgrads (B = pre-transposed weight).
 * A, B: `dtype` elements.  C: `c_dtype` elements.  bias: f32 or NULL.  residual: f32 or NULL; with c_dtype == LPI_F16 (bf16
 * operands, LPI_EPI_NONE only) the residual is required and is fp16 like C — the fp16 residual stream.  aux: `dtype` or NULL. */
int lpi_gemm_nt(int dtype, int c_dtype, int M, int N, int K,
                const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                const float* bias, const void* residual, int ldr,
                int epilogue, void* aux, int ldaux, float alpha, void* stream);

/* SEVERAL such GEMMs in one launch (a GROUPED launch): `count` problems of the same operand type, output type and epilogue kind
 * (residual / aux present in all of them or in none) — e.g. the vision tower's and the text tower's c_fc of the same layer
 * (models/clip/model.py:175-177 runs them as two nn.Linear calls of two independent towers).  With count == 2 and shapes the
 * persistent 256x256 kernel takes, the second problem's tiles follow the first's in one persistent launch: they fill the first
 * problem's partial last round of CUs and run at the large kernel's rate instead of as a small launch of their own.  Any other
 * combination is issued as `count` lpi_gemm_nt calls in order — the results are the same bits either way.  `descs` is HOST memory
 * (read during the call); the pointers inside are device memory.  The launches are attributed to kernel LPI_GEMM_K_256* when grouped
 * (lpi_gemm_last_kernel). */
typedef struct lpi_gemm_desc {
    int M, N, K;
    const void* A; int lda;
    const void* B; int ldb;
    void* C; int ldc;
    const float* bias;
    const void* residual; int ldr;
    void* aux; int ldaux;
} lpi_gemm_desc;
int lpi_gemm_nt_grouped(int dtype, int c_dtype, int epilogue, float alpha, int count, const lpi_gemm_desc* descs, void* stream);
/* 1 if the last lpi_gemm_nt_grouped call of this thread ran as ONE grouped launch, 0 if it fell back to separate launches */
int lpi_gemm_last_grouped(void);

/* 1 if lpi_gemm_nt / lpi_gemm_nt_grouped take the LayerNorm-fold epilogues (LPI_EPI_LN, LPI_EPI_LN_QUICKGELU) for this operand type and shape */
int lpi_gemm_ln_supported(int dtype, int M, int N, int K);
/* mean[m] = S / d, rstd[m] = 1 / sqrt(Q / d - mean^2 + eps) from the slot sums S, Q an LPI_EPI_RES_ROWSTATS GEMM left in `part` (d / 128 slots,
 * row stride ld): what nn.LayerNorm (model.py:154-160, biased variance, eps 1e-5) computes from the row.  `_pair`: the two towers' in one launch. */
int lpi_ln_stats_finalize(int rows, int d, const float* part, int ld, float eps, float* mean, float* rstd, void* stream);
/* Guard of the ONE-SWEEP row statistics (var = E[x^2] - mean^2 in f32: lpi_ln_stats_finalize(_pair), and the out_mean / out_rstd of lpi_vis_assemble_fwd,
 * lpi_txt_embed_fwd(_varlen), lpi_prompt_add(_varlen) and the PROMPT_ADD row job).  The form loses digits as (mean / std)^2 * 1e-7; LayerNorm itself
 * (model.py:154-160) does not.  `counter` (DEVICE int32; NULL = off, the default) and `flag` are remembered for the calling HOST THREAD: every later launch
 * of those kernels from this thread adds the number of rows with mean^2 > 64 var (8 deviations: the error is ~6e-6 there) to *counter, and the row that
 * takes it from 0 to 1 also stores 1 to *flag.  `flag` (optional) may be a word of PINNED HOST memory (hipHostMalloc / torch pin_memory: the library
 * resolves its device alias with hipPointerGetAttributes and returns LPI_EINVAL for memory the device cannot write): the host then reads its own word
 * whenever it likes — no copy, no event, no synchronisation — and switches to the two-sweep statistics pass (lpi_layernorm_fwd with y = NULL);
 * lpi_amd/engine.py does so from the next forward.  A speed knob's safety net: never changes a result by itself. */
int lpi_rowstat_guard(int32_t* counter, int32_t* flag);
int lpi_ln_stats_finalize_pair(int rows0, int d0, const float* part0, int ld0, float* mean0, float* rstd0,
                               int rows1, int d1, const float* part1, int ld1, float* mean1, float* rstd1, float eps, void* stream);

/* Few-row GEMM in ONE launch (csrc/gemm_rows.hip; round 6): `count` = 1 or 2 problems (the two towers' GEMM of the same op on the pooled rows of the last
 * block / the heads: model.py:172-177,185,257, prompt_learner.py:61 on B rows per tower, and their dgrads), same operand types and epilogue kind, epilogues
 * LPI_EPI_NONE (+ f32 residual) / LPI_EPI_QUICKGELU (+ aux = gelu'(u)) / LPI_EPI_DQUICKGELU as lpi_gemm_nt.  A workgroup owns a 32 x 32 output tile over the
 * whole K range (eight waves cut K, partial tiles meet in LDS in a fixed order): no scratch, no second launch, deterministic.  M, N multiples of 32, K a
 * multiple of 64 elements (32 for f32: whole 128-byte blocks), rows 16-byte aligned: lpi_gemm_nt_rows_supported says 1 for such a shape.  Replaces rounds 2-5's split-K partial + reduction launch pairs (lpi_gemm_nt_splitk*, gone from the ABI in 604): the same
 * sums in another f32 order, half the time (profiles/r06_experiments.md section 10).  Attributed to LPI_GEMM_K_ROWS. */
int lpi_gemm_nt_rows_supported(int dtype, int M, int N, int K);
int lpi_gemm_nt_rows(int dtype, int c_dtype, int epilogue, float alpha, int count, const lpi_gemm_desc* descs, void* stream);

/* ---- a5: LayerNorm (fp32 statistics, eps 1e-5)            replaces: models/clip/model.py:154-160 ------
 * x_dtype: storage type of the residual stream x — LPI_F32, or LPI_F16 in bf16 mode (statistics and arithmetic are f32 either way).
 * fwd: y[r,:] = (x[r,:]-mean)*rstd*gamma+beta for r < rows; x `x_dtype` [rows,d] (row stride ldx), y `dtype`.
 * bwd: dx[r,:] (f32, in/out: residual-stream gradient) += LN'(dy[r,:]); optionally also writes a `cast_dtype`
 *      copy dx_cast (the next dgrad GEMM's operand).  dx == NULL: dx_cast itself is the in/out gradient stream (bf16 mode keeps
 *      no f32 copy).  dy is `dy_dtype`.  accumulate == 0: the gradient stream is WRITTEN (= LN'(dy)) instead of added to — the first
 *      backward kernel of a tower then needs no zero-fill of the [rows, d] stream. */
int lpi_layernorm_fwd(int dtype, int x_dtype, int rows, int d, const void* x, int ldx, const float* gamma, const float* beta,
                      void* y, int ldy, float* mean, float* rstd, void* stream);
int lpi_layernorm_bwd(int dy_dtype, int cast_dtype, int x_dtype, int rows, int d, const void* dy, int lddy,
                      const void* x, int ldx, const float* gamma, const float* mean, const float* rstd,
                      float* dx, int lddx, void* dx_cast, int ldcast, int accumulate, void* stream);
/* TWO LayerNorms in one launch — the vision and the text tower's LayerNorm of the same layer (two independent nn.LayerNorm calls of two
 * independent towers, model.py:154-160): the text tower's alone is a few-microsecond kernel that is mostly launch ramp.  `d` points to two
 * descriptors in HOST memory; with the bf16 / f16 operand types over the fp16 residual stream they run as one kernel, any other combination
 * as two lpi_layernorm_fwd / _bwd calls (same results either way). */
typedef struct lpi_ln_fwd_desc {
    int rows, d;
    const void* x; int ldx;
    const float* gamma; const float* beta;
    void* y; int ldy;
    float* mean; float* rstd;
} lpi_ln_fwd_desc;
typedef struct lpi_ln_bwd_desc {
    int rows, d;
    const void* dy; int lddy;
    const void* x; int ldx;
    const float* gamma; const float* mean; const float* rstd;
    float* dx; int lddx;
    void* dx_cast; int ldcast;
    int accumulate;
} lpi_ln_bwd_desc;
/* y = NULL (both problems of a pair): the row statistics only — mean / rstd are written, nothing else (fp16 stream; gamma / beta unused).  Used
 * with the LayerNorm-fold GEMM epilogues (LPI_EPI_LN), which apply the normalisation themselves; also accepted by lpi_layernorm_fwd. */
int lpi_layernorm_fwd_pair(int dtype, int x_dtype, const lpi_ln_fwd_desc* d, void* stream);
int lpi_layernorm_bwd_pair(int dy_dtype, int cast_dtype, int x_dtype, const lpi_ln_bwd_desc* d, void* stream);
/* The same backward for P rows per sample only (the first block: nothing upstream of the prompt slots is trainable, sprompt.py:230-237):
 * dy is compact [B*P, d]; x, mean/rstd and the gradient stream are the full [B*L, .] arrays, touched at rows b*L + row0 + p. */
int lpi_layernorm_bwd_rows(int dy_dtype, int cast_dtype, int x_dtype, int B, int L, int row0, int P, int d, const void* dy, int lddy,
                           const void* x, int ldx, const float* gamma, const float* mean, const float* rstd,
                           float* dx, int lddx, void* dx_cast, int ldcast, int accumulate, void* stream);
/* dst[(b*P + p), 0:cols] = src[(b*L + row0 + p), 0:cols] (`dtype` elements, 16-byte aligned rows): packs the prompt rows of a stream */
int lpi_gather_batch_rows(int dtype, int B, int L, int row0, int P, int cols, const void* src, int ld_src, void* dst, int ld_dst,
                          void* stream);

/* ---- a4: prompted multi-head attention, head_dim 64   replaces: models/clip/model.py:183-185 -----------
 * qkv: [B*L, 3*d] `dtype` (q | k | v, heads contiguous by 64).  ctx: [B*L, d] `dtype`.  lse: [B, H, L] f32.
 * softmax(q k^T / 8 + causal mask) v; causal != 0 applies the text tower's strict upper-triangular -inf mask
 * (model.py:347-353).  Backward recomputes P from lse; `delta` is [B,H,L] f32 SCRATCH: the kernels that make two passes over the scores
 * leave rowsum(dctx*ctx) there, the streamed single-pass backward (attention4.hip) keeps it in LDS and does not touch the buffer at L <= 224; at
 * 224 < L <= 288 (two key-window launches) its first launch leaves -rowsum(dctx*ctx)/8 there for the second.  The contents are unspecified after the call.
 * dqkv: [B*L, 3*d] `dtype`.  L <= 288: one workgroup per (sample, head) keeps the head's K and V in LDS (attention.hip, attention4.hip).  Round 6: NON-CAUSAL
 * UNIFORM sequences of 288 < L <= 1024 tokens (ViT-L/14@336px: 577 + prompts) run tiled over the keys with an online softmax (attn_long.hip: forward one
 * launch, backward two — dQ + delta, then dK / dV; the backward computes every row, `rows_needed` of the _prefix form is ignored there); causal or ragged
 * sequences of that length are refused with LPI_EINVAL. */
int lpi_attn_fwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, void* ctx, int ldctx,
                 float* lse, int causal, void* stream);
int lpi_attn_bwd(int dtype, int B, int L, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx,
                 const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv,
                 int causal, void* stream);
/* The same attention for ONE query row per sample — token idx[b] (NULL: token 0) — used in the LAST block, whose output is read
 * at the pooled token only (model.py:255 CLS, prompt_learner.py:61 EOT): exact dead-row elimination.  q, ctx, dctx, dq: [B, d]
 * `dtype` (row b = sample b); K and V are read from columns d..3d of qkv [B*L, 3d]; lse: [B, H] f32.  causal != 0: keys <= idx[b].
 * bwd writes dK, dV into columns d..3d of dqkv for all L rows of every sample (zeros behind the mask); columns 0..d are untouched. */
int lpi_attn_pooled_fwd(int dtype, int B, int L, int H, const void* q, int ldq, const void* qkv, int ldqkv, const int32_t* idx,
                        void* ctx, int ldctx, float* lse, int causal, void* stream);
int lpi_attn_pooled_bwd(int dtype, int B, int L, int H, const void* q, int ldq, const void* qkv, int ldqkv, const int32_t* idx,
                        const void* dctx, int lddctx, const float* lse, void* dq, int lddq, void* dqkv, int lddqkv, int causal,
                        void* stream);

/* ---- RAGGED batches (`_varlen`): samples of different lengths packed back to back --------------------------------------------
 * The text tower is causal and read at the EOT token only (model.py:347-353, prompt_learner.py:61), so the token rows BEHIND a
 * sample's own EOT can reach neither its feature nor any gradient: the reference computes them (every caption is padded to 77
 * tokens, clip.py:185-221) and throws them away.  Here a batch may be packed: sample b owns rows row_start[b] .. row_start[b+1]-1 of
 * every [rows, *] array (row_start: B + 1 int32 on the device, ascending; L_b = row_start[b+1] - row_start[b] <= L).  `L` stays the
 * MAXIMUM length (launch geometry, the [B, H, L] layout of lse / delta, the [B, L] layout of ids).  row_start == NULL is exactly the
 * function without the suffix (row(b, l) = b*L + l).  Row-wise kernels (LayerNorm, GEMM) need no variant: they see sum_b L_b rows.
 * Kernels that take a per-sample token index `idx` (lpi_pool_ln_fwd/bwd, lpi_gather_rows, lpi_scatter_rows, lpi_scatter_add_rows)
 * accept L == 0, which makes idx[b] an ABSOLUTE row index (= row_start[b] + token). */
int lpi_attn_fwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* qkv, int ldqkv, void* ctx, int ldctx,
                        float* lse, int causal, void* stream);
int lpi_attn_bwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx,
                        const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv,
                        int causal, void* stream);
/* TWO attention forwards in one launch — the vision and the text tower's of the same layer (two independent nn.MultiheadAttention calls,
 * model.py:183-185): the text tower's alone is a ~15 us chain of dependent memory round trips with the chip nearly idle.  `d`: two descriptors
 * in HOST memory (row_start may be NULL).  bf16 / f16 operands run as one kernel, f32 as two launches; same results either way. */
typedef struct lpi_attn_fwd_desc {
    int B, L, H;
    const int32_t* row_start;
    const void* qkv; int ldqkv;
    void* ctx; int ldctx;
    float* lse;
    int causal;
    int shared_rows;       /* 0, or the shared prefix of lpi_attn_fwd_shared (causal, row_start given) */
    /* LAYOUT (round 6; all 0 = the interleaved default above).  Element (row, head h, which in {q, k, v}, c) of qkv sits at row * ldqkv + h * qkv_hs +
     * which * qkv_vs + c, element (row, h, c) of ctx at row * ldctx + h * ctx_hs + c (elements; multiples of 8).  Head-BLOCKED planes [3 H][rows][64] /
     * [H][rows][64] — what the GEMMs write and read with a plane stride (lpi_gemm_nt_planes): ldqkv = ldctx = 64, qkv_hs = ctx_hs = rows * 64, qkv_vs =
     * H * rows * 64: a (sample, head) slice is one contiguous run.  2-byte types, uniform sequences (row_start = NULL, shared_rows = 0) only. */
    int qkv_hs, qkv_vs, ctx_hs;
} lpi_attn_fwd_desc;
int lpi_attn_fwd_pair(int dtype, const lpi_attn_fwd_desc* d, void* stream);
/* The same backward when only the FIRST `rows_needed` token rows of dqkv are wanted (the first block: nothing upstream of the prompt slots
 * 1 .. P is trainable, sprompt.py:230-237, so only dQ / dK / dV of rows < 1 + P are read): delta is produced for every row, the rows of dqkv
 * behind rows_needed MAY be left unwritten (the 2-byte kernels skip whole 32-row blocks behind it; rows_needed >= L is lpi_attn_bwd_varlen). */
int lpi_attn_bwd_prefix(int dtype, int B, int L, const int32_t* row_start, int rows_needed, int H, const void* qkv, int ldqkv, const void* ctx,
                        int ldctx, const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv, int causal,
                        void* stream);
/* The backward of an explicit LAYOUT (lpi_attn_fwd_desc): lay = {qkv_hs, qkv_vs, dqkv_hs, dqkv_vs, ctx_hs, dctx_hs} (elements, multiples of 8); 2-byte
 * types, non-causal, uniform sequences; rows_needed as lpi_attn_bwd_prefix (<= 0 or >= L: all rows).  Where the one-head-per-workgroup kernels run (short
 * sequences, tuning keys 3 / 7) ctx_hs and dctx_hs must be 64.  delta: [B, H, L] f32 scratch as above. */
int lpi_attn_bwd_layout(int dtype, int B, int L, int rows_needed, int H, const void* qkv, int ldqkv, const void* ctx, int ldctx, const void* dctx, int lddctx,
                        const float* lse, float* delta, void* dqkv, int lddqkv, const int32_t* lay /* HOST, 6 ints */, void* stream);
/* ONE forward in descriptor form: the single-problem call that takes the layout strides (all zero: lpi_attn_fwd_varlen / _shared) */
int lpi_attn_fwd_one(int dtype, const lpi_attn_fwd_desc* d, void* stream);
/* idx[b] stays the token index WITHIN sample b (the causal limit) */
int lpi_attn_pooled_fwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* q, int ldq, const void* qkv, int ldqkv,
                               const int32_t* idx, void* ctx, int ldctx, float* lse, int causal, void* stream);
int lpi_attn_pooled_bwd_varlen(int dtype, int B, int L, const int32_t* row_start, int H, const void* q, int ldq, const void* qkv, int ldqkv,
                               const int32_t* idx, const void* dctx, int lddctx, const float* lse, void* dq, int lddq, void* dqkv,
                               int lddqkv, int causal, void* stream);
/* The two towers' pooled-row attention (last block) as ONE launch: d[i] = the arguments of lpi_attn_pooled_fwd_varlen / _bwd_varlen (fwd uses q, qkv, idx,
 * ctx, lse; bwd q, qkv, idx, dctx, lse, dq, dqkv).  Bit for bit the two single launches. */
typedef struct lpi_attn_pooled_desc {
    int B, L, H;
    const int32_t* row_start;
    const void* q; int ldq;
    const void* qkv; int ldqkv;
    const int32_t* idx;
    void* ctx; int ldctx;
    float* lse;
    const void* dctx; int lddctx;
    void* dq; int lddq;
    void* dqkv; int lddqkv;
    int causal;
    int shared_rows;       /* 0, or the shared prefix (below): idx[b] stays the query's POSITION in sample b's sequence, L the longest sequence */
    float* shared_dkv;     /* bwd with shared_rows > 0: f32 scratch [B, shared_rows, 2 H 64] */
} lpi_attn_pooled_desc;
int lpi_attn_pooled_fwd_pair(int dtype, const lpi_attn_pooled_desc* d /* [2] */, void* stream);
int lpi_attn_pooled_bwd_pair(int dtype, const lpi_attn_pooled_desc* d /* [2] */, void* stream);
/* ONE problem in descriptor form (the only single-problem form that takes the shared-prefix fields). */
int lpi_attn_pooled_fwd_desc(int dtype, const lpi_attn_pooled_desc* d, void* stream);
int lpi_attn_pooled_bwd_desc(int dtype, const lpi_attn_pooled_desc* d, void* stream);

/* ---- The LAST block's attention WITHOUT K and V (round 6; lpi_amd/csrc/attn_stream.hip).  replaces: models/clip/model.py:183-185 for the block whose output
 * only the pooled token reads (model.py:255).  With ONE live query per (sample, head), q_h . k_l / 8 = LN1(x_l) . (W_k,h^T q_h / 8) + const and
 * sum_l p_l v_l = W_v,h (sum_l p_l LN1(x_l)) + b_v,h: the block needs two passes over the sample's rows of the residual stream (LayerNorm applied on the fly from
 * the row statistics) instead of the [B L, 2 d] K / V projection, its dgrad and the single-query attention over them.  Exact algebra.
 *   w_dtype: LPI_BF16 | LPI_F16 = type of q [B, ldq], Wqkv [3 d, ldw] (in_proj weight: rows d..2d = W_k, 2d..3d = W_v), its transpose WqkvT [d, ldwt >= 3 d]
 *   and ctx [B, ldctx]; bqkv f32 [3 d];
 *   x: fp16 [B L, ldx], the block's input rows (uniform L per sample); mean / rstd f32 [B L]: ln_1's statistics of those rows; gamma / beta f32 [d]: ln_1's affine;
 *   scratch f32 [4 B H d] (the forward writes the first half, the backward reads its first quarter and writes the second half); lse f32 [B, H].
 *   Backward: dctx bf16 [B, lddctx] -> dq bf16 [B, lddq] (gradient of the pooled queries) and dh bf16 [B L, lddh] = d LN1(x_l) of EVERY row (the K / V path's
 *   gradient; the caller adds the pooled rows' query path as before); the backward's weight operands are the BF16 weight and its transpose whatever the
 *   forward's type was.  H <= 16 heads of 64, L <= 288: lpi_spool_attn_supported. */
int lpi_spool_attn_supported(int L, int H, int d);
int lpi_spool_attn_fwd(int w_dtype, int B, int L, int H, const void* q, int ldq, const void* Wqkv, int ldw, const void* WqkvT, int ldwt, const float* bqkv, const void* x, int ldx,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, float* scratch, float* lse, void* ctx, int ldctx, void* stream);
int lpi_spool_attn_bwd(int B, int L, int H, const void* Wqkv /* bf16 */, int ldw, const void* WqkvT /* bf16 */, int ldwt, const void* x, int ldx, const float* mean,
                       const float* rstd, const float* gamma, float* scratch, const float* lse, const void* dctx, int lddctx, void* dq, int lddq, void* dh, int lddh,
                       void* stream);

/* ---- SHARED PREFIX of the text tower (round 5).  replaces nothing new: the same model.py:179-193, 347-353 and prompt_learner.py:155-163 arithmetic
 * on fewer rows.  In training every caption is [SOT][n_ctx context slots][caption tokens][EOT] and the context / deep prompts are broadcast over the
 * batch (slinet.py:119-130), so positions 0 .. n_ctx of ALL samples hold the same rows in every block: under the causal mask they attend only to each
 * other.  Layout: global rows [0, shared_rows) = those positions, stored ONCE; sample b owns rows row_start[b] .. row_start[b+1]-1 = its positions
 * shared_rows, shared_rows + 1, ... (row_start[0] = shared_rows); L = the longest sequence INCLUDING the shared positions; lse / delta hold
 * (B + 1) x H x L floats, sample index B being the shared sequence, indexed by OWN row.  Every row-wise op (LayerNorm, GEMMs) just sees fewer rows;
 * attention reads keys [shared rows | own rows].  The gradient that reaches the shared rows is the batch SUM (what lpi_rows_sum_over_batch produces
 * in the plain layout): dK / dV of the shared keys are summed over the samples in f32 in a fixed order (bitwise reproducible), so the prompt
 * gradients differ from the plain layout's only by the rounding of intermediate bf16 stores (f32: by the order of f32 sums).  All three operand types; f32
 * runs the two-pass backward kernels and ignores rows_needed. */
int lpi_txt_embed_fwd_shared(int x_dtype, int B, int L, const int32_t* row_start, int shared_rows /* = 1 + P */, int P, int d, const int64_t* ids,
                             const float* tok_emb, const float* pos, const float* ctx /* [P, d], broadcast */, void* x0, float* out_mean,
                             float* out_rstd, void* stream);
int lpi_attn_fwd_shared(int dtype, int B, int L, const int32_t* row_start, int shared_rows, int H, const void* qkv, int ldqkv, void* ctx, int ldctx,
                        float* lse, void* stream);
/* rows_needed counts POSITIONS (>= shared_rows); shared_dkv: f32 scratch [B, shared_rows, 2 H 64].  Runs the fused backward, then
 * lpi_shared_kv_reduce(accumulate = 1). */
int lpi_attn_bwd_shared(int dtype, int B, int L, const int32_t* row_start, int shared_rows, int rows_needed, int H, const void* qkv, int ldqkv,
                        const void* ctx, int ldctx, const void* dctx, int lddctx, const float* lse, float* delta, void* dqkv, int lddqkv,
                        float* shared_dkv, void* stream);
/* dqkv[key][H 64 .. 3 H 64) (+)= sum_b partial[b][key][0 .. 2 H 64) for key < shared_rows (the K and V columns of the shared rows). */
int lpi_shared_kv_reduce(int dtype, int B, int shared_rows, int H, const float* partial, void* dqkv, int lddqkv, int accumulate, void* stream);
int lpi_layernorm_bwd_rows_varlen(int dy_dtype, int cast_dtype, int x_dtype, int B, int L, const int32_t* row_start, int row0, int P, int d,
                                  const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean, const float* rstd,
                                  float* dx, int lddx, void* dx_cast, int ldcast, int accumulate, void* stream);
int lpi_gather_batch_rows_varlen(int dtype, int B, int L, const int32_t* row_start, int row0, int P, int cols, const void* src, int ld_src,
                                 void* dst, int ld_dst, void* stream);
int lpi_rows_sum_over_batch_varlen(int dtype, int B, int L, const int32_t* row_start, int row0, int P, int d, const void* dx, float* out,
                                   int accumulate, void* stream);
int lpi_prompt_add_varlen(int x_dtype, int B, int L, const int32_t* row_start, int P, int d, void* x, const float* prompt_l, long prompt_bstride,
                          float* out_mean, float* out_rstd, void* stream);
/* ids stays the padded [B, L] matrix (clip.tokenize's output); tokens l >= L_b of sample b are not embedded */
int lpi_txt_embed_fwd_varlen(int x_dtype, int B, int L, const int32_t* row_start, int P, int d, const int64_t* ids, const float* tok_emb,
                             const float* pos, const float* ctx, long ctx_bstride, void* x0, float* out_mean, float* out_rstd, void* stream);

/* ---- a1: DecomposedPrompt                          replaces: models/prompts/prompts.py:38-57 -----------
 * out[l,p,d] = scale/r * sum_r d1[l,r]*d2[p,r]*d3[d,r].  bwd: the three factor gradients from dout; scratch: Lyr*P*r floats. */
int lpi_prompt_cp_fwd(int Lyr, int P, int D, int r, const float* d1, const float* d2, const float* d3,
                      float scale, float* out, void* stream);
int lpi_prompt_cp_bwd(int Lyr, int P, int D, int r, const float* d1, const float* d2, const float* d3,
                      float scale, const float* dout, float* g1, float* g2, float* g3, int accumulate_g1,
                      float* scratch, void* stream);
/* Both prompt stacks of a DecomposedPrompt (visual and textual share dim_1_share, prompts.py:38-57) per call (round 4): the forward in ONE launch,
 * the backward in TWO (lpi_prompt_cp_bwd for the visual stack with g1 overwritten, then for the textual one with g1 accumulated: six launches).
 * The same arithmetic per value.  scratch: 2 * Lyr * P * r floats. */
int lpi_prompt_cp_fwd2(int Lyr, int P, int Dv, int Dt, int r, const float* d1, const float* d2v, const float* d2t, const float* d3v, const float* d3t,
                       float scale, float* outv, float* outt, void* stream);
int lpi_prompt_cp_bwd2(int Lyr, int P, int Dv, int Dt, int r, const float* d1, const float* d2v, const float* d2t, const float* d3v, const float* d3t,
                       float scale, const float* doutv, const float* doutt, float* g1, float* g2v, float* g2t, float* g3v, float* g3t, float* scratch,
                       void* stream);

/* ---- a3: vision front end                          replaces: models/clip/model.py:227-251 --------------
 * patchify: image [B,3,R,R] f32 -> cols [B*G*G (padded rows untouched), Kp] `dtype`, Kp >= 3*ps*ps zero padded.
 * assemble + ln_pre: x0[b] = LN([cls+pos0 ; prompts[b,0] (no pos) ; patch_emb[b]+pos1..]) -> x0 `x_dtype` [B*L,d];
 * prompt0: f32, element (b,p,:) at prompt0 + b*prompt_bstride + p*d (bstride 0 = broadcast, slinet.py:119).
 * bwd (dx0 is `dtype`: f32, or the bf16 gradient stream): applies LN' to rows 1..P of dx0 IN PLACE (dx0 is dead afterwards; the other rows' input gradients are
 *      not needed because the backbone is frozen), then dprompt[p,:] = sum_b dx0[b,1+p,:] (f32 [P,d]; the batch
 *      sum is the gradient of the training-time stride-0 broadcast, slinet.py:119).
 * out_mean / out_rstd (here, in lpi_txt_embed_fwd and in lpi_prompt_add; both or neither, NULL = not wanted): the LayerNorm statistics of every row
 *      these kernels WRITE, taken from the row as stored (a wave holds it) and written at the row's index in the stream — the statistics the NEXT
 *      LayerNorm needs (model.py:172: ln_1 of the block that reads the stream) without a pass over it; lpi_prompt_add overwrites the entries of the
 *      rows it rewrites, which a GEMM epilogue (LPI_EPI_RES_ROWSTATS) had filled from their old contents. */
int lpi_patchify(int dtype, int B, int R, int ps, const float* image, void* cols, int ldcols, void* stream);
/* The same im2col from UINT8 pixels [B,3,R,R] (what the decoder produces: PIL -> HWC bytes -> CHW) with the loader's ToTensor + Normalize
 * (utils/data.py:201-204: x / 255, (x - mean) / std) folded in through `lut` (f32 [3][256], DEVICE: lut[c][v] = the f32 value the host pipeline gives byte v
 * in channel c — filled by the caller with those very operations, so the columns equal lpi_patchify's on the normalised f32 image bit for bit).  A quarter
 * of the host-to-device bytes of the f32 batch (38.5 MB instead of 154 MB per 256 images).  ps and R multiples of 4, image 4-byte aligned. */
int lpi_patchify_u8(int dtype, int B, int R, int ps, const uint8_t* image, const float* lut, void* cols, int ldcols, void* stream);
int lpi_vis_assemble_fwd(int x_dtype, int B, int G2, int P, int d, const float* patch_emb, int ldpe, const float* cls,
                         const float* pos, const float* prompt0, long prompt_bstride,
                         const float* gamma, const float* beta, void* x0, float* mean, float* rstd, float* out_mean, float* out_rstd, void* stream);
int lpi_vis_assemble_bwd(int dtype, int B, int G2, int P, int d, void* dx0, const float* prompt0, long prompt_bstride,
                         const float* gamma, const float* mean, const float* rstd, float* dprompt, void* stream);

/* ---- a6/a7: text front end          replaces: models/clip/prompt_learner.py:52-53,128-163 --------------
 * x0[b,l] = (l in 1..P ? ctx[b,l-1] : tok_emb[ids[b,l]]) + pos[l]      (CLASS_TOKEN_POSITION == "end")
 * bwd: dctx[p,:] (+)= sum_b dx0[b,1+p,:] */
int lpi_txt_embed_fwd(int x_dtype, int B, int L, int P, int d, const int64_t* ids, const float* tok_emb, const float* pos,
                      const float* ctx, long ctx_bstride, void* x0, float* out_mean, float* out_rstd, void* stream);   /* x0 is `x_dtype` */
int lpi_rows_sum_over_batch(int dtype, int B, int L, int row0, int P, int d, const void* dx, float* out, int accumulate,
                            void* stream);   /* dx is `dtype` */

/* ---- F1: deep prompts                               replaces: models/clip/model.py:189-193 -------------
 * x[b, 1..P, :] += prompt_l[b?, p, :]   (in place on the `x_dtype` residual stream; prompt_l f32) */
int lpi_prompt_add(int x_dtype, int B, int L, int P, int d, void* x, const float* prompt_l, long prompt_bstride, float* out_mean, float* out_rstd,
                   void* stream);

/* ---- a3/a7/a2: pooled head     replaces: model.py:255-257, prompt_learner.py:57-61, slinet.py:122,133 --
 * pool_ln: y[b,:] = LN(x[b*L + idx[b], :]) (x `x_dtype`; idx NULL -> row 0 = CLS; else EOT position) -> y `dtype` [B,d]
 * pool_ln_bwd: dx (f32 [B*L,d], pre-zeroed) row idx[b] = LN'(dy[b]);
 * l2norm fwd/bwd on f32 [B,E]:  y = x/||x||. */
int lpi_pool_ln_fwd(int dtype, int x_dtype, int B, int L, int d, const void* x, const int32_t* idx, const float* gamma,
                    const float* beta, void* y, int ldy, float* mean, float* rstd, void* stream);
int lpi_pool_ln_bwd(int cast_dtype, int B, int L, int d, const float* dy, int lddy, const float* x,
                    const int32_t* idx, const float* gamma, const float* mean, const float* rstd,
                    float* dx, void* dx_cast, void* stream);
/* pooled-row gather (src `x_dtype` -> f32) / scatter (f32): dst[b] = src[b*L + idx[b]]  /  dst[b*L + idx[b]] = src[b] (+ `cast_dtype` copy; the other
 * rows of dst are the caller's, pre-zeroed).  Used to run the LAST block's MLP on the B pooled rows only: the heads read nothing
 * else of its output (model.py:255, prompt_learner.py:61), so this is exact dead-row elimination. */
int lpi_gather_rows(int x_dtype, int B, int L, int d, const void* src, const int32_t* idx, float* dst, void* stream);
int lpi_scatter_rows(int cast_dtype, int B, int L, int d, const float* src, const int32_t* idx, float* dst, void* dst_cast,
                     void* stream);
/* dst[b*L + idx[b], :] += src[b, :]  (`dtype` both; idx NULL: token 0) — adds the pooled rows' dQ contribution to d(LN1 output). */
int lpi_scatter_add_rows(int dtype, int B, int L, int d, const void* src, int ld_src, const int32_t* idx, void* dst, int ld_dst,
                         void* stream);
int lpi_l2norm_fwd(int B, int E, const float* x, int ldx, float* y, int ldy, float* inv_norm, void* stream);
int lpi_l2norm_bwd(int B, int E, const float* y, int ldy, const float* dy, int lddy, const float* inv_norm,
                   float* dx, int lddx, void* stream);
int lpi_eot_index(int B, int L, const int64_t* ids, int32_t* idx, void* stream);   /* ids.argmax(-1), prompt_learner.py:61 */

/* ---- several small row kernels of ONE dependency level in one launch (round 4) ------------------------------------------------------
 * The tail of a training step — the pooled rows of the last block, the heads, the L2 norms (model.py:255-257, prompt_learner.py:57-61,
 * slinet.py:122,133), the prompt rows (model.py:189-193, 240-248) — is a chain of few-microsecond launches on B or B*P rows, one per tower
 * and op.  lpi_row_jobs issues up to LPI_ROW_JOBS_MAX INDEPENDENT ones (the two towers' launch of the same op; independent ops of one tower) as
 * one launch: every job runs the body of its single-op kernel, so the results are bit for bit those of the entry point named below.
 * Fields a job does not use are ignored.  dt_*: LPI_F32 / LPI_BF16 / LPI_F16.
 *   POOL_LN_FWD          lpi_pool_ln_fwd(dt_b, dt_a, B, L, d, a, idx, gamma, beta, out, ld_c, mean, rstd); out2 (optional) = the gathered row itself
 *                        as f32 [B, d] (lpi_gather_rows(dt_a, B, L, d, a, idx, out2))
 *   L2NORM_FWD           lpi_l2norm_fwd(B, d, a, ld_a, out, ld_c, mean [inv_norm])
 *   L2NORM_BWD           lpi_l2norm_bwd(B, d, a [y], ld_a, b [dy], ld_b, mean_in [inv_norm], out, ld_c); out2 (optional, dt_b = BF16) = the bf16
 *                        copy of out, same row stride (lpi_cast)
 *   POOL_LN_BWD          lpi_pool_ln_bwd(dt_b, B, L, d, a [dy], ld_a, b [x], idx, gamma, mean_in, rstd_in, out, out2 [cast copy or NULL])
 *   LN_BWD               lpi_layernorm_bwd(dt_a, dt_b, LPI_F32, B, d, a [dy], ld_a, b [x f32], ld_b, gamma, mean_in, rstd_in, out [dx or NULL], ld_c,
 *                        out2 [cast], ld_c, flag [accumulate])   with (dt_a, dt_b) = (BF16, BF16) or (F32, F32)
 *   SCATTER_ADD          lpi_scatter_add_rows(dt_a, B, L, d, a, ld_a, idx, out, ld_c)
 *   GATHER_BATCH_ROWS    lpi_gather_batch_rows_varlen with d = 16-byte chunks per row and ld_a / ld_c in 16-byte units: (B, L, row_start, row0, P, a, out)
 *   PROMPT_ADD           lpi_prompt_add_varlen(dt_a, B, L, row_start, P, d, out [x, in place], a [prompt_l], bstride, mean, rstd)
 *   LN_BWD_ROWS_H16      lpi_layernorm_bwd_rows_varlen(BF16, BF16, F16, B, L, row_start, row0, P, d, a [dy compact], ld_a, b [x], ld_b, gamma, mean_in,
 *                        rstd_in, NULL, 0, out2 [bf16 gradient stream], ld_c, flag [accumulate])   (the 16-byte half-wave kernel's alignment rules)
 *   VIS_PROMPT_ROWS_BWD  the LayerNorm part of lpi_vis_assemble_bwd: LN' on rows 1..P of out (dt_a, in place) with the rows of a (prompt0, bstride)   */
enum { LPI_ROWOP_POOL_LN_FWD = 1, LPI_ROWOP_L2NORM_FWD = 2, LPI_ROWOP_L2NORM_BWD = 3, LPI_ROWOP_POOL_LN_BWD = 4, LPI_ROWOP_LN_BWD = 5,
       LPI_ROWOP_SCATTER_ADD = 6, LPI_ROWOP_GATHER_BATCH_ROWS = 7, LPI_ROWOP_PROMPT_ADD = 8, LPI_ROWOP_LN_BWD_ROWS_H16 = 9,
       LPI_ROWOP_VIS_PROMPT_ROWS_BWD = 10 };
#define LPI_ROW_JOBS_MAX 4
typedef struct lpi_row_job {
    int op;
    int B, L, d;
    int P, row0;
    int dt_a, dt_b;
    int ld_a, ld_b, ld_c;
    int flag;
    long bstride;
    const void* a; const void* b;
    const int32_t* idx; const int32_t* row_start;
    const float* gamma; const float* beta; const float* mean_in; const float* rstd_in;
    void* out; void* out2;
    float* mean; float* rstd;
} lpi_row_job;
int lpi_row_jobs(int n, const lpi_row_job* jobs, void* stream);      /* jobs: HOST array, n <= LPI_ROW_JOBS_MAX */
typedef struct lpi_rows_sum_desc {
    int B, L, row0, P, d, accumulate;
    const int32_t* row_start;
    const void* dx; float* out;
} lpi_rows_sum_desc;
int lpi_rows_sum_over_batch_pair(int dtype, const lpi_rows_sum_desc* d /* [2] */, void* stream);

/* ---- a8: symmetric contrastive loss   replaces: loss/loss.py:75-87 (ClipLoss.forward), slinet.py:139-141
 * logits f32 [n_rows, n_cols] (ld), rows r0..r0+n_local-1 are this rank's pairs; square global matrix
 * (n_rows == n_cols == W*B).  loss[0] = (CE(logits,arange)+CE(logits^T,arange))/2 over ALL rows/cols;
 * dlogits (f32, same shape) = upstream * dloss/dlogits. */
int lpi_clip_loss_fwd_bwd(int n, const float* logits, int ld, float upstream, float* loss, float* dlogits,
                          int lddl, float* row_lse, float* col_lse, void* stream);
/* Data-parallel backward of the same loss (`local_loss=False` semantics of gather_features, sprompt.py:75-80): only the nloc rows
 * r0..r0+nloc of the global matrix belong to this rank.  g[i,j] = dL/dlogits[r0+i, j], gt[i,j] = dL/dlogits[j, r0+i] (both
 * [nloc, n] row-major, row stride ldg), from the row/column log-sum-exp vectors lpi_clip_loss_fwd_bwd left behind. */
int lpi_clip_loss_local_grad(int n, const float* logits, int ld, const float* row_lse, const float* col_lse, float upstream,
                             int r0, int nloc, float* g, float* gt, int ldg, void* stream);
/* lpi_clip_loss_fwd_bwd (dlogits = NULL) + lpi_clip_loss_local_grad as TWO launches instead of four (round 4): the two log-sum-exp vectors in one
 * launch, the loss value in the first workgroup of the local-gradient kernel.  g / gt NULL: the loss and the lse vectors only.  The same arithmetic
 * per value: bit for bit the four-launch results (loss/loss.py:75-87, sprompt.py:75-80). */
int lpi_clip_loss_local(int n, const float* logits, int ld, float upstream, int r0, int nloc, float* loss, float* row_lse, float* col_lse,
                        float* g, float* gt, int ldg, void* stream);
/* `local_loss=True` form of the same loss (sprompt.py:278-283 with loss/loss.py:62-73): row-wise cross-entropy of a [rows, n] block of
 * logits (this rank's images against ALL texts, or its texts against all images) whose row i has label label0 + i (= rank * B + i).
 * loss_rows[i] = logsumexp(logits[i, :]) - logits[i, label0 + i]; dlogits (optional, may alias nothing) = upstream * (softmax - onehot).
 * lpi_sum_scaled: out[0] = scale * sum(a[i] + b[i]) (b may be NULL), fixed order — the mean of the two row-loss vectors. */
int lpi_ce_rows_fwd_bwd(int rows, int n, const float* logits, int ld, int label0, float upstream, float* loss_rows, float* dlogits,
                        int lddl, void* stream);
int lpi_sum_scaled(int n, const float* a, const float* b, float scale, float* out, void* stream);
/* dst[r, 0:cols] = src[r, 0:cols], f32, row strides lds / ldd elements (packing / zero-padded operands; no framework copy kernels) */
int lpi_zero(void* ptr, long bytes, void* stream);   /* hipMemsetAsync(ptr, 0, bytes) on `stream` */
int lpi_copy_rows(int rows, int cols, const float* src, long lds, float* dst, long ldd, void* stream);
/* a11 task-id selection, replaces get_visual_task_id / get_textual_task_id (methods/sprompt.py:336-368):
 * sel[i] = argmin_t min_c sum_e |feat[i,e] - keys[t,c,e]|, keys [T, C, E] f32 (the KMeans centres of every task seen so far);
 * dist (optional) [n, T] receives the per-task minima. */
int lpi_l1_task_id(int n, int E, int T, int C, const float* feat, int ldf, const float* keys, int32_t* sel, float* dist,
                   void* stream);
/* ---- a10/f3: the task keys — KMeans(n_clusters=5, random_state=0).fit(features)   replaces: methods/sprompt.py:370-397 (clustering), round 4 -----------
 * The device parts of scikit-learn's fit (k-means++ seeding + Lloyd iterations, sklearn/cluster/_kmeans.py), driven by lpi_amd/kmeans.py, which keeps the
 * random draws (numpy RandomState) and the convergence logic on the host on vectors of n floats at most: the features X [n, E] f32 stay on the device.
 *   lpi_kmeans_sqdist   out[c, i] = |x_i - x_cand[c]|^2, c < nc (the distances of every point to candidate centres, which are points)
 *   lpi_kmeans_assign   labels[i] = argmin_c |x_i - centers[c]|^2 (first minimum; labels in / out), *changed = 1 if any label changed (never cleared here);
 *                       mindist (NULL = not wanted) [n]: that minimum — what scikit-learn's empty-cluster relocation ranks the points by
 *   lpi_kmeans_update   new_centers[c] = mean of the points labelled c (0 for an empty cluster), counts[c] = their number; fixed summation order
 *   lpi_kmeans_colstats colsum[e] = sum_i (x[i, e] - center[e]), colsq[e] = sum_i (x[i, e] - center[e])^2, center NULL = 0 (the tolerance mean_e var_i x[i, e] * tol:
 *                       column means from a first call, centred squares from a second — two passes, no cancellation) */
int lpi_kmeans_sqdist(int n, int E, int nc, const float* X, int ldx, const int32_t* cand, float* out, void* stream);
int lpi_kmeans_assign(int n, int E, int k, const float* X, int ldx, const float* centers, int32_t* labels, int32_t* changed, float* mindist, void* stream);
int lpi_kmeans_update(int n, int E, int k, const float* X, int ldx, const int32_t* labels, float* new_centers, float* counts, void* stream);
int lpi_kmeans_colstats(int n, int E, const float* X, int ldx, const float* center, float* colsum, float* colsq, void* stream);

/* a10 optimiser step, replaces optim.SGD(momentum, weight_decay).step() (methods/sprompt.py:253,311) on one flat f32 vector:
 * d = grad + wd*p; buf = first ? d : momentum*buf + d; p -= lr*buf. */
int lpi_sgd_step(long n, float* param, const float* grad, float* momentum_buf, float lr, float momentum, float weight_decay,
                 int first, void* stream);

/* ---- a8: alignment loss                                   replaces: models/slinet.py:143-158 ------------
 * v = mean_d(vis)/temp, t = mean_d(txt)/temp ([Lyr,P]); S = v t^T [Lyr,Lyr]; loss[0] = weight * ClipLoss(S);
 * dvis [Lyr,P,Dv], dtxt [Lyr,P,Dt] = dense gradients of loss[0] (both NULL: forward only). */
int lpi_align_loss_fwd_bwd(int Lyr, int P, int Dv, int Dt, const float* vis, const float* txt, float temp,
                           float weight, float* loss, float* dvis, float* dtxt, void* stream);
/* The same loss and gradients in TWO launches instead of three (round 4): the row means go to `scratch` (2 * Lyr * P floats), and every workgroup of the
 * second launch recomputes the Lyr x Lyr core from them and writes its own gradient row (workgroup 0 the loss).  Value for value lpi_align_loss_fwd_bwd. */
int lpi_align_loss_fwd_bwd2(int Lyr, int P, int Dv, int Dt, const float* vis, const float* txt, float temp, float weight, float* loss, float* dvis,
                            float* dtxt, float* scratch, void* stream);

/* ---- a9: task loss (nt_bxent over the tasks' flattened prompts)   replaces: loss/loss.py:6-33, models/slinet.py:167-183 ----
 * X f32 [T, D] (row t = prompts of task t, flattened), target int32 [T,T] (task_sim > 0.4), T <= 32.  loss[0] = weight * nt_bxent(X);
 * dx_row [D] = d loss / d X[row,:] (only the current task's prompts train; NULL or row < 0: forward only); accumulate != 0: ADDED to what dx_row
 * holds (the fused training step adds the task term onto the alignment gradient already sitting in the towers' prompt-gradient buffers).
 * scratch: 2*T*T floats. */
int lpi_nt_bxent_fwd_bwd(int T, int D, int row, const float* X, const int32_t* target, float temp, float weight,
                         float* loss, float* dx_row, int accumulate, float* scratch, void* stream);

/* ---- misc ---------------------------------------------------------------------------------------------- */
int lpi_cast(int src_dtype, int dst_dtype, long n, const void* src, void* dst, void* stream);
int lpi_transpose(int dtype, int rows, int cols, const void* src, int lds, void* dst, int ldd, void* stream);
/* two f32 transposes in one launch (the two feature matrices of the contrastive loss's gradient GEMMs) */
int lpi_transpose2(int dtype, int rows0, int cols0, const void* src0, int lds0, void* dst0, int ldd0,
                   int rows1, int cols1, const void* src1, int lds1, void* dst1, int ldd1, void* stream);
/* per-row descending rank of the best ground-truth column (itm_eval, methods/sprompt.py:558-599):
 * rank[i] = #{j : s[i,j] > s[i,gt] or (s[i,j] == s[i,gt] and j > gt)} minimised over gt in gt_list row i.
 * (np.argsort(score)[::-1] places later indices first among ties.) */
int lpi_retrieval_rank(int n_rows, int n_cols, const float* scores, int ld, const int32_t* gt, int gt_per_row,
                       int32_t* rank, void* stream);
int lpi_topk(int n_rows, int n_cols, int k, const float* scores, int ld, int32_t* idx, float* val, void* stream);

/* ---- a6 (host side): CLIP byte-level BPE      replaces: models/clip/simple_tokenizer.py:62-132, clip.py:185-221 ------
 * HOST functions (no GPU work, no stream).  create: `merges_utf8` is the decompressed text of bpe_simple_vocab_16e6.txt(.gz) —
 * third-party data that is not shipped with this library; returns NULL on a malformed table.  encode: pattern split + byte mapping
 * + pair merging of ONE cleaned, lower-cased UTF-8 string (cleaning — ftfy / html.unescape / whitespace — stays with the caller);
 * writes at most max_ids ids and returns the full count.  tokenize = clip.tokenize: row t of out [n, context_length] int64 is
 * SOT, ids, EOT, zero padding; returns 0, or t+1 for the first text that does not fit when truncate == 0 (clip.py:218 raises). */
/* ---- f4 (optional): low-rank cross-modal interaction of the prompt rows, "InteractModule" ------------------------------------------
 * replaces: InteractModule.forward (grounding/maskrcnn_benchmark/modeling/bert/modeling_bert.py:616-651; parameters :558-590) and its backward.
 * xv [N, Dv], xt [N, Dt] f32: the visual / textual prompt rows of the batch ([bs, P, D] flattened).  Per direction the rank-r CP weights
 * d1 [Lyr, R], d2 [Din + 1, R] (last row = bias), d3 [Dout, R]:  new = x M[:Din] + M[Din],  M[i, j] = mean_r d1[layer, r] d2[i, r] d3[j, r];
 * out_v = LayerNorm_v((1 - mix) xv + mix new_v), out_t likewise (mix = 0.1, eps = 1e-5 in the reference).  Dv, Dt <= 1024, R <= 8.
 * stat: [4, N] f32 (mean / rstd of the two LayerNorms).  The backward recomputes the forward; dxv / dxt are overwritten; `grads` receives
 *   [v2t: dd3 [Dt R] | dd2 [(Dv + 1) R] | dd1 row `layer` [R] | dgamma_t [Dt] | dbeta_t [Dt]] [t2v: dd3 [Dv R] | dd2 [(Dt + 1) R] | dd1 row [R] | dgamma_v [Dv] | dbeta_v [Dv]];
 * workspace: lpi_interact_workspace_floats(N, Dv, Dt, R) floats (per-workgroup partial sums, added in a fixed order). */
int lpi_interact_workspace_floats(int N, int Dv, int Dt, int R);
int lpi_interact_fwd(int N, int Dv, int Dt, int R, int Lyr, int layer, const float* xv, int ldv, const float* xt, int ldt,
                     const float* d1_v2t, const float* d2_v2t, const float* d3_v2t, const float* d1_t2v, const float* d2_t2v,
                     const float* d3_t2v, const float* gamma_v, const float* beta_v, const float* gamma_t, const float* beta_t, float mix,
                     float eps, float* out_v, int ldov, float* out_t, int ldot, float* stat, void* stream);
int lpi_interact_bwd(int N, int Dv, int Dt, int R, int Lyr, int layer, const float* xv, int ldv, const float* xt, int ldt,
                     const float* d1_v2t, const float* d2_v2t, const float* d3_v2t, const float* d1_t2v, const float* d2_t2v,
                     const float* d3_t2v, const float* gamma_v, const float* beta_v, const float* gamma_t, const float* beta_t, float mix,
                     float eps, const float* g_out_v, int ldgv, const float* g_out_t, int ldgt, float* dxv, int lddv, float* dxt, int lddt,
                     float* grads, float* workspace, void* stream);

/* ---- host side: batch assembly for the input pipeline      replaces: the DataLoader's default_collate (torch.stack of the decoded images) + the pageable
 * `images.cuda()` copy of methods/sprompt.py:166-167, 301.  HOST pointers, no stream: dst[i * bytes_each ..] = srcs[i][0 .. bytes_each) for i < n on
 * `threads` host threads (dst: the pinned staging buffer the H2D DMA reads; 154 MB of f32 pixels per 256-pair step).  lpi_amd/pipeline.py. */
int lpi_host_gather(void* dst, const void* const* srcs, int n, long bytes_each, int threads);

void* lpi_bpe_create(const char* merges_utf8, long nbytes);
void lpi_bpe_destroy(void* handle);
int lpi_bpe_encode(void* handle, const char* text_utf8, int32_t* ids, int max_ids);
int lpi_bpe_tokenize(void* handle, const char* const* texts, int n, int context_length, int truncate, int64_t* out);

/* Which kernel the calling thread's last lpi_gemm_nt / lpi_gemm_nt_rows launched (-1: none yet): measurement tools attribute a
 * launch to a kernel with this instead of re-deriving the dispatch rules. */
#define LPI_GEMM_K_128 0        /* gemm_nt_kernel, 128x128 tiles                                  */
#define LPI_GEMM_K_256 1        /* gemm256_kernel, 256x256 tiles                                  */
#define LPI_GEMM_K_256_TAIL 2   /* gemm256_tail_kernel: 256x256 tiles, short last round as halves */
#define LPI_GEMM_K_256X128 3    /* gemm256x128_kernel                                             */
#define LPI_GEMM_K_ROWS 4       /* gemm_rows_kernel: few-row GEMM, 32x32 tiles over the whole K   */
int lpi_gemm_last_kernel(void);


#ifdef __cplusplus
}
#endif
#endif /* LPI_HIP_H */
