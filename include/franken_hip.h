/* franken_hip.h — C ABI of libfranken_hip.so: the MI355X (gfx950) hot path of the
 * brainformer / GPT-2 decoder training step.
 *
 * The reference (ALVI-Labs/frankenstein) is pure Python on PyTorch: it has no FFI / plugin
 * interface, so each entry point below replaces a *PyTorch op call site* of the reference
 * (file:line given per function, relative to the reference root).  Conventions:
 *   - plain pointers + sizes only (device pointers unless stated), no torch types;
 *   - the library never allocates or frees device memory: outputs and scratch are caller
 *     allocated, scratch size comes from the matching *_workspace_bytes() query;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and returns immediately; no host synchronisation inside (graph-capturable);
 *   - return 0 on success, a negative FK_E* code otherwise; fk_last_error() returns a
 *     thread-local message.  Nothing throws across the boundary.
 *   - dtype: FK_F32 (exact-fp32 MFMA parity mode) or FK_BF16 (bf16 operands, fp32 accumulate).
 */
#ifndef FRANKEN_HIP_H
#define FRANKEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FK_VERSION 302

#define FK_OK 0
#define FK_EINVAL (-1)       /* bad shape / dtype / alignment / null pointer */
#define FK_ELAUNCH (-2)      /* hipGetLastError() after launch */
#define FK_EUNSUPPORTED (-3)

enum { FK_F32 = 0, FK_BF16 = 1 };
enum { FK_MASK_NONE = 0, FK_MASK_CAUSAL = 1, FK_MASK_BLOCK_CAUSAL = 2, FK_MASK_PREFIX = 3, FK_MASK_KEYPAD = 4, FK_MASK_DENSE = 5 };
/* fk_attn_* flags.  FK_ATTN_Q_PRESCALED: Q already holds scale * log2(e) * q (written so by fk_gemm_nt_rope's pre-scaled query
 * table), bf16 with D = 64 only: the kernels then work in the exp2 domain with the row constants (running reference maximum,
 * -LSE, -delta) as the initial MFMA accumulators, i.e. without any per-score multiply / subtract.  dQ, dK, dV are the same
 * quantities as without the flag (gradients w.r.t. the UNSCALED q, k, v).                                                  */
enum { FK_ATTN_Q_PRESCALED = 1 };
enum { FK_NORM_LAYER = 0, FK_NORM_RMS = 1 };
enum { FK_ACT_SWIGLU = 0, FK_ACT_GELU = 1 };

int fk_version(void);
const char* fk_last_error(void);

/* ---- GEMM (nn.Linear: models/brainformer.py:119-124,141-145,149,171,285,342,501; models/gpt2_model.py:35,37,56,75,82-90,133,205)
 * fk_gemm_nt: C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (+ residual[m % res_rows, n]); row-major, ld* in elements.
 *   A,B,bias,residual have `dtype`; C has `out_dtype` (= dtype, or FK_F32).  res_rows = 0 means one residual
 *   row per output row; res_rows = R broadcasts an [R, N] table over rows (the space embedding of
 *   models/brainformer.py:343).  K, lda, ldb multiples of 16 bytes.                                          */
int fk_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
               int64_t K, const void* bias, const void* residual, int64_t ldr, int64_t res_rows, int dtype,
               int out_dtype, void* stream);
/* fk_gemm_nt_rope: fk_gemm_nt (+ bias) with apply_rope (models/brainformer.py:70-91) fused into the epilogue: the first
 * rot_cols output columns (q and k of a packed q|k|v projection, heads of width D) of row m are rotated by
 * table[m / T][pos_off + m % T][(n % D) / 2] = (cos, sin)  (table_bs = 0: one cache shared by all samples).
 * The first q_cols columns (the queries; 0 = none) take their pairs from table + q_table_off floats instead: a second copy
 * of the cache multiplied by softmax_scale * log2(e), which hands Q to fk_attn_* in the FK_ATTN_Q_PRESCALED form with a
 * single rounding (models/brainformer.py:153-168: the 1/sqrt(dh) of SDPA folded into the rotation).                      */
int fk_gemm_nt_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, const void* bias, const float* table, int64_t table_bs, int64_t T, int64_t pos_off, int64_t D,
                    int64_t rot_cols, int64_t q_cols, int64_t q_table_off, int dtype, void* stream);
/* SwiGLU MLP fused into the projection epilogues (models/brainformer.py:124 and its autograd).  Hidden units use the
 * INTERLEAVED layout: for every 4 hidden units, 4 columns of h1 = w1 x followed by 4 columns of h3 = w3 x (W13 rows are
 * packed the same way, see fk_cast_pack_rows).
 *   fk_gemm_nt_swiglu : H13[M,2H] = A[M,K] * W13[2H,K]^T   and   G[M,H] = silu(h1) * h3        (one pass, G never re-read)
 *   fk_gemm_nt_dswiglu: dg = dY[M,K] * W2T[H,K]^T (never stored);  dH13[M,2H] = d(silu(h1) h3)/d(h1,h3) * dg         */
int fk_gemm_nt_swiglu(const void* A, int64_t lda, const void* W13, int64_t ldb, void* H13, int64_t ldh, void* G,
                      int64_t ldg, int64_t M, int64_t H, int64_t K, int dtype, void* stream);
int fk_gemm_nt_dswiglu(const void* dY, int64_t lda, const void* W2T, int64_t ldb, const void* H13, int64_t ldh, void* dH13,
                       int64_t lddh, int64_t M, int64_t H, int64_t K, int dtype, void* stream);
/* fk_mlp_bwd_fused (ABI 302): the two calls above that make up the data-gradient chain of the SwiGLU MLP's backward
 *   (autograd of models/brainformer.py:119-124) as ONE launch:   dg = dY W2T^T (never stored),   dH13 = SwiGLU'(H13) * dg (written once, for
 *   the weight gradients),   dX[M,D] = dH13 W13T^T with dH13 taken from registers instead of being read back.  bf16, D = 384, H % 32 == 0;
 *   W2T [H, D] and W13T [D, 2H] are the transposed shadows the two separate calls take.  Results are bit-identical to
 *   fk_gemm_nt_dswiglu followed by fk_gemm_nt (same products, operand slots and summation order).                        */
int fk_mlp_bwd_fused(const void* dY, int64_t lddy, const void* W2T, int64_t ldw2t, const void* H13, int64_t ldh, const void* W13T,
                     int64_t ldw13t, void* dH13, int64_t lddh, void* dX, int64_t lddx, int64_t M, int64_t H, int64_t D, int dtype,
                     void* stream);
/* fk_gemm_tn: C[N1,N2] (fp32) (+)= sum_m A[m,N1] * B[m,N2]  — the weight gradient dW = dY^T X of a Linear
 *   (autograd of the call sites above).  Split over m with deterministic slab reduction.                     */
size_t fk_gemm_tn_workspace_bytes(int64_t M, int64_t N1, int64_t N2, int dtype);
int fk_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N1,
               int64_t N2, int accumulate, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* fk_colsum: out[c] (+)= sum_r X[r,c]  (bias / space-embedding gradients).                                   */
size_t fk_colsum_workspace_bytes(int64_t rows, int64_t cols);
int fk_colsum(const void* X, int64_t ld, float* out, int64_t rows, int64_t cols, int accumulate, int dtype,
              void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused attention (F.scaled_dot_product_attention: models/brainformer.py:168,215; models/gpt2_model.py:64)
 * Q,K,V,O are [B, N, H, D] views: element (b,n,h,d) at base + b*bs + n*rs + h*D + d.  LSE [B,H,Nq] fp32.
 * mask: NONE | CAUSAL (k + k_off <= q + q_off) | BLOCK_CAUSAL ((k + k_off)/mask_c <= (q + q_off)/mask_c), the
 * analytic form of build_advanced_causal_mask (models/brainformer.py:93-111) incl. the [-t_q:, -t_k:] slice (:160-162).
 * PREFIX (per-sample masks of sorted token subsets, MAE's get_sub_att_matrix models/brainformer.py:392-413):
 * visible(q,k) = k < limits[b,q] <=> q >= qfirst[b,k], int32 tables from fk_prefix_mask.
 * KEYPAD (padding mask of models/simple_mae:228-236,349-352): visible(q,k) = limits[b,q] != 0 && qfirst[b,k] != 0, i.e. the two
 * int32 tables are the query / key validity flags (FK_ATTN_Q_PRESCALED kernels: the tiles in front of a sample's first zero key flag
 * take the mask-free path, same bits).  DENSE (any boolean mask: models/brainformer.py:160-168 passes whatever it is given):
 * `limits` points to uint8 [Bm, Hm, Nq, Nk] (non-zero = attend), mask_c = its batch stride and q_off its head stride in elements (0: one
 * mask for every sample / every head), k_off = 0, qfirst unused; every tile takes the per-element path of the generic kernels
 * (FK_ATTN_Q_PRESCALED is refused).
 * Fully masked rows give 0.  D in {8,16,32,64} (+128 for bf16).  Backward: dO shares O's strides, dQ/dK/dV share Q/K/V's strides; delta_ws
 * is fp32 scratch of 2 * B * H * roundup(Nq, 64) floats (the row statistics the dQ kernel hands to the dK/dV kernel).  rope_table != NULL
 * (self-attention only) additionally applies the inverse RoPE (rotation by -angle at position rope_off + index) to dQ and dK as they are
 * stored = apply_rope's backward.
 * *_dropout: the same with dropout on the attention probabilities (F.scaled_dot_product_attention(dropout_p=...) in training mode,
 * models/gpt2_model.py:64): softmax over all visible keys, then entries dropped with probability drop_p and the kept ones scaled by
 * 1 / (1 - drop_p).  No mask is stored: forward and backward regenerate it from (drop_seed[0..1] in DEVICE memory, drop_site, b, h, q, k)
 * — see fk_dropout.  drop_p == 0: identical to the plain entry points.  Generic kernels only (FK_ATTN_Q_PRESCALED is refused).            */
int fk_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t H, int64_t Nq,
                int64_t Nk, int64_t D, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs,
                int64_t v_rs, int64_t o_bs, int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off,
                const int32_t* limits, const int32_t* qfirst, float scale, int flags, int dtype, void* stream);
int fk_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                void* dQ, void* dK, void* dV, float* delta_ws, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t o_bs,
                int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off, const int32_t* limits,
                const int32_t* qfirst, float scale, const float* rope_table, int64_t rope_bs, int64_t rope_off, int flags,
                int dtype, void* stream);

int fk_attn_fwd_dropout(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t H, int64_t Nq,
                        int64_t Nk, int64_t D, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs,
                        int64_t v_rs, int64_t o_bs, int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off,
                        const int32_t* limits, const int32_t* qfirst, float scale, int flags, float drop_p, const uint32_t* drop_seed,
                        uint32_t drop_site, int dtype, void* stream);
int fk_attn_bwd_dropout(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                        void* dQ, void* dK, void* dV, float* delta_ws, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                        int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t o_bs,
                        int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off, const int32_t* limits,
                        const int32_t* qfirst, float scale, const float* rope_table, int64_t rope_bs, int64_t rope_off, int flags,
                        float drop_p, const uint32_t* drop_seed, uint32_t drop_site, int dtype, void* stream);

/* ---- dropout (nn.Dropout in training mode: models/gpt2_model.py:40,75 resid_dropout, :85,91 MLP, :129 embeddings).
 *      y[i] = (res ? res[i] : 0) + (keep(i) ? x[i] / (1 - p) : 0), n contiguous elements (multiple of 16 bytes), y may alias x.  The backward
 *      is the same call on dy without res.  keep(i) is a counter-based decision, nothing is stored:
 *        bits = mix32(mix32(mix32(hi ^ seed[0]) + seed[1] * 0x85EBCA6B + site) ^ (lo * 0x9E3779B9)),  keep <=> bits >= p * 2^32,
 *      (step and site go THROUGH a mixer round, so the masks of two sites or steps are not shifted copies of one another),
 *      mix32 = lowbias32 (x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16), (hi, lo) = the two words of i (for
 *      attention probabilities: hi = (b * H + h) * Nq + q, lo = k).  seed[0] = the run's seed, seed[1] = a step counter the caller advances
 *      once per forward — both read from DEVICE memory so that a captured graph draws a new mask on every replay; `site` numbers the
 *      dropout applications within one forward.  The stream is the library's own (torch's CPU and CUDA dropout streams differ too).     */
int fk_dropout(const void* x, const void* res, void* y, int64_t n, float p, const uint32_t* seed, uint32_t site, int dtype, void* stream);

/* ---- normalisation (nn.LayerNorm: models/brainformer.py:237,239,252,254,287,500; F.layer_norm models/gpt2_model.py:27;
 *      RMSNorm models/brainformer.py:221-232).  x,y [rows, dim] contiguous; gamma/beta fp32 (beta may be NULL);
 *      mean/rstd fp32 [rows] saved for backward (mean unused for RMS).                                        */
int fk_norm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                int64_t rows, int64_t dim, float eps, int kind, int dtype, void* stream);
/* dx = (dres ? dres : 0) + norm_bwd(dy); dgamma/dbeta (fp32, +=  when accumulate) via workspace partials.     */
size_t fk_norm_bwd_workspace_bytes(int64_t rows, int64_t dim);
int fk_norm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t dim, int kind,
                int accumulate, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- RoPE (apply_rope, models/brainformer.py:70-91): in place on the first n_rot_heads*D columns of each row of
 *      x [B, T, ld]; interleaved pairs rotated by table[(pos_off + t)] ([.., D/2, 2] fp32 = (cos, sin), i.e.
 *      torch.view_as_real of the reference's complex cache); table_bs = 0 for a shared 2-D cache, else the per-sample
 *      stride of a 3-D cache.  conj != 0 rotates by -angle (the backward).                                     */
int fk_rope(void* x, int64_t B, int64_t T, int64_t ld, int64_t nheads, int64_t D, const float* table, int64_t table_bs,
            int64_t pos_off, int conj, int dtype, void* stream);

/* ---- patch tokeniser (Rearrange 'b (t p1) c -> b (t c) p1', models/brainformer.py:282,338): x fp32 [B,T,C] ->
 *      tok [B*(T/P)*C, ldp] in `dtype`, columns >= P zero-filled (ldp >= P, multiple of 16 bytes).              */
int fk_patchify(const float* x, void* tok, int64_t B, int64_t T, int64_t C, int64_t P, int64_t ldp, int dtype,
                void* stream);

/* ---- gated / pointwise MLP activations.  SwiGLU (models/brainformer.py:124): h13 [rows, 2*H] = [w1 x | w3 x],
 *      g = silu(h1) * h3.  GELU exact erf (models/gpt2_model.py:83,89).                                        */
int fk_swiglu_fwd(const void* h13, void* g, int64_t rows, int64_t H, int dtype, void* stream);
int fk_swiglu_bwd(const void* h13, const void* dg, void* dh13, int64_t rows, int64_t H, int dtype, void* stream);
int fk_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int fk_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);

/* ---- dtype / layout plumbing for weight shadows: dst[r*ldd + c] = src[r*lds + c] (or transposed: dst[c*ldd + r]).*/
int fk_cast_pack(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int transpose,
                 int dtype, void* stream);
/* same with a row map: source row j lands in destination row (j / rblk) * rstride + j % rblk + roff (rblk = 0: identity). */
int fk_cast_pack_rows(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int transpose,
                      int64_t rblk, int64_t rstride, int64_t roff, int dtype, void* stream);
/* many fk_cast_pack_rows jobs in ONE launch: the per-step refresh of every weight shadow after the optimizer update
 * (replaces ~150 tiny launches).  `jobs` is a DEVICE array; job j owns the chunks [chunk_begin, next job's chunk_begin):
 * ceil(rows*cols / 1024) chunks of 1024 consecutive elements for a plain job, ceil(rows/32) * ceil(cols/32) tiles of 32 x 32 source
 * elements for a transposed one; total_chunks = end of the last job.                                                  */
typedef struct fk_pack_job {
  const float* src; void* dst; int64_t lds, ldd;
  int32_t rows, cols, transpose, rblk, rstride, roff;
  int64_t chunk_begin;
} fk_pack_job;
int fk_cast_pack_multi(const fk_pack_job* jobs, int64_t njobs, int64_t total_chunks, int dtype, void* stream);
int fk_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* y = a + b (same dtype) */
int fk_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);

/* fk_attn_combine: split-key attention.  A few queries against a long context (the perceiver's 32 learnable queries x 6144 tokens,
 * models/brainformer.py:204-215) fill the chip only when the KEYS are split over workgroups: the caller runs fk_attn_fwd / fk_attn_bwd on
 * views with the S key ranges folded into the batch dimension (queries replicated) and combines the partial results here.
 * parts [B, S, T, H, D]; lse_parts [B, S, H, T] or NULL.  With lse_parts (forward): out[b] = sum_s softmax_s(lse_parts)[s] * parts[b, s] and
 * lse_out [B, H, T] = log sum_s exp(lse_parts) (nullable).  Without (query gradient, already normalised by the global LSE): out = sum_s parts. */
int fk_attn_combine(const void* parts, const float* lse_parts, void* out, float* lse_out, int64_t B, int64_t S, int64_t T, int64_t H,
                    int64_t D, int dtype, void* stream);

/* ---- single-token decode step with the position on the DEVICE (`pos`: int32[1]), so one captured hipGraph serves every new token
 * of GPT.generate (models/gpt2_model.py:328-353; the reference re-forwards the whole sequence per token).
 * fk_gpt_embed_step: out[b,:] = wte[idx[b],:] + wpe[*pos,:].   fk_kv_append: kv[b, *pos, :] = qkv[b, d:3d] (kv [B, tmax, 2d]).
 * fk_attn_decode: o[b,h,:] = softmax_j(q[b,h,:] . k[b,j,h,:] * scale) v[b,j,h,:] over j = 0..*pos; k at kv + b*kv_bs + j*kv_rs + h*D,
 * v = k + H*D; strides in elements.                                                                                    */
int fk_gpt_embed_step(const int64_t* idx, const float* wte, const float* wpe, const int32_t* pos, void* out, int64_t B, int64_t dim,
                      int64_t vocab, int dtype, void* stream);
int fk_kv_append(const void* qkv, void* kv, const int32_t* pos, int64_t B, int64_t d, int64_t tmax, int dtype, void* stream);
int fk_attn_decode(const void* q, int64_t q_bs, const void* kv, int64_t kv_bs, int64_t kv_rs, void* out, int64_t o_bs, const int32_t* pos,
                   int64_t B, int64_t H, int64_t D, float scale, int dtype, void* stream);
/* fk_sample_topk: the sampling tail of GPT.generate (models/gpt2_model.py:340-351) as one launch: logits[b,:] / temperature, top-k
 * crop (values below the k-th largest -> -inf; top_k <= 0: none), softmax, ONE multinomial draw per row by inverse CDF with a
 * Philox4x32-10 uniform keyed by (*seed; *step, b).  Writes the token to cur[b] and, if out != NULL and *step < out_cols, to out[b*out_ld + *step] (out is [B, out_cols] with row
 * stride out_ld: a reused state or one graph replay too many cannot write past it; out != NULL needs 0 < out_cols <= out_ld); the last
 * block to finish sets *step += 1 and, if pos_inc != NULL, *pos_inc += 1 (so a captured decode graph advances by itself).  logits fp32,
 * `ticket` a zero-initialised uint32 scratch word owned by the caller.  top_k = 1 is the deterministic argmax (first maximum).        */
int fk_sample_topk(const float* logits, int64_t ld, int64_t B, int64_t V, float temperature, int64_t top_k, const uint64_t* seed,
                   int64_t* step, int32_t* pos_inc, int64_t* cur, int64_t* out, int64_t out_ld, int64_t out_cols, uint32_t* ticket,
                   void* stream);

/* ---- VQ-VAE tokenizer convolutions (models/vq_brain.py), channels-last [B, T, C], causal left padding dil*(K-1):
 * fk_im2col1d: cols[b, t, k, :] = x[b, t*stride + k*dil - pad, :] (zeros outside), Tout = (T-1)/stride + 1, so that
 *   CausalConv1d (:22-28) = fk_gemm_nt(cols, W') with W'[o, k*Cin + c] = W[o, c, k], and CausalConvTranspose1d(kernel 2s,
 *   stride s, :31-45) = the same with K = 2 taps and s*Cout phase-major output columns viewed as [B, s*T, Cout].
 * fk_col2im1d: adjoint of fk_im2col1d (dx from dcols; gather form, deterministic).   fk_elu_*: nn.ELU().
 * fk_argmax_rows: idx[r] = argmax_c x[r, c] (first on ties): nearest code of the cosine-similarity VQ lookup.          */
int fk_im2col1d(const void* x, void* cols, int64_t B, int64_t T, int64_t Cin, int64_t K, int64_t stride, int64_t dil, int dtype, void* stream);
int fk_col2im1d(const void* dcols, void* dx, int64_t B, int64_t T, int64_t Cin, int64_t K, int64_t stride, int64_t dil, int dtype, void* stream);
int fk_elu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int fk_elu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);
int fk_argmax_rows(const void* x, int64_t ld, int64_t* idx, int64_t rows, int64_t cols, int dtype, void* stream);

/* ---- input pipeline on device (utils/data_utils.py:115-155 process_signal, :243-267 pad_truncate_brain_list).
 * Trials are packed row-wise: trial i = rows off[i] .. off[i+1]-1 of x [total_rows, C] fp32; block[i] in [0, nblocks) is its
 * recording block.  fk_block_stats: mean / population std per (block, channel) over every row of the block's trials
 * (fp64 accumulation, deterministic; std == 0 -> 1 like :145).  fk_zscore_smooth_pad: out[i, t, :] =
 * gaussian_filter1d((x_i - mean) / std, sigma, axis 0, mode 'reflect', radius 4)[t] for t < min(len_i, Tmax), zeros after
 * (the filter sees the whole trial, truncation to Tmax comes last, as in the reference).                          */
size_t fk_block_stats_workspace_bytes(int64_t ntrials, int64_t C);
int fk_block_stats(const float* x, const int64_t* off, const int32_t* block, int64_t ntrials, int64_t C, int64_t nblocks,
                   float* mean, float* stdv, void* workspace, size_t workspace_bytes, void* stream);
int fk_zscore_smooth_pad(const float* x, const int64_t* off, const int32_t* block, const float* mean, const float* stdv,
                         float* out, int64_t ntrials, int64_t C, int64_t Tmax, double sigma, void* stream);

/* ---- MAE gather / scatter (models/brainformer.py:429-457,468,472): rows of W elements.
 * gather : dst[b, i, :] = src[b, idx[b,i] (% idx_mod), :]     (src_bs = 0 broadcasts one table: embedding / pos-emb lookup)
 * scatter: dst[b, idx[b,i], :] = src[b, i, :]                 dtypes converted on the fly; batch strides in elements.
 * fk_scatter_add_rows: table[idx[r] (% idx_mod), :] += src[r, :] (fp32 atomics; gradient of a broadcast gather).
 * fk_prefix_mask: limits[b,i] = #{j : kid[b,j]/block <= qid[b,i]/block}, qfirst[b,j] = min{i : qid[b,i]/block >= kid[b,j]/block}
 * for ascending per-sample token ids = the analytic form of the gathered block-causal sub-mask.                        */
int fk_gather_rows(const void* src, int64_t src_bs, int src_dtype, const int64_t* idx, int64_t idx_mod, void* dst, int64_t dst_bs,
                   int dst_dtype, int64_t B, int64_t n, int64_t W, int scatter, void* stream);
int fk_scatter_add_rows(const void* src, int src_dtype, const int64_t* idx, int64_t idx_mod, float* table, int64_t rows, int64_t W,
                        void* stream);
int fk_prefix_mask(const int64_t* q_ids, const int64_t* k_ids, int64_t block, int32_t* limits, int32_t* qfirst, int64_t B,
                   int64_t nq, int64_t nk, void* stream);
/* 2-D strided copy (same dtype): dst[r*ldd + c] = src[r*lds + c]. */
int fk_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype, void* stream);

/* dst[r, c] += src[r, c] (fp32, strided): accumulates a weight-gradient slab into the flat gradient arena. */
int fk_add2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream);

/* ---- GPT input embedding (models/gpt2_model.py:183-196): out[b, t, :] = (t < t_ctx ? prefix[b, t, :]
 *      : wte[idx[b, t - t_ctx], :]) + wpe[t, :];  wte/wpe are the fp32 master tables, prefix/out have `dtype`.
 *      Backward of the wte gather: dwte[idx[b, j], :] += dout[b, t_ctx + j, :] (fp32 atomics; dwte is the same
 *      buffer the tied lm_head gradient accumulates into, models/gpt2_model.py:138).                          */
int fk_gpt_embed_fwd(const int64_t* idx, const void* prefix, const float* wte, const float* wpe, void* out, int64_t B,
                     int64_t t_ctx, int64_t t_words, int64_t dim, int64_t vocab, int dtype, void* stream);
int fk_gpt_embed_bwd_wte(const int64_t* idx, const void* dout, float* dwte, int64_t B, int64_t t_ctx, int64_t t_words,
                         int64_t dim, int64_t vocab, int dtype, void* stream);

/* ---- losses.  L1 (F.l1_loss, models/brainformer.py:557) / MSE (F.mse_loss, :473), mean reduction: loss2 fp32[2] = {loss, weight sum};
 *      row_weight (nullable, one weight per row of row_len elements) gives the masked mean of models/simple_mae:393-395;
 *      bwd: dpred = grad_out[0] * d(loss)/d(pred) with grad_out a DEVICE scalar (no host sync).
 *      CE (F.cross_entropy ignore_index mean, models/gpt2_model.py:210; train_brainformer.ipynb cell 3):
 *      logits [rows, V] (ld), targets int64 [rows]; loss2 = {mean nll over valid rows, #valid}; row_lse [rows].  */
size_t fk_loss_workspace_bytes(int64_t n);
int fk_l1_loss_fwd(const void* pred, const void* target, float* loss2, int64_t n, int squared, const float* row_weight,
                   int64_t row_len, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int fk_l1_loss_bwd(const void* pred, const void* target, const float* grad_out, void* dpred, int64_t n, int squared,
                   const float* row_weight, int64_t row_len, const float* loss2, int dtype, void* stream);
size_t fk_ce_workspace_bytes(int64_t rows);
int fk_ce_loss_fwd(const void* logits, int64_t ld, const int64_t* targets, float* loss2, float* row_lse, int64_t rows,
                   int64_t V, int64_t ignore_index, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int fk_ce_loss_bwd(const void* logits, int64_t ld, const int64_t* targets, const float* row_lse, const float* loss2,
                   const float* grad_out, void* dlogits, int64_t ldd, int64_t rows, int64_t V, int64_t ignore_index,
                   int dtype, void* stream);
/* The same loss with the vocabulary processed in CHUNKS, for a head whose [rows, V] logits are never materialised (lm_head + CE of
 * models/gpt2_model.py:205-210 at V = 50257): the caller runs the head GEMM chunk by chunk (fp32 output) and folds every chunk into
 * per-row running statistics; the backward recomputes each chunk and gets its d-logits in the compute dtype (GEMM operand).
 *   fk_ce_chunk_fwd   : merge chunk [rows, cw] (first vocabulary index col0) into row_m / row_s (running max / sum exp) and pick the
 *                       target logit into row_t when the target falls into the chunk; first != 0 initialises the statistics.
 *   fk_ce_chunk_finish: row_lse = row_m + log(row_s); loss2 = {mean nll over valid rows, #valid}  (workspace: fk_ce_workspace_bytes(rows)).
 *   fk_ce_chunk_bwd   : dlogits[rows, cw] = (exp(logit - lse) - onehot) * grad_out / #valid on valid rows and columns < cw_valid, else 0. */
int fk_ce_chunk_fwd(const float* logits, int64_t ld, const int64_t* targets, int64_t col0, float* row_m, float* row_s, float* row_t,
                    int64_t rows, int64_t cw, int first, void* stream);
int fk_ce_chunk_finish(const float* row_m, const float* row_s, const float* row_t, const int64_t* targets, float* row_lse, float* loss2,
                       int64_t rows, int64_t V, int64_t ignore_index, void* workspace, size_t workspace_bytes, void* stream);
int fk_ce_chunk_bwd(const float* logits, int64_t ld, const int64_t* targets, int64_t col0, const float* row_lse, const float* loss2,
                    const float* grad_out, void* dlogits, int64_t ldd, int64_t rows, int64_t cw, int64_t cw_valid, int64_t V,
                    int64_t ignore_index, int dtype, void* stream);

/* ---- vocabulary head + cross entropy as one product each way (csrc/head_ce.hip): lm_head + F.cross_entropy of models/gpt2_model.py:205-210
 *      and the `to_words` head of the notebook CE BrainFormer (notebooks_trainer/train_brainformer.ipynb cell 3) when only the loss is
 *      needed (utils/train_utils.py:138-139).  No [rows, V] tensor in either direction.
 *   fk_head_ce_fwd : row_m / row_s / row_t (running maximum, sum of exponentials, target logit; the inputs of fk_ce_chunk_finish) of
 *                    logits = H W^T (+ bias) straight from the MFMA accumulators.  H [rows, K] (ldh), W [wrows >= V, K] (ldw, rows past V
 *                    unused), bias [>= V] or NULL in the compute dtype, targets int64 [rows].  workspace: fk_head_ce_workspace_bytes.
 *   fk_head_ce_bwd : dlT [vpad, rows_pad] (row stride lddl) = TRANSPOSED d-logits, (exp(logit - row_lse) - onehot) * grad_out[0] / loss2[1]
 *                    on rows whose target != ignore_index and vocabulary entries < V, zero elsewhere (the padding feeds GEMMs):
 *                    dH = fk_gemm_tn(dlT, W), dW = fk_gemm_nt(dlT, H^T).  dbpart (nullable): [2 * ceil(rows / 128), vpad] partial column
 *                    sums of the d-logits (bias gradient = their column sum).  rows <= rows_pad <= ceil(rows/128)*128, rows_pad % 4 == 0,
 *                    V <= vpad <= ceil(V/128)*128.
 *   fk_transpose2d : dst[c, r] = src[r, c] (H^T for the product above). */
size_t fk_head_ce_workspace_bytes(int64_t rows, int64_t V);
int fk_head_ce_fwd(const void* H, int64_t ldh, const void* W, int64_t ldw, int64_t wrows, const void* bias, const int64_t* targets,
                   float* row_m, float* row_s, float* row_t, int64_t rows, int64_t V, int64_t K, int dtype, void* workspace,
                   size_t workspace_bytes, void* stream);
int fk_head_ce_bwd(const void* H, int64_t ldh, const void* W, int64_t ldw, int64_t wrows, const void* bias, const int64_t* targets,
                   const float* row_lse, const float* loss2, const float* grad_out, void* dlT, int64_t lddl, int64_t rows_pad,
                   int64_t vpad, float* dbpart, int64_t rows, int64_t V, int64_t K, int64_t ignore_index, int dtype, void* stream);
int fk_transpose2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype, void* stream);

/* ---- optimizer: torch.optim.AdamW step fused with clip_grad_value_ (utils/train_utils.py:117-119,142-143) over a
 *      flat fp32 arena: g' = clamp(g * grad_scale, -clip, clip) (clip <= 0: no clamp); p *= 1 - lr*wd;
 *      m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).
 *      step is 1-based.  zero_grad != 0 also clears g (optimizer.zero_grad of the next step, :134).              */
int fk_adamw_step(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int64_t step, double clip, double grad_scale, int zero_grad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRANKEN_HIP_H */
