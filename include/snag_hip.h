/* libsnag_hip.so -- C ABI of the MI355X (gfx950) fusion + negative-sampled scoring hot path.
 *
 * The reference (zjukg/SNAG) is pure Python/PyTorch and has no FFI: the path sits behind
 * nn.Module / free-function interfaces.  Each entry point below names the reference interface
 * (file:line under /root/reference/SNAG_MMEA) whose device work it replaces; INTEGRATION.md shows
 * the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions (SURVEY.md par. 8b):
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense
 *     row-major fp32; index arrays are int32 or int64 (idx_bytes = 4 or 8);
 *   - the caller owns every buffer, including workspaces and saved-for-backward tensors;
 *     the library never allocates, frees or synchronises: all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value 0 = enqueued, negative = refused (nothing enqueued); snag_last_error()
 *     returns a thread-local description of the last refusal on this thread.
 */
#ifndef SNAG_HIP_H
#define SNAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNAG_OK 0
#define SNAG_ERR_ARG (-1)         /* null pointer / negative size / inconsistent shapes */
#define SNAG_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define SNAG_ERR_LAUNCH (-3)      /* HIP refused the launch (hipGetLastError text in snag_last_error) */

/* Bumped whenever a struct layout or an entry's argument list changes: snag_amd/_lib.py refuses a library whose snag_version()
 * differs from the constant it was written against (a stale or variant .so would otherwise be read with the wrong field offsets). */
#define SNAG_ABI_VERSION 3

#if defined(__GNUC__)
#define SNAG_API __attribute__((visibility("default")))
#else
#define SNAG_API
#endif

SNAG_API int snag_version(void);
SNAG_API const char* snag_last_error(void);

/* ------------------------------------------------------------------------------------------
 * a7  Negative-sampled margin scoring.
 * Replaces the device work of MSNEA.kge term 1, model/MSNEA.py:110-122 with r_rep :151-152
 * (embedding gather, F.normalize, h + r - t, squared L2, relu(margin + pos - neg), sum), in the
 * MKGC shape named by BASELINE.json: one positive (h, r, t) against n_neg negative tails.
 *
 *   s(h,r,x) = || nrm(ent[h]) + nrm(rel[r]) - nrm(ent[x]) ||^2 ,  nrm(v) = v / max(||v||, 1e-12)
 *   pos[b]       = s(h[b], r[b], t[b])
 *   neg[b,k]     = s(h[b], r[b], neg_t[b,k])
 *   loss_part[b] = sum_k relu(margin + pos[b] - neg[b,k])          (optional, may be NULL)
 * neg_t == NULL with n_neg > 0 means "candidates 0..n_neg-1 for every triple" (full-table ranking).
 *
 * ent [n_ent, dim] with row stride ent_ld floats (>= dim; lets the scorer read the read-out slot
 * hidden[:, g, :] of the fusion output in place), rel [n_rel, dim] dense; h, r, t [batch];
 * neg_t [batch, n_neg]; dim <= 512, dim % 4 == 0 and ent_ld % 4 == 0 for the forward.
 * Indices are trusted to be in range (as torch's embedding gather on device is).
 */
SNAG_API int snag_score_margin_fwd(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin,
                          float* pos, float* neg, float* loss_part, void* stream);

/* Same scores from bf16 tables (raw bfloat16 bit patterns), fp32 arithmetic and outputs: the storage format of
 * BASELINE config 5 (|E| = 1M, dim 256, 128 negatives: a 512 MB table that does not fit the Infinity Cache).
 * dim and ent_ld (in elements) multiples of 8, dim <= 1024. */
SNAG_API int snag_score_margin_fwd_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin,
                          float* pos, float* neg, float* loss_part, void* stream);

/* Backward of the above (replaces autograd of model/MSNEA.py:110-122).
 * Upstream gradient, one of:
 *   - hinge mode (dpos == dneg == NULL): d loss / d neg[b,k] = -grad_scale * [margin + pos - neg > 0],
 *     d loss / d pos[b] = -sum_k of those; the scores are recomputed in-kernel;
 *   - explicit mode: dpos [batch], dneg [batch, n_neg] given (grad_scale multiplies them).
 * d_ent [n_ent, dim] (row stride d_ent_ld) and d_rel [n_rel, dim] are ACCUMULATED into with fp32
 * atomics (caller zeroes them).
 */
SNAG_API int snag_score_margin_bwd(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          const float* dpos, const float* dneg,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, void* stream);

/* Training form of the hinge scorer: forward AND backward in one launch (replaces MSNEA.kge term 1 model/MSNEA.py:110-122 together
 * with its autograd, and the scalar sum :122).  Per triple the K+3 table rows are gathered ONCE; the launch writes
 *   pos [batch], neg [batch, n_neg], loss_part [batch]           exactly as snag_score_margin_fwd does,
 *   loss[0] = sum_b loss_part[b]  (optional; summed in a fixed order by the block that finishes last: bit-reproducible)
 * and ACCUMULATES grad_scale * d(sum_b loss_part[b]) / d ent, d rel into d_ent (row stride d_ent_ld) / d_rel with fp32 atomics
 * (caller zeroes them), exactly as snag_score_margin_bwd does in hinge mode.
 * loss != NULL needs loss_part and `ticket`: SNAG_SCORE_TICKET_WORDS uint32 the caller zeroes ONCE (every launch leaves them
 * zero again; launches that share a ticket buffer must be stream-ordered).
 * neg_t must be given when n_neg > 0 (no implicit candidates here). */
#define SNAG_SCORE_TICKET_WORDS 1056
SNAG_API int snag_score_margin_fwd_bwd(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          float* pos, float* neg, float* loss_part, float* loss, uint32_t* ticket,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, void* stream);

/* The training form WITHOUT floating-point atomics (deterministic, and past the ~1.3 TB/s memory-side atomic rate that bounds
 * snag_score_margin_fwd_bwd): pass A, one wave per triple, writes scores / losses, the 2*batch rows q_b = h^ + r^ and dq_b, and one
 * 8-byte (row, weight) entry per touched destination; pass B, one 16-lane group per entity / relation, sums its entries' rows in
 * 64-bit fixed point (order-free: bit-reproducible), applies the backward of F.normalize once and WRITES the gradient row.
 * Same outputs as snag_score_margin_fwd_bwd, except that d_ent / d_rel are OVERWRITTEN, not accumulated into: with
 * write_untouched = 1 every row of d_ent [n_ent, dim] and d_rel is written (zeros where nothing arrived: no fill launch needed),
 * with 0 only the rows of the batch's own heads, tails and negative tails are written (a large, otherwise clean buffer: config 5).
 * |grad_scale * 2 * n_neg * (occurrences of one entity in the batch)| must stay below 2^29 (fixed point with 32 fractional bits).
 * work: snag_score_seg_work_bytes(...) bytes, 256-byte aligned, zeroed ONCE by the caller (every launch leaves it ready for the
 * next); dim % 4 == 0, 16-byte aligned rows; launches sharing a workspace must be stream-ordered.  The summed loss is formed by one
 * extra workgroup of pass B (fixed order); `ticket` is unused since round 3 and may be NULL. */
SNAG_API int64_t snag_score_seg_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim);
SNAG_API int snag_score_margin_fwd_bwd_seg(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          float* pos, float* neg, float* loss_part, float* loss, uint32_t* ticket,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, int write_untouched,
                          void* work, int64_t work_bytes, void* stream);

/* snag_score_margin_fwd_bwd_seg when the entity table `ent` [n_ent, dim] (dense) IS the output of a LayerNorm -- the Mformer's read-out
 * rows, model/SNAG_tools.py:218-222 -- and the gradient is wanted on the LayerNorm's INPUT: the 16-lane group of pass B that has just formed
 * an entity's gradient row g applies the LayerNorm backward to it in registers, so the row never travels through memory:
 *   dt[n]  = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)),  dxhat = g gamma,  xhat = (t[n] - mean) rstd     (every row written)
 *   dtm[n] = dt[n] * keep[n] * keep_scale                         (NULL keep / dtm: not formed)
 *   part[ceil(n_ent / 16)][3][dim]: per 16 rows, the column sums of g xhat (dgamma), g (dbeta) and dtm-or-dt (the bias gradient of the dense
 *   layer in front of the LayerNorm); the caller reduces the slab over its rows (snag_fusion_bwd does, snag_fusion_bwd_desc.ln_bwd_done).
 * The relation gradient d_rel and all scores / losses are those of snag_score_margin_fwd_bwd_seg; d_ent is not formed. */
typedef struct snag_ln_bwd_tail {
    const float* t;          /* [n_ent, dim] pre-LayerNorm rows */
    const float* stats;      /* [n_ent, 2] (mean, rstd) of the forward */
    const float* gamma;      /* [dim] */
    const uint8_t* keep; int64_t keep_ld; float keep_scale;
    float* dt;               /* [n_ent, dim] */
    float* dtm;              /* [n_ent, dim] or NULL */
    float* part;             /* [ceil(n_ent / 16), 3, dim] */
} snag_ln_bwd_tail;
SNAG_API int snag_score_margin_fwd_bwd_seg_ln(const float* ent, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          float* pos, float* neg, float* loss_part, float* loss, float* d_rel,
                          void* work, int64_t work_bytes, const snag_ln_bwd_tail* tail, void* stream);

/* The segmented form for tables far LARGER than a batch's footprint (BASELINE configs[4]: 1M entities, 134k entries per batch): pass A
 * writes one record per entry and -- for a negative tail, whose row and query it holds in registers -- the destination's gradient row as if
 * that entry were its only one (round 5); pass B walks the records, at most batch * (n_neg + 3) of them, one LANE each, instead of all
 * n_ent + n_rel counters, and redoes only the destinations that received more.  Only touched rows of d_ent are written (overwritten, not
 * accumulated into): the caller keeps the rest of that buffer zero.  EVERY row of d_rel is written (zeros where no triple of the batch names
 * the relation): the caller does not clear it.  Hot rows of skewed index
 * distributions: entries beyond a destination's 16 inline slots go to an open-addressed hash of 64-entry chunks keyed (dest, chunk)
 * (wait-free insertion, one home per entry) and are summed in 64-bit fixed point -- bit-reproducible whatever the arrival order.
 * Same scores / losses / gradients as snag_score_margin_fwd_bwd_seg.  _bf16: tables of raw bfloat16 bit patterns (8-byte aligned rows),
 * fp32 arithmetic and gradients.  work: snag_score_touched_work_bytes(...) bytes, 256-byte aligned, zeroed ONCE by the caller. */
SNAG_API int64_t snag_score_touched_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim);
SNAG_API int snag_score_margin_fwd_bwd_touched(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          float* pos, float* neg, float* loss_part, float* loss,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, void* work, int64_t work_bytes, void* stream);
SNAG_API int snag_score_margin_fwd_bwd_touched_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          float* pos, float* neg, float* loss_part, float* loss,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, void* work, int64_t work_bytes, void* stream);

/* The same three training entries over bf16 TABLES (raw bfloat16 bit patterns; BASELINE config 5: |E| = 1M, dim 256, 128 negatives):
 * fp32 arithmetic, fp32 scores, fp32 gradient accumulation -- only the gathered rows are 2-byte.  d_ent / d_rel stay fp32. */
SNAG_API int snag_score_margin_fwd_bwd_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          float* pos, float* neg, float* loss_part, float* loss, uint32_t* ticket,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, void* stream);
SNAG_API int snag_score_margin_bwd_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          const float* dpos, const float* dneg,
                          float* d_ent, int64_t d_ent_ld, float* d_rel, void* stream);
SNAG_API int snag_score_margin_bwd_fixed_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          const float* dpos, const float* dneg, long long* acc_ent, long long* acc_rel, void* stream);

/* Config-5 optimiser pieces: torch.optim.AdamW on a slice of fp32 MASTER weights (the rows of the entity table this rank owns,
 * SURVEY par. 8e "row-sharded ownership of the entity table + optimizer state") that also writes the bf16 copy of the updated weights
 * (bf16_out, optional) -- the slice of the replicated bf16 table the scorer gathers from, all-gathered afterwards.  n % 4 == 0.
 * snag_f32_to_bf16: round-to-nearest-even conversion of a whole table (initialisation). */
SNAG_API int snag_adamw_step_bf16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                  const uint64_t* tick, float grad_scale, uint16_t* bf16_out, void* stream);
SNAG_API int snag_f32_to_bf16(const float* src, int64_t n, uint16_t* dst, void* stream);
/* Clear exactly the rows of an [n_ent, dim] gradient buffer that one batch scatter-added into (h[b], t[b], neg_t[b, :]): 136 MB of
 * stores per 1,024 x 130 rows at dim 256 instead of a 1 GB memset of the whole config-5 buffer.  dim % 4 == 0. */
SNAG_API int snag_zero_touched_rows(float* d_ent, int64_t d_ent_ld, int dim, const void* h, const void* t, const void* neg_t, int idx_bytes,
                                    int64_t batch, int n_neg, void* stream);

/* Bit-reproducible variant of snag_score_margin_bwd: the same gradients accumulated with 64-bit INTEGER atomics in fixed point
 * (32 fractional bits: sums are associative, so the result does not depend on the order the triples arrive in; |sum| < 2^31,
 * resolution 2.3e-10).  acc_ent [n_ent, dim] and acc_rel [n_rel, dim] are dense int64 accumulators the caller zeroes;
 * snag_fixed_to_float then writes (or adds) them into fp32 gradient rows with stride dst_ld. */
SNAG_API int snag_score_margin_bwd_fixed(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          const float* dpos, const float* dneg, long long* acc_ent, long long* acc_rel, void* stream);
SNAG_API int snag_fixed_to_float(const long long* acc, int64_t rows, int dim, float* dst, int64_t dst_ld, int accumulate, void* stream);

/* Optional loss epilogue over the scores of snag_score_margin_fwd: self-adversarial negative sampling (SURVEY par. 8 a7; the only
 * in-tree traces are the unused flags --adv_temp / --emb_dim, config.py:53-54, so PARITY IS UNPINNED: oracle/ref_cpu.py selfadv_loss
 * restates the published formula).  Lower score = more plausible:
 *   p_k = softmax_k(-adv_temp * neg[b,k]) (constant),  loss_part[b] = -log sigmoid(margin - pos[b]) - sum_k p_k log sigmoid(neg[b,k] - margin)
 *   dpos[b] = sigmoid(pos[b] - margin), dneg[b,k] = -p_k sigmoid(margin - neg[b,k])   (either may be NULL); feed them to
 *   snag_score_margin_bwd in explicit mode. */
SNAG_API int snag_selfadv_loss(const float* pos, const float* neg, int64_t batch, int n_neg, float margin, float adv_temp,
                               float* loss_part, float* dpos, float* dneg, void* stream);

/* MKGC link prediction (the "Hits@1/10" half of BASELINE.json's metric; PARITY UNPINNED: the MKGC eval is not in the
 * reference tree).  Score every entity as a candidate tail by calling snag_score_margin_fwd with neg_t == NULL and
 * n_neg == n_ent (candidate k = entity k), then
 *   rank[b] = #{ n : scores[b,n] < scores[b, t[b]] and filter[b,n] == 0 }
 * (filter: optional uint8 [batch, n_ent] marking the other known true tails of (h,r): the "filtered" setting). */
SNAG_API int snag_rank_full_table(const float* scores, int64_t batch, int64_t n_ent, const void* t, int idx_bytes,
                                  const uint8_t* filter, int32_t* rank, void* stream);

/* Deterministic (fixed-order) sum of n floats into out[0]; used for the scalar loss. */
SNAG_API int snag_reduce_sum(const float* x, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 contraction on the matrix cores.  fp32 in, fp32 out, fp32 accumulation; the multiplies run as split-bf16 MFMA:
 * every operand value is split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, a = h + m + l) in registers and a
 * product a*b is the sum of the six bf16 x bf16 products of weight >= 2^-16 (v_mfma_f32_32x32x16_bf16; the three dropped ones are
 * below 2^-23 |a*b|, the size of fp32's own product rounding).  On gfx950 the bf16 matrix rate is 16x the f32 one
 * (v_mfma_f32_32x32x2_f32), so six products cost 3/8 of the f32-MFMA time; the measured error against an fp64 product is the
 * f32-MFMA chain's or smaller (tools/probe_gemm_accuracy.py, DESIGN.md par. 5).  Build with -DSNAG_GEMM_BF16X=0 for the f32 MFMA
 * form, =9 for all nine products.
 * Replaces the aten addmm/mm calls of the path: per-modality projections nn.Linear(F_m, D)
 * (model/SNAG_tools.py:92-96,134-150), ProjectionHead (model/layers.py:135-148), the Mformer
 * linears (model/SNAG_tools.py:183-186,219,251,263) and their autograd.
 *
 *   C[m,n] (=|+=) add1[m,n] + add2[m,n] + keep(m,n) * act( alpha * sum_k A(m,k) B(n,k) + bias[n] )
 *
 * a_kcontig: 1 -> A(m,k) = A[m*lda + k] ; 0 -> A(m,k) = A[k*lda + m]        (same for B with n)
 * act: 0 none, 1 relu, 2 gelu(erf), 3 multiply by gelu'(aux[m,n]), 4 multiply by (aux[m,n] > 0)
 * (3/4 are the backward of 2/1 with aux = the saved pre-activation / activation).
 * pre_act (optional, ld = ldc) receives alpha*acc + bias.
 * keep_mask (optional) is a uint8 [M,N] inverted-dropout mask with row stride ldmask bytes (0 = dense, N), keep_scale = 1/(1-p).
 * workspace (optional, workspace_floats long) enables deterministic split-K for small-output /
 * long-K shapes; snag_gemm_workspace_floats(M,N,K) is the size that never limits the split.
 * Non-finite / extreme operands: every finite fp32 value (normals up to FLT_MAX, down to the smallest normal) multiplies as in fp32.  An
 * INFINITE operand splits to (inf, NaN, NaN) -- a - bf16(a) is inf - inf --, so outputs that fp32 would make +-inf come out NaN; a NaN
 * operand gives NaN as in fp32; a denormal operand keeps what bf16's denormal range holds (absolute error < 2^-133 per unit of the other operand).  The tables this path
 * multiplies (features, weights, activations) are finite; a non-finite input still shows in its outputs, as NaN
 * (tests/test_gemm_gpu.py::test_gemm_non_finite_and_extreme_operands_are_documented_behaviour). */
typedef struct snag_gemm_desc {
    int32_t M, N, K;
    int32_t a_kcontig, b_kcontig;
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    const float* bias;
    float alpha;
    int32_t act;
    float* pre_act;
    const uint8_t* keep_mask; float keep_scale;
    const float* add1; int64_t ld1;
    const float* add2; int64_t ld2;
    int32_t accumulate;
    float* workspace; int64_t workspace_floats;
    const float* aux; int64_t ldaux;
    int64_t ldmask;
    int32_t defer_reduce;   /* 1: when the K split of this problem produces slabs, leave them in the workspace (C is not touched);
                             * snag_gemm_reduce_pair / snag_gemm_reduce_group finishes the job.  Ignored when the problem runs without a slab split;
                             * snag_gemm_f32_pair honours it when both members set it. */
    /* optional second copy of the FINAL C values (after accumulate): mirror[m*ldmirror + n] = C[m,n] * (mirror_rowmask && mirror_rowmask[m]
     * ? mirror_scale : 1).  The entity table's gradient (the structural token's rows of d_x0 through the noise blend) leaves the
     * Mformer backward's last GEMM this way instead of through a copy launch. */
    float* mirror; int64_t ldmirror; const uint8_t* mirror_rowmask; float mirror_scale;
    /* 1: C is stored transposed -- value (m, n) goes to C[n * ldc + m], ldc >= M (accumulate, bias, act, masks, add1 / add2 still index
     * their [M, N] operands by (m, n)); no pre_act, no mirror.  The weight gradient dW^T = X^T dY lands in the nn.Linear layout
     * [out, in] this way, and an activation gradient in the K-contiguous layout its consumer streams fastest. */
    int32_t c_transposed;
    /* 0: the library chooses the K split.  > 0: this many K splits are wanted (capped by K / 64 and by what the workspace holds; the
     * reduce entries read the same field, so a deferred reduce agrees with its launch).  A caller that puts several long-K problems
     * into one grouped launch balances the workgroups' k-ranges across them with it. */
    int32_t splits;
} snag_gemm_desc;

/* Up to SNAG_GEMM_GROUP_MAX independent problems of ANY shapes in one call -- one launch when each of them would run the 64 x 128 tile
 * (plain or split K, no stream-K plan; b_kcontig equal across the group, a_kcontig free per problem since ABI version 2), or each of them
 * the eight-wave 256 x 128 tile of the long-K shapes (M >= 1024, K >= 512, N <= 128; a_kcontig / b_kcontig equal across the group), one
 * launch per problem otherwise.  Results are the same bits as separate snag_gemm_f32 calls: every problem keeps its own K split, workspace
 * (distinct per split problem) and epilogue; defer_reduce is honoured per problem.  The outputs must not overlap.  (The seven independent
 * products behind the read-out layer's attention backward -- four weight gradients and three input gradients --, reference ops: the
 * nn.Linear backward of BertSelfAttention / ProjectionHead, model/SNAG_tools.py:32-51, 134-142.) */
#define SNAG_GEMM_GROUP_MAX 8
SNAG_API int snag_gemm_f32_group(const snag_gemm_desc* const* descs, int n, void* stream);

SNAG_API int snag_gemm_f32(const snag_gemm_desc* desc, void* stream);
SNAG_API int64_t snag_gemm_workspace_floats(int M, int N, int K);
/* Two independent problems; when they have the same shape, layouts and split decision they run as ONE launch (the step's many
 * [rows x D x D]-sized GEMMs are grids of fewer workgroups than the chip has CUs: a pair fills it once instead of twice), otherwise
 * one after the other.  Results are those of two snag_gemm_f32 calls, bit for bit.  Split-K workspaces must be distinct. */
SNAG_API int snag_gemm_f32_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, void* stream);
/* Finishes two problems launched with defer_reduce = 1 (same descriptors, distinct workspaces): ONE launch sums both problems' K-split
 * slabs in slab order and applies their epilogues -- the two modalities' first projections, or their weight gradients, pay one
 * reduce launch instead of two.  A problem whose launch made no slabs (no split, stream-K) was finished by snag_gemm_f32 already and
 * is skipped.  Results are those of two plain snag_gemm_f32 calls, bit for bit. */
SNAG_API int snag_gemm_reduce_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, void* stream);
/* The same for up to four problems (n <= 4): the slab splits of f32_pair launches count too when BOTH members carried defer_reduce.
 * Problems that ran without slabs are skipped; every slab workspace must be distinct. */
SNAG_API int snag_gemm_reduce_group(const snag_gemm_desc* const* descs, int n, void* stream);

/* Gram matrix S [n, n] = alpha * Z Z^T for Z [n, k] (row stride ldz), fp32 via split-bf16 MFMA.  Only the tiles on or above the diagonal are
 * computed; each is also stored transposed, so S is symmetric bit for bit.  Used for the stacked InfoNCE logits
 * (icl_loss, model/SNAG_loss.py:99-110: the four z_a z_b^T blocks are one Gram matrix of [z1; z2]).  k == 0 gives zeros. */
SNAG_API int snag_gram_f32(const float* Z, int64_t ldz, int n, int k, float alpha, float* S, int64_t lds, void* stream);

/* ------------------------------------------------------------------------------------------
 * a1-a4, a9  Mformer modality fusion.
 * Replaces MformerFusion.forward (model/SNAG_tools.py:32-51) and the BertLayer stack under it
 * (BertSelfAttention :178-209, BertSelfOutput :218-222, BertIntermediate/BertOutput :244-266,
 * BertLayer :279-298), plus their autograd.  Tokens arrive stacked: x0[n, m, :] is what
 * torch.stack(embs, dim=1) (:36) would hold after the None entries are dropped.
 *
 * Per layer the HOST-side array `layers[l]` carries device pointers to the parameters in the
 * reference's state_dict layout (query/key/value stacked row-wise into wqkv [3D,D], bqkv [3D]) and
 * the optional inverted-dropout keep masks of that layer (NULL = eval mode / no dropout):
 * keep_probs [N,H,M,M] (:201), keep_attn_out [N*M,D] (:220), keep_ffn_out [N*M,D] (:264).
 *
 * Outputs: hidden [N,M,D]; joint, joint_fz [N, M*D]; weight_norm [N,M]  (any of the last three may
 * be NULL to skip them).  `save` receives the activations the backward needs; `work` is scratch.
 * Sizes: snag_fusion_save_floats / snag_fusion_work_floats.  M <= 6, D <= 512, D % H == 0, H*M*M <= 192 (the reference's default
 * 5 heads with all 6 modalities is 180).
 * Read-out mode with D == 128, M == 3, two co_fwd products, no FFN, H <= 4: the whole layer forward is ONE launch (csrc/readout_fused.hip;
 * the last 864 KB of `save` hold its split weight images and the backward's Wo^T ones); results and saved activations are the bits of the separate launches.
 */
typedef struct snag_fusion_layer {
    const float *wqkv, *bqkv;
    const float *wo, *bo;
    const float *ln1_g, *ln1_b;
    const float *w1, *b1;          /* intermediate.dense [I,D],[I]   (use_intermediate only) */
    const float *w2, *b2;          /* output.dense [D,I],[D] */
    const float *ln2_g, *ln2_b;    /* output.LayerNorm */
    const uint8_t *keep_probs, *keep_attn_out, *keep_ffn_out;
} snag_fusion_layer;

typedef struct snag_fusion_layer_grad {
    float *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
} snag_fusion_layer_grad;

typedef struct snag_fusion_desc {
    int32_t N, M, D, H, I, L, use_intermediate;
    float drop_p;
    const snag_fusion_layer* layers;      /* host array [L] */
    const float* weight_raw;              /* [6] */
    const float* x0;                      /* [N,M,D] */
    float *joint, *joint_fz, *hidden, *weight_norm;
    float* save; int64_t save_floats;
    float* work; int64_t work_floats;
    /* 0: every token of every layer (the reference's computation).  g + 1: the caller consumes hidden[:, g, :] ONLY (README
     * joint_way Mformer_hd_graph): the LAST layer computes keys / values for all M tokens but the query, attention row, output
     * projection, LayerNorm (and FFN) for token g alone -- same values, (2M+2)/(4M) of the GEMM work -- and `hidden` is [N, D].
     * joint / joint_fz / weight_norm must be NULL in that mode. */
    int32_t readout_slot_p1;
    /* read-out mode with L = 1 only.  1: the caller SUPPLIES the keys / values of every token but the read-out one, and takes their
     * gradient back (projection folding: a token that is x_m = mid_m l2_m^T enters the layer through K, V alone, so
     * KV_m = mid_m (Wkv l2_m)^T + b needs neither x_m nor its own GEMM pass over l2).  Forward: before the call the caller has written
     * KV_m into rows n * M + m, columns D..3D of the layer's q/k/v activation (snag_fusion_readout_offsets: `qkv` floats into `save`,
     * row pitch 3 D); only the read-out token's K, V, Q are projected here.  Backward: d_x0 receives the read-out token's rows only, the
     * key / value weight gradient its contribution only, and the caller reads dKV_m from rows n * M + m of the [N * M, 2 D] block at
     * `dkv` floats into `work` (it owes d mid_m = dKV_m (Wkv l2_m), d(Wkv l2_m) = dKV_m^T mid_m and the chain rule onto Wkv and l2_m). */
    int32_t fold_others;
    /* fold_others only.  The caller's own key / value products (A and B K-contiguous, no K split) that snag_fusion_fwd launches TOGETHER
     * with the read-out token's q/k/v projection -- one grouped launch under the rules of snag_gemm_f32_group (each of these
     * [rows x 2D x D] products alone is a grid of fewer workgroups than the chip has CUs).  They must be complete descriptors whose
     * operands are ready when snag_fusion_fwd is called. */
    const snag_gemm_desc* co_fwd[2];
    int32_t n_co_fwd;
    /* 1: the split weight images of the one-launch read-out forward were prepared for THIS step's weights by snag_fusion_fwd_prepare_reduce
     * (which reports it); snag_fusion_fwd then skips its own preparation launch.  0 (default): snag_fusion_fwd prepares them itself. */
    int32_t weight_images_ready;
    /* The one-launch read-out forward (and the backward's fused launch) pay off from ~5,000 (10,000) entities on: a workgroup walks the whole
     * chain, so its time does not shrink with N -- one rank's 1,875 rows of an 8-way row-sharded step run faster as separate launches (224 vs 238 us
     * for the rank's step).  Below the thresholds the library keeps the separate launches; 1 here takes the one-launch paths whatever N (tests, A/B). */
    int32_t force_one_launch;
} snag_fusion_desc;

/* snag_gemm_reduce_group(descs, n) AND, in the same launch, the preparation of the read-out forward's weight images for `f` (its layers'
 * parameters and co_fwd products as snag_fusion_fwd will see them; the weights must be final, the activations need not exist yet): the
 * preparation is parameter-sized, depends on nothing the reduce touches, and a launch of its own costs ~5 us on the step's critical path.
 * *prepared = 1 when the images were written (set f->weight_images_ready = 1 for the coming snag_fusion_fwd), 0 when `f` does not take the
 * one-launch path (nothing else changes: the reduce ran). */
SNAG_API int snag_fusion_fwd_prepare_reduce(const snag_fusion_desc* f, const snag_gemm_desc* const* descs, int n, int32_t* prepared, void* stream);
SNAG_API int64_t snag_fusion_save_floats(int N, int M, int D, int H, int I, int L, int use_intermediate);
SNAG_API int64_t snag_fusion_work_floats(int N, int M, int D, int H, int I, int L, int use_intermediate);
SNAG_API int snag_fusion_fwd(const snag_fusion_desc* desc, void* stream);
/* where the read-out layer keeps its stacked q/k/v activation ([N * M, 3 D], floats into `save`) and the gradient of its keys / values
 * ([N * M, 2 D], floats into `work`): see fold_others */
SNAG_API int snag_fusion_readout_offsets(const snag_fusion_desc* desc, int64_t* qkv_save_offset, int64_t* dkv_work_offset);
/* Up to four small dense products in ONE launch, plain fp32 fused multiply-adds in k order (an f32 fmaf chain per output):
 *   C[m * ldc + n] (=|+=) sum_k A(m, k) B(k, n),  A(m, k) = A[m * sam + k * sak],  B(k, n) = B[k * sbk + n * sbn]
 * (the problems of one launch must write disjoint outputs)
 * for the parameter-sized products of projection folding (Wkv l2_m and the chain rule back onto Wkv / l2_m: 128..256-sized). */
typedef struct snag_small_mm {
    int32_t M, N, K, accumulate;
    const float* A; int64_t sam, sak;
    const float* B; int64_t sbk, sbn;
    float* C; int64_t ldc;
    const float* A2; const float* B2;     /* optional second term with the same sizes and strides: C (=|+=) A B + A2 B2 (NULL: none) */
} snag_small_mm;
/* Where snag_fusion_bwd keeps what the LayerNorm backward of the read-out rows reads and writes (float offsets into f->save / the work
 * buffer; read-out mode without FFN, N <= 16384): the pre-LayerNorm rows t [N, D] and (mean, rstd) [N, 2] in save; dt [N, D], dtm [N, D] and
 * the partial slab [1024, 3, D] in work.  A caller that sets snag_fusion_bwd_desc.ln_bwd_done fills dt / dtm / the slab itself. */
SNAG_API int snag_fusion_readout_ln_offsets(const snag_fusion_desc* f, int64_t* t_save_offset, int64_t* stats_save_offset, int64_t* dt_work_offset,
                                            int64_t* dtm_work_offset, int64_t* part_work_offset);
SNAG_API int snag_small_matmul(const snag_small_mm* problems, int n_problems, void* stream);
/* snag_gemm_reduce_group (n <= 4 deferred K-split problems) AND up to four small products in ONE launch: the two jobs are independent
 * (the small products must not read what the reduces write, nor write what they read) and each alone is a short, latency-bound launch --
 * the reduce of the first projections' weight gradients and the chain rule of projection folding onto Wkv / l2_m run side by side
 * instead of one after the other.  Results are those of the two separate calls, bit for bit.  n_small = 0: snag_gemm_reduce_group. */
SNAG_API int snag_gemm_reduce_group_mm(const snag_gemm_desc* const* descs, int n, const snag_small_mm* small_products, int n_small, void* stream);

/* Backward.  `f` is the forward's descriptor (same save buffer, still holding the forward's
 * activations; f.hidden etc. are not read).  Upstream gradients d_* may each be NULL (= zero).
 * d_x0 [N,M,D] is overwritten; parameter gradients are overwritten (accumulate = 0) or added to.
 * Deterministic: every cross-row reduction uses fixed-order partial sums, no atomics.
 */
typedef struct snag_fusion_bwd_desc {
    snag_fusion_desc f;
    const float *d_joint, *d_joint_fz, *d_hidden, *d_weight_norm;
    float* d_x0;
    const snag_fusion_layer_grad* grads;  /* host array [L] */
    float* d_weight_raw;                  /* [6] */
    int32_t accumulate;
    int32_t zero_d_hidden;                /* 1: d_hidden is overwritten with zeros once it has been read (a scatter-add target is
                                             clean for the next step without a fill launch); d_hidden is [N, D] in read-out mode */
    /* read-out mode with L = 1 only (SNAG_ERR_UNSUPPORTED otherwise): the rows of token readout_slot_p1 - 1 of d_x0 are ALSO written
     * to d_slot_out [N, d_slot_ld], scaled by d_slot_scale where d_slot_rowmask[n] != 0 -- what snag_entity_token_bwd would compute
     * from d_x0 in a launch of its own (the entity-table gradient through the noise blend). */
    float* d_slot_out; int64_t d_slot_ld; const uint8_t* d_slot_rowmask; float d_slot_scale;
    /* fold_others only: the caller's products on the key / value gradient (see fold_others), launched together with this call's own
     * products of the same operand layouts, in one grouped launch each (snag_gemm_f32_group rules):
     *   co_dx: A K-contiguous, B row-contiguous, no K split (d mid_m = dKV_m Wf_m ...), with the read-out token's d x = dqkv Wqkv;
     *   co_dw: A and B row-contiguous, K split over the rows with defer_reduce = 1 and a workspace of their own (d Wf_m = dKV_m^T mid_m),
     *          with d Wo and the read-out token's d Wqkv; ONE reduce launch finishes all four. */
    const snag_gemm_desc* co_dx[2]; int32_t n_co_dx;
    const snag_gemm_desc* co_dw[2]; int32_t n_co_dw;
    /* read-out mode without FFN: 1 = the LayerNorm backward of the read-out rows has been done by the caller's scorer
     * (snag_score_margin_fwd_bwd_seg_ln wrote dt, dtm and its partial slab of ceil(N / 16) rows at the offsets of
     * snag_fusion_readout_ln_offsets): d_hidden is not read, the backward starts at the output projection. */
    int32_t ln_bwd_done;
    /* 1: snag_fusion_fwd_prepare_reduce reported `prepared` for this step's weights (it also writes the output projection's weight as the backward's
     * d ctx = dz Wo operand); 0 (default): the backward prepares that operand with a launch of its own when it needs it. */
    int32_t wo_images_ready;
    /* 1: the read-out layer's d ctx product and attention backward stay two launches where the library would run them as one (hidden_size 128,
     * three tokens, fold_others): an A/B and test handle -- every gradient is the same bits either way except dbqkv, whose column sums over the
     * entities are associated differently (one partial per 32 entities instead of one per strided workgroup). */
    int32_t separate_attention_backward;
    /* (ABI 3) A grouped launch the caller would issue RIGHT BEHIND this call -- snag_gemm_f32_group(next_group, n_next_group): e.g. the first
     * projections' weight gradients, which read what this call's input-gradient products write -- and that nobody of this call depends on.
     * When given (read-out mode with fold_others, whose last launch is a reduce of the layer's weight-gradient slabs + its slab column sums; the
     * full layer, whose layer-0 reduce of the same kind nothing of this call reads), this call launches that group itself and that reduce RIDES in it: a group of eight-wave 256-row tiles leaves CUs idle
     * (247 workgroups on 256 CUs) and the reduce runs on those instead of as a 9 us launch of its own.  Same bits.  Where the path or the
     * group's plan does not allow it the reduce and the group go out one after the other, as if the caller had issued the group. */
    const snag_gemm_desc** next_group; int32_t n_next_group;
} snag_fusion_bwd_desc;

SNAG_API int snag_fusion_bwd(const snag_fusion_bwd_desc* desc, void* stream);

/* torch.stack(embs, dim=1) for up to 6 [N,D] tensors (emb[m] = src[m]) with the entity-noise blend of
 * model/SNAG_tools.py:127-128 folded into slot g_slot when noise != NULL:
 *   out[n,g,:] = (1 - a) e + a noise[n,:] for rows with noise_mask[n] != 0, a = mask_ratio / 2. */
SNAG_API int snag_pack_tokens(const float* const* src_host, int M, int64_t N, int D, int g_slot,
                              const float* noise, const uint8_t* noise_mask, float half_mask_ratio,
                              float* out, void* stream);

/* README joint_way "Mformer_hd_mean": out[n,:] = mean_m hidden[n,m,:]  (a9; parity unpinned). */
SNAG_API int snag_token_mean(const float* hidden, int64_t N, int M, int D, float* out, void* stream);

/* README joint_way "atten_weight" (a9; README.md:51): non-transformer attention over an entity's M modality tokens x [rows, M, D] (dense).
 * The only arithmetic of that name in the reference tree is EVA_tools.SelfAttention.forward (model/EVA_tools.py:190-232, dead code):
 *   pre_m = x_m . a ;  w = softmax_m(relu(pre)) ;  out[n, :] = sum_m w_m x[n, m, :]            (a [D] learnable; PARITY UNPINNED).
 * aw [rows, 2 M] receives (w, pre) for the backward.  1 <= M <= 6, 4 | D <= 512.
 * snag_atten_weight_bwd: d_x [rows, M, D] = w_m d_out + d pre_m a (written), d_a [D] (=|+=) sum_n sum_m d pre_m x_m through
 *   snag_atten_weight_work_floats(D) floats of fixed-order partial slabs (no float atomics). */
SNAG_API int64_t snag_atten_weight_work_floats(int D);
SNAG_API int snag_atten_weight_fwd(const float* x, int64_t rows, int M, int D, const float* a, float* out, int64_t ld_out, float* aw,
                                   void* stream);
SNAG_API int snag_atten_weight_bwd(const float* x, int64_t rows, int M, int D, const float* a, const float* aw, const float* d_out,
                                   int64_t ld_dout, float* d_x, float* d_a, int accumulate_da, float* work, void* stream);

/* README POOL=1 / use_pool (README.md:68,79: "pooling operations are applied to all pre-extracted visual/text features for
 * dimensionality reduction and uniformity in dimensions"; no code in the tree: PARITY UNPINNED).  Every row of x [N, H * W] (row stride ld),
 * viewed as an H x W image, is adaptively average-pooled to oh x ow (torch.nn.functional.adaptive_avg_pool2d's windows; H = oh = 1 is
 * the 1-D pool) into out [N, oh * ow] (row stride ld_out).  Once per feature table at load time. */
SNAG_API int snag_adaptive_avg_pool_rows(const float* x, int64_t N, int64_t ld, int H, int W, int oh, int ow, float* out, int64_t ld_out,
                                         void* stream);

/* ------------------------------------------------------------------------------------------
 * a6  Gaussian noise-mask injection (GMNM) and its statistics.
 * snag_colstats: per-column mean and UNBIASED std over the rows with exclude[n] == 0
 *   (replaces SNAG.get_mean_std model/SNAG.py:77-84 and the entity stats :94-95; n_valid = number
 *   of such rows, exclude may be NULL).  work: snag_colstats_work_floats(C) floats.  std may be NULL.
 * snag_noise_apply: replaces SNAG.add_noise_to_embeddings model/SNAG.py:66-75 for one feature table:
 *   out[n,:] = (1 - mask_ratio) x[n,:] + mask_ratio (mean + std * z[n,:]) where row_mask[n] != 0, else x[n,:].
 *   Parity mode: z is the reference's compact randn_like(X[mask]) [n_sel, F] and sel_index[n] the
 *   row of z that belongs to entity n.  Perf mode (z == sel_index == NULL): z = Philox(seed).
 * snag_dropout_keep_mask: Bernoulli keep mask (u >= p) from Philox(seed): the row mask of :68 with
 *   p = 1 - noise_ratio inverted by the caller, and the nn.Dropout(0.1) masks of the Bert blocks.
 *   tick (optional device counter, see snag_tick) is mixed into the seed so that a replayed
 *   hipGraph draws a fresh mask every step.
 * snag_gaussian_fill: out[n,c] = mean[c] + std[c] * z, z ~ Philox(seed)  (entity_noise, :96).
 */
SNAG_API int64_t snag_colstats_work_floats(int C);
SNAG_API int snag_colstats(const float* x, int64_t R, int C, const uint8_t* exclude, int64_t n_valid,
                           float* mean, float* std_, float* work, void* stream);
SNAG_API int snag_noise_apply(const float* x, int64_t N, int F, const float* mean, const float* std_,
                              const uint8_t* row_mask, const int32_t* sel_index, const float* z,
                              uint64_t seed, float mask_ratio, float* out, void* stream);
SNAG_API int snag_dropout_keep_mask(uint64_t seed, const uint64_t* tick, int64_t n, float p, uint8_t* out, void* stream);
/* Bernoulli byte masks as REGIONS of global streams: region i covers the elements [offset, offset + n) of the stream Philox(seed [+ tick]):
 * element g is word g % 4 of philox4x32_10(seed, g / 4), whichever region asks for it, and out[j] = (u(offset + j) >= p) != invert.
 * A rank that owns the entity rows [lo, hi) of a [N, W] mask therefore draws, with offset = lo * W, exactly the bytes the single rank
 * draws for those rows (SURVEY.md par. 8e: the G-rank step equals the 1-rank step with dropout and noise ON).  invert = 0: the keep mask
 * of nn.Dropout(p); invert = 1: the "rand < p" row mask of the noise injection (model/SNAG.py:68).  snag_dropout_keep_mask is the one
 * region (seed, 0, n, p, 0).  Up to SNAG_MAX_MASK_REGIONS regions per launch; the table is read on the host. */
#define SNAG_MAX_MASK_REGIONS 8
typedef struct snag_mask_region {
    uint64_t seed;
    int64_t offset;
    int64_t n;
    float p;
    int32_t invert;
    uint8_t* out;
    /* period > 0: only the elements j of the region with skip <= j % period < skip + take are drawn (the others are left alone); offset,
     * period, skip and take multiples of 4, n a multiple of period.  The read-out-only last Mformer layer consumes the output-dropout mask
     * of ONE token: rows of `take` = D bytes every `period` = M D.  The bytes drawn are those of the dense region.  period = 0: dense. */
    int64_t period, skip, take;
} snag_mask_region;
SNAG_API int snag_bernoulli_regions(const snag_mask_region* regions, int n_regions, const uint64_t* tick, void* stream);
/* snag_noise_apply for a table that is refreshed once per epoch (model/SNAG.py:86-92), incremental and keyed on GLOBAL rows:
 *   rows with new_mask[n] != 0:            noisy[n,:] = (1 - mask_ratio) x[n,:] + mask_ratio (mean + std * z), z = Philox(seed, ((row0 + n) F + c) / 4)
 *   rows with only old_mask[n] != 0:       noisy[n,:] = x[n,:]      (last epoch's noise taken back)
 *   rows in neither mask:                  not touched (they equal x already; old_mask == NULL: every row is written)
 * x holds the rows [row0, row0 + N) of the whole table: a row-owning rank draws the noise the single rank draws for its rows.
 * noisy_t (optional): the transposed copy [F, ld_t] (ld_t >= N), rewritten whole; noisy may then be NULL. */
SNAG_API int snag_noise_refresh(const float* x, int64_t N, int F, const float* mean, const float* std_, const uint8_t* new_mask,
                                const uint8_t* old_mask, uint64_t seed, int64_t row0, float mask_ratio, float* noisy, float* noisy_t,
                                int64_t ld_t, void* stream);
SNAG_API int snag_gaussian_fill(uint64_t seed, int64_t N, int F, const float* mean, const float* std_,
                                float* out, void* stream);

/* Pieces of the projection / read-out backward that are not GEMMs.
 * snag_colsum: out[c] (=|+=) sum_r x[r*ld + c]  (nn.Linear bias gradient), fixed-order partial sums;
 *   work: 256*C floats.
 * snag_token_mean_bwd: d_hidden[n,m,:] = d_out[n,:] / M  (backward of snag_token_mean); zero_src = 1 clears d_out once read
 *   (d_out is the scorer's scatter-add target: clean for the next step without a fill launch).
 * snag_entity_token_bwd: gradient of the entity table through the noise blend of
 *   model/SNAG_tools.py:127-128: d_table[n,:] (=|+=) (noise_mask[n] ? 1 - a : 1) * d_tok[n*ld + :];
 *   snag_entity_token_fwd is the forward blend written straight into a token slot (row stride out_ld). */
SNAG_API int snag_colsum(const float* x, int64_t R, int C, int64_t ld, float* out, int accumulate,
                         float* work, void* stream);
/* Two column-sum problems of one shape in one pair of launches (the two modalities' projection bias gradients); bit-equal to two
 * snag_colsum calls.  work: 2 x 256 x C floats. */
SNAG_API int snag_colsum_pair(const float* x0, const float* x1, int64_t R, int C, int64_t ld, float* out0, float* out1, int accumulate,
                              float* work, void* stream);
/* dst [cols, rows] (row stride ld_dst) = transpose of src [rows, cols] (row stride ld_src).  The MKGC engine keeps each raw feature
 * table in both layouts so that the first projection X W^T (a5, model/SNAG_tools.py:134-150) and its weight gradient both stream
 * row-contiguous operands; the transposed copy is refreshed once per noise epoch (model/SNAG.py:86-98). */
SNAG_API int snag_transpose_f32(const float* src, int64_t ld_src, int64_t rows, int64_t cols, float* dst, int64_t ld_dst, void* stream);
SNAG_API int snag_token_mean_bwd(float* d_out, int64_t N, int M, int D, float* d_hidden, int zero_src, void* stream);
SNAG_API int snag_entity_token_fwd(const float* table, int64_t N, int D, const float* noise, const uint8_t* noise_mask,
                                   float half_mask_ratio, float* out, int64_t out_ld, void* stream);
SNAG_API int snag_entity_token_bwd(const float* d_tok, int64_t ld, int64_t N, int D, const uint8_t* noise_mask,
                                   float half_mask_ratio, float* d_table, int accumulate, void* stream);

/* torch.optim.AdamW update over one flat buffer (the reference's optimiser, src/utils.py:25-80,
 * is out of the hot path; this only closes the benchmark's training step without leaving HIP).
 * grad is multiplied by grad_scale first (1/world_size after a summed all-reduce). */
SNAG_API int snag_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             const uint64_t* tick, float grad_scale, void* stream);

/* The same update with the reference's parameter groups (set_optim, src/utils.py:47-55: no weight decay for names containing
 * 'bias' / 'weight_raw' / 'multi_loss_layer', 5x learning rate for 'multi_loss_layer'): the flat buffer is cut into n_seg
 * consecutive segments, segment s = [seg_end[s-1], seg_end[s]) (seg_end[n_seg-1] >= n), each with its own weight decay and
 * learning-rate multiplier.  seg_* are DEVICE arrays.  grad_scale_dev: optional device scalar multiplied into grad_scale. */
SNAG_API int snag_adamw_step_groups(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    float lr, float beta1, float beta2, float eps, const int64_t* seg_end,
                                    const float* seg_weight_decay, const float* seg_lr_scale, int n_seg, int64_t step,
                                    const uint64_t* tick, float grad_scale, const float* grad_scale_dev, int tick_advanced, void* stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) of the reference's training step (main.py:266-268, --clip default 1.0) for a
 * flat gradient buffer, without a host round trip: out2[0] = min(1, max_norm / (|grad_scale * grad|_2 + 1e-6)), out2[1] = that norm
 * (fixed-order two-level sum).  Pass out2 as grad_scale_dev of snag_adamw_step_groups: the update then uses the clipped gradient
 * (the gradient buffer itself is left as it is).  work: 1024 floats.  advance_tick (optional): the device step counter is incremented
 * by this call's single-block tail instead of a separate snag_tick launch; pass tick_advanced = 1 to the AdamW call that follows. */
SNAG_API int snag_clip_grad_norm(const float* grad, int64_t n, float grad_scale, float max_norm, float* out2, float* work,
                                uint64_t* advance_tick, void* stream);

/* snag_clip_grad_norm + snag_adamw_step_groups + the step counter's increment as TWO launches instead of three: the sum-of-squares
 * partials (work: 1024 floats) are turned into the clip coefficient by every AdamW workgroup itself, in one fixed order; the partial-
 * sum launch advances *tick.  out2 (optional): {coefficient, gradient norm}.  The update is that of
 * clip_grad_norm_(max_norm) followed by AdamW at step *tick + 1 (bias correction from the device counter). */
SNAG_API int snag_clip_adamw_step_groups(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                         float beta2, float eps, const int64_t* seg_end, const float* seg_weight_decay,
                                         const float* seg_lr_scale, int n_seg, uint64_t* tick, float grad_scale, float max_norm, float* out2,
                                         float* work, void* stream);

/* Device-resident step counter: tick[0] += 1.  When `tick` is given to snag_adamw_step the bias
 * correction uses t = tick[0] + 1 instead of the host `step`, which lets a captured hipGraph of a
 * whole training step be replayed without re-recording. */
SNAG_API int snag_tick(uint64_t* tick, void* stream);

/* ------------------------------------------------------------------------------------------
 * a8  Entity x entity similarity evaluation (the MMEA eval path).
 * snag_row_sqnorm:    out[i] = |x_i|^2                               (src/utils.py:210-212)
 * snag_pairwise_dist: out[i,j] = max(xn[i] + yn[j] - 2 x_i.y_j, 0)   (pairwise_distances, src/utils.py:202-218);
 *                     as_similarity != 0 stores 1 - that (main.py:392).  The x.y^T product is the path's one
 *                     dense entity x entity GEMM and runs on the matrix cores (fp32 in, fp32 accumulate, split-bf16 products).
 * snag_csls_neighbourhood: r1[i] = mean of the k largest S[i,:], r2[j] = mean of the k largest S[:,j]
 *                     (csls_sim, src/utils.py:431-432), S = mat or 1 - mat (from_dist).  k <= 16.
 *                     work: snag_eval_work_floats(n, m) floats.
 * snag_csls_apply:    out = (2 S - r1[:,None]) - r2[None,:]          (src/utils.py:433-434)
 * snag_rank_diagonal: rank_l2r[i] = #{j : D'[i,j] < D'[i,i]}, rank_r2l[j] = #{i : D'[i,j] < D'[j,j]} with
 *                     D' = dist, or (r1/r2 given) 1 - ((2 (1 - dist) - r1) - r2): the position of the true
 *                     match in the reference's ascending torch.sort (main.py:400-402,422-424) for tie-free rows.
 * snag_rank_summary:  out[0..ntop) = #{rank < topk[q]}, out[ntop] = sum(rank+1), out[ntop+1] = sum 1/(rank+1)
 *                     in double (main.py:403-407,430-436); the caller divides by n and rounds Hits to 4 dp.
 * snag_argmin_rows_cols: arg-min of every row and column (ties -> smallest index): the mutual-nearest-
 *                     neighbour test of SNAG.Iter_new_links (model/SNAG.py:200-204).  work as above.
 */
SNAG_API int snag_row_sqnorm(const float* x, int64_t n, int d, int64_t ld, float* out, void* stream);
SNAG_API int snag_pairwise_dist(const float* x, int64_t ldx, int n, const float* y, int64_t ldy, int m, int d,
                                const float* xn, const float* yn, float* out, int64_t ld_out, int as_similarity, void* stream);
/* Row-sharded similarity evaluation across ranks (SURVEY.md par. 8e, BASELINE configs[3] on several GPUs): a rank holds rows
 * [row0, row0 + n) of the n_total x m distance matrix (all m columns).  Per rank:
 *   snag_csls_neighbourhood on its block      -> r1 of its rows (complete) and a LOCAL r2 that is not used;
 *   snag_col_topk                             -> out[k, m]: the k largest similarities per column among its rows (descending);
 *   all-gather of those candidates [G, k, m], snag_col_topk_mean -> r2[m], bit-equal to the single-rank value;
 *   snag_rank_rows_block                      -> rank_l2r of its rows and the adjusted diagonal values diag[i] = D'[row0+i, row0+i];
 *   all-gather of diag [m], snag_rank_cols_block -> count[j] = #{its rows i : D'[i,j] < diag[j]}; all-reduce(sum) = rank_r2l.
 * The ranks equal those of snag_rank_diagonal on the whole matrix exactly (integers; same float expressions). */
SNAG_API int snag_col_topk(const float* mat, int64_t ld, int n, int m, int k, int from_dist, float* out, float* work, void* stream);
SNAG_API int snag_col_topk_mean(const float* cand, int chunks, int m, int k, float* out, void* stream);
SNAG_API int snag_rank_rows_block(const float* dist, int64_t ld, int n, int m, int row0, const float* r1, const float* r2, int32_t* rank,
                                  float* diag, void* stream);
SNAG_API int snag_rank_cols_block(const float* dist, int64_t ld, int n, int m, const float* r1, const float* r2, const float* diag,
                                  int32_t* count, void* stream);

/* --distance 1 of Runner._test (main.py:388-391: scipy cdist(..., metric="cityblock") on the CPU): out[i,j] = sum_k |x[i,k] - y[j,k]|. */
SNAG_API int snag_cityblock_dist(const float* x, int64_t ldx, int n, const float* y, int64_t ldy, int m, int d, float* out, int64_t ld_out,
                                 void* stream);
SNAG_API int64_t snag_eval_work_floats(int n, int m);
SNAG_API int snag_csls_neighbourhood(const float* mat, int64_t ld, int n, int m, int k, int from_dist,
                                     float* r1, float* r2, float* work, void* stream);
SNAG_API int snag_csls_apply(const float* sim, int64_t ld, int n, int m, const float* r1, const float* r2, float* out, void* stream);
SNAG_API int snag_rank_diagonal(const float* dist, int64_t ld, int n, int m, const float* r1, const float* r2,
                                int32_t* rank_l2r, int32_t* rank_r2l, void* stream);
SNAG_API int snag_rank_summary(const int32_t* rank, int n, const int32_t* topk, int ntop, double* out, void* stream);
/* Both directions of one evaluation in one launch: out[0 .. ntop+2) summarises rank_a, out[ntop+2 .. 2(ntop+2)) rank_b (same n, same
 * thresholds); the numbers of two snag_rank_summary calls. */
SNAG_API int snag_rank_summary_pair(const int32_t* rank_a, const int32_t* rank_b, int n, const int32_t* topk, int ntop, double* out, void* stream);
SNAG_API int snag_argmin_rows_cols(const float* dist, int64_t ld, int n, int m, int32_t* row_argmin, int32_t* col_argmin,
                                   float* work, void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY par. 8(f) "next" #1: the MMEA contrastive objective.
 * Replaces the non-GEMM device work of icl_loss.forward / softXEnt (model/SNAG_loss.py:42-128); the four B x B logit
 * blocks and the eight gradient products are snag_gemm_f32 launches issued by the host module.
 * snag_gather_normalize_fwd: z[b,:] = x / max(|x|, 1e-12) (normalize != 0) or x, x = emb[idx[b], :]; den[b] = that
 *   denominator (F.normalize(emb) then emb[train_links[:, s]], :59-64).   _bwd scatters into d_emb with fp32 atomics.
 * snag_infonce_rows: S [2B, 2B] = Z Z^T / tau for the stacked views Z = [z1; z2] (one GEMM holds the reference's four blocks
 *   ab, aa, ba, bb of :99-110).  lse[r] = logsumexp_j S[r, j] over j != r (the reference's "- masks * 1e9" on the self block),
 *   rowloss[r] = lse[r] - S[r, (r + B) mod 2B]  (-log_softmax at the target, :45-53); rows 0..B-1 are loss_a's, B..2B-1 loss_b's.
 * snag_infonce_grad: in place S <- G + G^T, G[r, j] = c_r (softmax_r[j] - [j == target(r)]), 0 on the diagonal,
 *   c_r = (r < B ? coef_a : coef_b) * (w ? w[r mod B] : 1) * (scale ? *scale : 1)
 *   (scale: the upstream gradient as a device scalar, so the backward needs no host sync); then d loss / d Z = (G + G^T) Z / tau is one GEMM.
 */
SNAG_API int snag_gather_normalize_fwd(const float* emb, int64_t ld, int64_t n_rows, int d, const void* idx, int idx_bytes, int64_t B,
                                       int normalize, float* z, float* den, void* stream);
SNAG_API int snag_gather_normalize_bwd(const float* dz, const float* z, const float* den, int d, const void* idx, int idx_bytes,
                                       int64_t B, int normalize, float* d_emb, int64_t ld, void* stream);
SNAG_API int snag_infonce_rows(const float* S, int64_t ld, int B, float* lse, float* rowloss, void* stream);
SNAG_API int snag_infonce_grad(float* S, int64_t ld, int B, const float* lse, const float* w, float coef_a, float coef_b,
                               const float* scale, void* stream);

/* SURVEY par. 8(f) "next" #3: negative sampling on the device (no Python sampler, no H2D index copies).
 * Follows generate_neg_triples_fast (model/MSNEA_tools.py:59-90): a candidate that forms a known triple is redrawn, up to
 * 10 tries, the last one kept regardless.  mode 0: n_neg negative TAILS per positive (the MKGC shape of BASELINE.json);
 * mode 1: per positive a Bernoulli(0.5) choice between corrupting the head and the tail (the in-tree behaviour), heads in
 * neg_h, tails in neg_t.  known_sorted: ascending int64 keys (h * n_rel + r) * n_ent + t of all known triples, or NULL.
 * Deviations (parity unpinned, RNG streams cannot match): candidates are drawn WITH replacement (random.sample draws
 * without), from 0..n_ent-1 (the reference samples the KG's own entity list).  tick: see snag_tick. int32 indices. */
SNAG_API int snag_sample_negatives(uint64_t seed, const uint64_t* tick, const int32_t* h, const int32_t* r, const int32_t* t,
                                   int64_t batch, int n_neg, int n_ent, int n_rel, int mode, const int64_t* known_sorted,
                                   int64_t n_known, int32_t* neg_h, int32_t* neg_t, void* stream);

/* ONE launch for the three independent, small launches that open an MKGC training step: snag_sample_negatives (mode 0: negative
 * tails; same Philox stream, same results), snag_entity_token_fwd (the structural token with the noise blend) and
 * snag_bernoulli_regions (the step's Bert dropout masks, one region per mask tensor, each keyed on the global row of its first
 * element), and up to four parameter-sized products (snag_small_matmul's problems: the folded-projection weights of the step, which
 * depend on nothing but the parameters).  Workgroup ranges of the one grid run the bodies side by side; any part is skipped when its
 * size is zero (batch * n_neg, n_rows, n_mask_regions, n_small_products). */
SNAG_API int snag_step_prologue(uint64_t sample_seed, const uint64_t* tick, const int32_t* h, const int32_t* r, const int32_t* t, int64_t batch,
                                int n_neg, int n_ent, int n_rel, const int64_t* known_sorted, int64_t n_known, int32_t* neg_t,
                                const float* table, int64_t n_rows, int dim, const float* noise, const uint8_t* noise_mask, float half_mask_ratio,
                                float* tok, int64_t tok_ld, const snag_mask_region* mask_regions, int n_mask_regions,
                                const snag_small_mm* small_products, int n_small_products, void* stream);

/* The same sampler with the two in-tree properties snag_sample_negatives gives up: candidates are drawn from the KG's OWN entity
 * list ent_list [n_list] (int32 ids; NULL = 0..n_ent-1; model/MSNEA_tools.py:64-65) and the n_neg negatives of one positive are
 * DISTINCT (random.sample draws without replacement, :69,:72).  Rounds = the reference's 10 tries: in round a slot k of positive
 * b draws ent_list[(philox4x32_10(seed, (b*n_neg + k)*16 + a).x * n_list) >> 32]; it is accepted unless it forms a known triple,
 * was already accepted for this positive, or a lower slot accepted the same value in this round; round 9 accepts whatever it
 * drew (:76-78).  n_neg <= 1024 and <= n_list.  oracle/sampler.py restates rule and stream bit for bit; the python `random` stream of
 * the reference cannot be matched, so parity with the reference itself stays unpinned. */
SNAG_API int snag_sample_negatives_distinct(uint64_t seed, const uint64_t* tick, const int32_t* h, const int32_t* r, const int32_t* t,
                                   int64_t batch, int n_neg, int n_ent, int n_rel, int mode, const int64_t* known_sorted,
                                   int64_t n_known, const int32_t* ent_list, int n_list, int32_t* neg_h, int32_t* neg_t, void* stream);

/* The filtered setting of MKGC tail prediction (PARITY UNPINNED like the eval itself): mask [batch, n_ent] (uint8, overwritten) gets 1
 * at every x for which (h_b, r_b, x) is a known triple, except x == t_b; known_sorted as for snag_sample_negatives.  The mask is
 * what snag_rank_full_table takes as `filter`.  int32 indices. */
SNAG_API int snag_build_filter_mask(const int32_t* h, const int32_t* r, const int32_t* t, int64_t batch, int n_ent, int n_rel,
                                    const int64_t* known_sorted, int64_t n_known, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY par. 8(f) "next" #2: sparse GAT structural encoder, diag variant, eval / zero-dropout configuration.
 * Replaces one MultiHeadGraphAttention layer + the head mean (+ ELU) of GAT.forward (model/layers.py:68-94,
 * model/Tool_model.py:75-94; SpecialSpmmFunction :9-29 disappears: no N x N sparse tensors are built).
 * Adjacency: CSR over the SOURCE node of the COO index list adj._indices() (duplicates and self loops kept, values unused):
 * rowptr [N+1], col [E], int32.  w [H, f] (the reference's [H,1,f]), a_src_dst [H, 2f] ([H,2f,1]).  H <= 4, f <= 512.
 * fwd keeps s_src, s_dst [N,H], agg [N,H,f], rsum [N,H] (and pre [N,f] when apply_elu) for the backward.
 * bwd: d_x is overwritten; d_w / d_a are overwritten or accumulated; work: snag_gat_work_floats(N, f, H) floats.
 * The backward also takes the same edge set grouped by DESTINATION (t_rowptr [N+1], t_col [E] = sources): it runs as two gather
 * passes (out-edges per source, in-edges per destination) with no atomics, and parameter gradients use fixed-order partial
 * sums, so it is deterministic.
 * Training options (ABI version 2): edge_keep [H, n_edges] (bytes, in the STORED order of the COO list; NULL = no dropout) is the keep mask
 * of F.dropout(edge_e, attn_dropout) (model/layers.py:83: applied after the row sums -- the aggregation uses e * keep * edge_keep_scale, the
 * normaliser the undropped e); eid [E] / t_eid [E] give the stored edge id of every position of the two CSR orders.  The input dropout of
 * the non-final layers (model/Tool_model.py:78-79) is snag_keep_scale: out = keep ? x * scale : 0, forward and backward alike.
 * instance_normalization: the reference's nn.InstanceNorm1d(n_units[0]) raises on [N, f] node features (ValueError, torch >= 1.x): not built.
 */
SNAG_API int64_t snag_gat_work_floats(int64_t N, int f, int H);
SNAG_API int snag_keep_scale(const float* x, int64_t n, const uint8_t* keep, float scale, float* out, void* stream);
SNAG_API int snag_gat_layer_fwd(const float* x, int64_t N, int f, int H, const float* w, const float* a_src_dst,
                                const int32_t* rowptr, const int32_t* col, int apply_elu,
                                float* s_src, float* s_dst, float* agg, float* rsum, float* pre, float* out,
                                const uint8_t* edge_keep, const int32_t* eid, int64_t n_edges, float edge_keep_scale, void* stream);
SNAG_API int snag_gat_layer_bwd(const float* x, int64_t N, int f, int H, const float* w, const float* a_src_dst,
                                const int32_t* rowptr, const int32_t* col, const int32_t* t_rowptr, const int32_t* t_col,
                                const float* s_src, const float* s_dst, const float* agg, const float* rsum, const float* pre, int had_elu, const float* d_out,
                                float* d_x, float* d_w, float* d_a, int accumulate, float* work,
                                const uint8_t* edge_keep, const int32_t* eid, const int32_t* t_eid, int64_t n_edges, float edge_keep_scale, void* stream);

/* GCN structural encoder option (structure_encoder == "gcn": GraphConvolution model/layers.py:102-130, GCN model/Tool_model.py:98-110):
 * out [N, f] = A x (+ bias) with A given as CSR (rowptr [N+1], col/val [E]; duplicate coordinates add up) -- the torch.spmm of
 * GraphConvolution.forward.  The backward w.r.t. x is the same call on the transposed CSR.  f <= 512. */
SNAG_API int snag_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t N, int f,
                           const float* bias, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SNAG_HIP_H */
