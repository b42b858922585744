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

#define SNAG_ABI_VERSION 1

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
 *
 * ent [n_ent, dim], rel [n_rel, dim]; h, r, t [batch]; neg_t [batch, n_neg]; dim <= 512.
 * Indices are trusted to be in range (as torch's embedding gather on device is).
 */
SNAG_API int snag_score_margin_fwd(const float* ent, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin,
                          float* pos, float* neg, float* loss_part, void* stream);

/* Backward of the above (replaces autograd of model/MSNEA.py:110-122).
 * Upstream gradient, one of:
 *   - hinge mode (dpos == dneg == NULL): d loss / d neg[b,k] = -grad_scale * [margin + pos - neg > 0],
 *     d loss / d pos[b] = -sum_k of those; the scores are recomputed in-kernel;
 *   - explicit mode: dpos [batch], dneg [batch, n_neg] given (grad_scale multiplies them).
 * d_ent [n_ent, dim] and d_rel [n_rel, dim] are ACCUMULATED into with fp32 atomics (caller zeroes them).
 */
SNAG_API int snag_score_margin_bwd(const float* ent, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                          int64_t batch, int n_neg, float margin, float grad_scale,
                          const float* dpos, const float* dneg,
                          float* d_ent, float* d_rel, void* stream);

/* Deterministic (fixed-order) sum of n floats into out[0]; used for the scalar loss. */
SNAG_API int snag_reduce_sum(const float* x, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 contraction on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
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
 * keep_mask (optional) is a dense uint8 [M,N] inverted-dropout mask, keep_scale = 1/(1-p).
 * workspace (optional, workspace_floats long) enables deterministic split-K for small-output /
 * long-K shapes; snag_gemm_workspace_floats(M,N,K) is the size that never limits the split.
 */
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
} snag_gemm_desc;

SNAG_API int snag_gemm_f32(const snag_gemm_desc* desc, void* stream);
SNAG_API int64_t snag_gemm_workspace_floats(int M, int N, int K);

#ifdef __cplusplus
}
#endif
#endif /* SNAG_HIP_H */
