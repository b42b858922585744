// SURVEY.md par. 8(f) "next" #1: the MMEA contrastive objective icl_loss (model/SNAG_loss.py:31-128).
//   z1 = nrm(emb)[left], z2 = nrm(emb)[right]
//   logits_a = [z1 z2^T | z1 z1^T - 1e9 I] / tau       (the four B x B blocks are snag_gemm_f32 launches)
//   CE_a = -(1/B) sum_i w_i log_softmax(logits_a)[i, i]
// This file holds what is not a GEMM: gather + L2-normalise (fwd/bwd), the masked row log-sum-exp, and the in-place
// transformation of the logits into their gradient.
#include <algorithm>

#include "common.hip.h"

namespace snag {

// z[b,:] = x / max(|x|, 1e-12), x = emb[idx[b], :]; den[b] = max(|x|, 1e-12)   (F.normalize then row gather, :59-64)
template <typename IdxT>
__global__ __launch_bounds__(256) void gather_normalize_fwd_kernel(const float* __restrict__ emb, int64_t ld, int d, const IdxT* __restrict__ idx,
                                                                   int64_t B, int normalize, float* __restrict__ z, float* __restrict__ den) {
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float* x = emb + (int64_t)idx[b] * ld;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) ss = fmaf(x[c], x[c], ss);
    const float dn = normalize ? l2_denominator(sum64(ss)) : 1.0f;
    const float inv = 1.0f / dn;
    for (int c = lane; c < d; c += 64) z[b * d + c] = x[c] * inv;
    if (lane == 0) den[b] = dn;
}

// d_emb[idx[b], :] += (dz - z (z . dz)) / den   (clamped rows: dz / eps), fp32 atomics (an entity may occur in several pairs)
template <typename IdxT>
__global__ __launch_bounds__(256) void gather_normalize_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                                   const float* __restrict__ den, int d, const IdxT* __restrict__ idx, int64_t B,
                                                                   int normalize, float* __restrict__ d_emb, int64_t ld) {
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float dn = den[b];
    float dot = 0.f;
    if (normalize && dn > 1e-12f) {
        for (int c = lane; c < d; c += 64) dot = fmaf(z[b * d + c], dz[b * d + c], dot);
        dot = sum64(dot);
    }
    const float inv = 1.0f / dn;
    float* g = d_emb + (int64_t)idx[b] * ld;
    for (int c = lane; c < d; c += 64) atomicAdd(g + c, (dz[b * d + c] - z[b * d + c] * dot) * inv);
}

// One block per row of logits [B, 2B]: lse over all columns except the masked self-similarity (i, B+i) (its -1e9 makes
// exp underflow to exactly 0 in the reference); rowloss = lse - logits[i, i].
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ L, int64_t ld, int B, float* __restrict__ lse,
                                                           float* __restrict__ rowloss) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const float* row = L + (int64_t)i * ld;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < 2 * B; j += 256)
        if (j != B + i) mx = fmaxf(mx, row[j]);
    mx = max64(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = threadIdx.x; j < 2 * B; j += 256)
        if (j != B + i) s += expf(row[j] - mx);
    s = sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = mx + logf(((red[0] + red[1]) + red[2]) + red[3]);
        lse[i] = l;
        rowloss[i] = l - row[i];
    }
}

// in place: L[i,j] <- c_i (softmax(L)[i,j] - [j == i]), masked entry -> 0;  c_i = coef * (w ? w[i] : 1)
__global__ __launch_bounds__(256) void infonce_grad_kernel(float* __restrict__ L, int64_t ld, int B, const float* __restrict__ lse,
                                                           const float* __restrict__ w, float coef) {
    const int i = blockIdx.x;
    float* row = L + (int64_t)i * ld;
    const float c = coef * (w ? w[i] : 1.0f), l = lse[i];
    for (int j = threadIdx.x; j < 2 * B; j += 256) {
        float p = (j == B + i) ? 0.f : expf(row[j] - l);
        if (j == i) p -= 1.0f;
        row[j] = c * p;
    }
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_gather_normalize_fwd(const float* emb, int64_t ld, int64_t n_rows, int d, const void* idx, int idx_bytes, int64_t B,
                                                  int normalize, float* z, float* den, void* stream) {
    SNAG_REQUIRE(B >= 0 && d >= 1 && ld >= d && n_rows >= 1, SNAG_ERR_ARG, "gather_normalize_fwd: bad sizes");
    if (B == 0) return SNAG_OK;
    SNAG_REQUIRE(emb && idx && z && den, SNAG_ERR_ARG, "gather_normalize_fwd: null pointer");
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "gather_normalize_fwd: idx_bytes=%d", idx_bytes);
    const dim3 grid((unsigned)ceil_div(B, 4));
    hipStream_t st = (hipStream_t)stream;
    if (idx_bytes == 4) hipLaunchKernelGGL((gather_normalize_fwd_kernel<int32_t>), grid, dim3(256), 0, st, emb, ld, d, (const int32_t*)idx, B, normalize, z, den);
    else hipLaunchKernelGGL((gather_normalize_fwd_kernel<int64_t>), grid, dim3(256), 0, st, emb, ld, d, (const int64_t*)idx, B, normalize, z, den);
    return check_launch("snag_gather_normalize_fwd");
}

extern "C" SNAG_API int snag_gather_normalize_bwd(const float* dz, const float* z, const float* den, int d, const void* idx, int idx_bytes, int64_t B,
                                                  int normalize, float* d_emb, int64_t ld, void* stream) {
    SNAG_REQUIRE(B >= 0 && d >= 1 && ld >= d, SNAG_ERR_ARG, "gather_normalize_bwd: bad sizes");
    if (B == 0) return SNAG_OK;
    SNAG_REQUIRE(dz && z && den && idx && d_emb, SNAG_ERR_ARG, "gather_normalize_bwd: null pointer");
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "gather_normalize_bwd: idx_bytes=%d", idx_bytes);
    const dim3 grid((unsigned)ceil_div(B, 4));
    hipStream_t st = (hipStream_t)stream;
    if (idx_bytes == 4) hipLaunchKernelGGL((gather_normalize_bwd_kernel<int32_t>), grid, dim3(256), 0, st, dz, z, den, d, (const int32_t*)idx, B, normalize, d_emb, ld);
    else hipLaunchKernelGGL((gather_normalize_bwd_kernel<int64_t>), grid, dim3(256), 0, st, dz, z, den, d, (const int64_t*)idx, B, normalize, d_emb, ld);
    return check_launch("snag_gather_normalize_bwd");
}

extern "C" SNAG_API int snag_infonce_rows(const float* logits, int64_t ld, int B, float* lse, float* rowloss, void* stream) {
    SNAG_REQUIRE(B >= 1 && ld >= 2 * (int64_t)B && logits && lse && rowloss, SNAG_ERR_ARG, "infonce_rows: bad arguments");
    hipLaunchKernelGGL(infonce_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, B, lse, rowloss);
    return check_launch("snag_infonce_rows");
}

extern "C" SNAG_API int snag_infonce_grad(float* logits, int64_t ld, int B, const float* lse, const float* w, float coef, void* stream) {
    SNAG_REQUIRE(B >= 1 && ld >= 2 * (int64_t)B && logits && lse, SNAG_ERR_ARG, "infonce_grad: bad arguments");
    hipLaunchKernelGGL(infonce_grad_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, B, lse, w, coef);
    return check_launch("snag_infonce_grad");
}
