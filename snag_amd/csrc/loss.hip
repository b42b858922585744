// SURVEY.md par. 8(f) "next" #1: the MMEA contrastive objective icl_loss (model/SNAG_loss.py:31-128).
//   z1 = nrm(emb)[left], z2 = nrm(emb)[right]
//   logits_a = [z1 z2^T | z1 z1^T - 1e9 I] / tau, logits_b = [z2 z1^T | z2 z2^T - 1e9 I] / tau: all four B x B blocks are
//   the 2B x 2B matrix S = Z Z^T / tau of the stacked views Z = [z1; z2] -- one snag_gemm_f32 launch
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

// S [2B, 2B] = Z Z^T / tau over the stacked views Z = [z1; z2].  Row r < B is the reference's logits_a row r with its two
// column blocks swapped ([aa | ab] instead of [ab | aa], :99-110), row B + r is logits_b ([ba | bb]); softmax does not care
// about column order.  The masked self-similarity is the diagonal (its -1e9 makes exp underflow to exactly 0 in the
// reference), the target is column (r + B) mod 2B.  One block per row, single read: online max/sum per thread.
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ S, int64_t ld, int B, float* __restrict__ lse,
                                                           float* __restrict__ rowloss) {
    __shared__ float red_m[4], red_s[4];
    const int i = blockIdx.x, n = 2 * B;
    const float* row = S + (int64_t)i * ld;
    float mx = -INFINITY, sm = 0.f;
    auto push = [&](float x) {
        const float m2 = fmaxf(mx, x);
        sm = sm * __expf(mx - m2) + __expf(x - m2);
        mx = m2;
    };
    if ((ld & 3) == 0 && (((uintptr_t)S) & 15) == 0) {
        const int n4 = n >> 2;
        for (int q = threadIdx.x; q < n4; q += 256) {
            const float4 v = reinterpret_cast<const float4*>(row)[q];
            const int j = q * 4;
            if (j + 0 != i) push(v.x);
            if (j + 1 != i) push(v.y);
            if (j + 2 != i) push(v.z);
            if (j + 3 != i) push(v.w);
        }
        for (int j = n4 * 4 + threadIdx.x; j < n; j += 256)
            if (j != i) push(row[j]);
    } else {
        for (int j = threadIdx.x; j < n; j += 256)
            if (j != i) push(row[j]);
    }
    const float wm = max64(mx);
    sm = (mx == -INFINITY) ? 0.f : sm * __expf(mx - wm);
    sm = sum64(sm);
    if ((threadIdx.x & 63) == 0) { red_m[threadIdx.x >> 6] = wm; red_s[threadIdx.x >> 6] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
        float t = 0.f;
        for (int k = 0; k < 4; ++k) t += (red_m[k] == -INFINITY) ? 0.f : red_s[k] * expf(red_m[k] - m);
        const float l = m + logf(t);
        lse[i] = l;
        rowloss[i] = l - row[i < B ? i + B : i - B];
    }
}

// in place: S <- G + G^T with G[r,j] = c_r (softmax_r[j] - [j == target(r)]) off the diagonal, 0 on it; c_r = (r < B ? ca : cb) * w[r mod B].
// S is symmetric bit for bit (every entry is the same k-ordered sum), so G^T needs no transposed read:
//   (G + G^T)[r,j] = c_r (e^{S_rj - lse_r} - [j == target(r)]) + c_j (e^{S_rj - lse_j} - [r == target(j)])
// and d loss / d Z = (G + G^T) Z / tau is then ONE product.
__global__ __launch_bounds__(256) void infonce_grad_kernel(float* __restrict__ S, int64_t ld, int B, const float* __restrict__ lse,
                                                           const float* __restrict__ w, float ca, float cb, const float* __restrict__ scale) {
    const int r = blockIdx.x, n = 2 * B;
    if (scale) { ca *= scale[0]; cb *= scale[0]; }
    float* row = S + (int64_t)r * ld;
    const float cr = (r < B ? ca : cb) * (w ? w[r < B ? r : r - B] : 1.0f), lr = lse[r];
    const int tgt = r < B ? r + B : r - B;
    // rows and lse 16-byte addressable, B a multiple of 4 (a quad never straddles the two halves): one dwordx4 per lane
    const bool vec = (ld & 3) == 0 && (B & 3) == 0 && ((((uintptr_t)S) | ((uintptr_t)lse) | ((uintptr_t)w)) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    for (int q = threadIdx.x; q < n4; q += 256) {
        const int j0 = 4 * q;
        const float4 x4 = reinterpret_cast<const float4*>(row)[q], l4 = reinterpret_cast<const float4*>(lse)[q];
        const float4 w4 = w ? reinterpret_cast<const float4*>(w)[(j0 < B ? j0 : j0 - B) >> 2] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float cside = j0 < B ? ca : cb;
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ls[4] = {l4.x, l4.y, l4.z, l4.w}, ws[4] = {w4.x, w4.y, w4.z, w4.w};
        float gs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            const float cj = cside * ws[e];
            float g = cr * __expf(xs[e] - lr) + cj * __expf(xs[e] - ls[e]);
            if (j == tgt) g -= cr + cj;
            gs[e] = (j == r) ? 0.f : g;
        }
        reinterpret_cast<float4*>(row)[q] = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    for (int j = 4 * n4 + threadIdx.x; j < n; j += 256) {
        const float cj = (j < B ? ca : cb) * (w ? w[j < B ? j : j - B] : 1.0f);
        const float x = row[j];
        float g = cr * __expf(x - lr) + cj * __expf(x - lse[j]);
        if (j == tgt) g -= cr + cj;
        row[j] = (j == r) ? 0.f : g;
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

extern "C" SNAG_API int snag_infonce_rows(const float* S, int64_t ld, int B, float* lse, float* rowloss, void* stream) {
    SNAG_REQUIRE(B >= 1 && ld >= 2 * (int64_t)B && S && lse && rowloss, SNAG_ERR_ARG, "infonce_rows: bad arguments");
    hipLaunchKernelGGL(infonce_rows_kernel, dim3(2 * B), dim3(256), 0, (hipStream_t)stream, S, ld, B, lse, rowloss);
    return check_launch("snag_infonce_rows");
}

extern "C" SNAG_API int snag_infonce_grad(float* S, int64_t ld, int B, const float* lse, const float* w, float coef_a, float coef_b,
                                          const float* scale, void* stream) {
    SNAG_REQUIRE(B >= 1 && ld >= 2 * (int64_t)B && S && lse, SNAG_ERR_ARG, "infonce_grad: bad arguments");
    hipLaunchKernelGGL(infonce_grad_kernel, dim3(2 * B), dim3(256), 0, (hipStream_t)stream, S, ld, B, lse, w, coef_a, coef_b, scale);
    return check_launch("snag_infonce_grad");
}
