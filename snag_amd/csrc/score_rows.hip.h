// Row helpers shared by the scoring kernels (score.hip, score_train.hip): a table row spread over the WHOLE wave
// (col = lane + 64*i), the layout in which one fp32-atomic wave-instruction adds 256 contiguous bytes of a gradient row
// (the full-rate shape, MI355X_MICROARCH.md "Global float atomics").
#pragma once
#include "common.hip.h"

namespace snag {

// score.hip (host): argument checks common to every scoring entry point
int check_score_args(const void* ent, int64_t n_ent, const void* rel, int64_t n_rel, int dim, const void* h, const void* r, const void* t,
                     const void* neg_t, int idx_bytes, int64_t batch, int n_neg);

template <int NS>
struct RowS {
    float v[NS];
};

// table element -> fp32: float tables as they are, bf16 tables (raw bit patterns, BASELINE config 5) widened exactly
__device__ __forceinline__ float elem_to_float(float v) { return v; }
__device__ __forceinline__ float elem_to_float(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

template <int NS, typename ElemT>
__device__ __forceinline__ RowS<NS> load_row_s(const ElemT* __restrict__ base, int64_t row, int64_t ld, int dim, int lane) {
    RowS<NS> r;
    const ElemT* p = base + row * ld;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int c = lane + 64 * i;
        r.v[i] = (c < dim) ? elem_to_float(p[c]) : 0.f;
    }
    return r;
}

// x -> xhat = x / max(||x||, eps); returns the denominator
template <int NS>
__device__ __forceinline__ float normalize_s(RowS<NS>& x) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) ss = fmaf(x.v[i], x.v[i], ss);
    const float den = l2_denominator(sum64(ss));
    const float inv = 1.0f / den;
#pragma unroll
    for (int i = 0; i < NS; ++i) x.v[i] *= inv;
    return den;
}

// backward of F.normalize: dx = (dy - yhat (yhat . dy)) / ||x||   (clamped branch: dx = dy / eps)
// Accumulators: float (fp32 atomics: fast, the last bits depend on the arrival order) or 64-bit fixed point with 32 fractional bits
// (integer atomics: associative, so the sums are bit-reproducible whatever the order; |sum| < 2^31, resolution 2.3e-10).
__device__ __forceinline__ void grad_add(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void grad_add(long long* p, float v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * 4294967296.0));
}

template <int NS, typename AccT>
__device__ __forceinline__ void normalize_bwd_atomic(const RowS<NS>& yhat, float den, const RowS<NS>& dy,
                                                     AccT* __restrict__ grad_row, int dim, int lane) {
    float dot = 0.f;
    if (den > 1e-12f) {
#pragma unroll
        for (int i = 0; i < NS; ++i) dot = fmaf(yhat.v[i], dy.v[i], dot);
        dot = sum64(dot);
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int c = lane + 64 * i;
        if (c < dim) grad_add(grad_row + c, (dy.v[i] - yhat.v[i] * dot) * inv);
    }
}


}  // namespace snag
