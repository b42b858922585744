// Row helpers of the segmented scorers (score_seg.hip, score_touched.hip): a table row spread over a 16-lane group, float4 per lane and
// step (column c = sub + 16 j in float4 units), so that one wave-instruction gathers four rows; fp32 tables or bf16 bit patterns.
#pragma once
#include "common.hip.h"

namespace snag {

constexpr double kFix = 4294967296.0;       // fixed point with 32 fractional bits: resolution 2.3e-10, |sum| < 2^31

struct SegEntry {                           // one contribution: weight * rows[src]
    int src;
    float w;
};

template <int NV>
struct Row4 {
    float4 v[NV];
};

template <int NV>
__device__ __forceinline__ Row4<NV> ld_row4(const float* __restrict__ base, int64_t row, int64_t ld, int dim, int sub) {
    Row4<NV> r;
    const float4* p = reinterpret_cast<const float4*>(base + row * ld);
    // (unconditional loads from a clamped column, zeros selected afterwards: no branch around a load, nothing for the compiler to wait on
    // between two loads)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        r.v[j] = p[(c * 4 < dim) ? c : 0];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (!((sub + 16 * j) * 4 < dim)) r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    return r;
}

template <int NV>
__device__ __forceinline__ void st_row4(float* __restrict__ base, int64_t row, int64_t ld, int dim, int sub, const Row4<NV>& r) {
    float4* p = reinterpret_cast<float4*>(base + row * ld);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        if (c * 4 < dim) p[c] = r.v[j];
    }
}

template <int NV>
__device__ __forceinline__ float sumsq4(const Row4<NV>& r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        s = fmaf(r.v[j].x, r.v[j].x, s); s = fmaf(r.v[j].y, r.v[j].y, s);
        s = fmaf(r.v[j].z, r.v[j].z, s); s = fmaf(r.v[j].w, r.v[j].w, s);
    }
    return s;
}

// bf16 table (raw bit patterns): the same column mapping, 8 bytes (4 values) per lane and j -- a 16-lane group covers 128 contiguous bytes
template <int NV>
__device__ __forceinline__ Row4<NV> ld_row4(const uint16_t* __restrict__ base, int64_t row, int64_t ld, int dim, int sub) {
    Row4<NV> r;
    const uint2* p = reinterpret_cast<const uint2*>(base + row * ld);
    uint2 u[NV];
    // (unconditional loads from a clamped column, values selected afterwards: with the load inside the bounds branch the compiler waits
    // for each one before issuing the next -- vmcnt(0) after every load, a wave's 16 rows in flight became one)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        u[j] = p[(c * 4 < dim) ? c : 0];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        const bool ok = c * 4 < dim;
        r.v[j] = make_float4(ok ? __uint_as_float(u[j].x << 16) : 0.f, ok ? __uint_as_float(u[j].x & 0xffff0000u) : 0.f, ok ? __uint_as_float(u[j].y << 16) : 0.f,
                             ok ? __uint_as_float(u[j].y & 0xffff0000u) : 0.f);
    }
    return r;
}

// acc += round(wt * y * 2^32): order-free sums
template <int NV>
__device__ __forceinline__ void fix_add_row(long long (&acc)[4 * NV], const Row4<NV>& y, float wt) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        acc[4 * j + 0] += __double2ll_rn((double)(wt * y.v[j].x) * kFix);
        acc[4 * j + 1] += __double2ll_rn((double)(wt * y.v[j].y) * kFix);
        acc[4 * j + 2] += __double2ll_rn((double)(wt * y.v[j].z) * kFix);
        acc[4 * j + 3] += __double2ll_rn((double)(wt * y.v[j].w) * kFix);
    }
}

// A contiguous run of entries summed in fixed point, 16 at a time: lane i of the group fetches entry i (one coalesced 128-byte read), then
// the rows they name are gathered four at a time -- two dependent round trips per 16 entries instead of two per entry.
// (U rows in flight per step: 4, or 2 where the caller's registers are short)
template <int NV, int U = 4>
__device__ __forceinline__ void fix_add_run(long long (&acc)[4 * NV], const SegEntry* __restrict__ run, unsigned count, const float* __restrict__ rows, int dim,
                                            int sub, int gbase) {
    for (unsigned i0 = 0; i0 < count; i0 += 16) {
        const SegEntry e = (i0 + sub < count) ? run[i0 + sub] : SegEntry{0, 0.f};
        const unsigned m = min(16u, count - i0);
        for (unsigned u0 = 0; u0 < m; u0 += U) {
            Row4<NV> y[U];
            float wt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = __shfl(e.src, gbase + (int)((u0 + u) & 15u));
                wt[u] = (u0 + u < m) ? __shfl(e.w, gbase + (int)((u0 + u) & 15u)) : 0.f;
                y[u] = ld_row4<NV>(rows, (u0 + u < m) ? src : 0, dim, dim, sub);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) fix_add_row<NV>(acc, y[u], wt[u]);
        }
    }
}

// the backward of F.normalize on a row x with upstream sum g:  (g - x^ (x^ . g)) / |x|      (clamped norm: g / eps); g is overwritten
template <int NV>
__device__ __forceinline__ void normalize_bwd4(Row4<NV>& x, Row4<NV>& g) {
    const float den = l2_denominator(sum16(sumsq4<NV>(x)));
    const float inv = 1.0f / den;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        x.v[j].x *= inv; x.v[j].y *= inv; x.v[j].z *= inv; x.v[j].w *= inv;
        dot = fmaf(x.v[j].x, g.v[j].x, dot); dot = fmaf(x.v[j].y, g.v[j].y, dot);
        dot = fmaf(x.v[j].z, g.v[j].z, dot); dot = fmaf(x.v[j].w, g.v[j].w, dot);
    }
    dot = sum16(dot);
    if (!(den > 1e-12f)) dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g.v[j].x = (g.v[j].x - x.v[j].x * dot) * inv; g.v[j].y = (g.v[j].y - x.v[j].y * dot) * inv;
        g.v[j].z = (g.v[j].z - x.v[j].z * dot) * inv; g.v[j].w = (g.v[j].w - x.v[j].w * dot) * inv;
    }
}

}  // namespace snag
