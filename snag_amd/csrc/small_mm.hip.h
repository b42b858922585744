// Parameter-sized dense products (snag_small_matmul): the body is a role of three launches -- its own, the step prologue's and the
// split-K reduce launch that finishes the first projections' weight gradients (snag_gemm_reduce_group_mm).
#pragma once
#include "common.hip.h"

namespace snag {

// ---- parameter-sized dense products (snag_small_matmul; also a role of the step prologue's launch) --------------------------------
struct SmallMMSet {
    snag_small_mm p[4];
    int first[5];                     // block ranges
    int n;
};
// One 16x16 output tile per workgroup, one output per thread; the operands go through LDS in 64-deep k slices, read from memory along
// whichever of their two strides is 1 (a weight and its transpose are both operands here: the one-thread-per-output form with direct
// strided reads took 43 / 87 us per launch of these parameter-sized products).  Sums run in k order: one result whatever the grid.
__device__ __forceinline__ void small_matmul_body(const SmallMMSet& s, int block) {
    const int n = s.n;

    __shared__ float As[64][17], Bs[64][17];
    int q = 0;
    while (q + 1 < n && block >= s.first[q + 1]) ++q;
    const snag_small_mm& P = s.p[q];
    const int tiles_n = (P.N + 15) / 16;
    const int tile = block - s.first[q];
    const int m0 = tile / tiles_n * 16, n0 = tile % tiles_n * 16;
    const int t = threadIdx.x, tm = t >> 4, tn = t & 15;
    float acc = 0.f;
    auto term = [&](const float* __restrict__ A, const float* __restrict__ B) {
        for (int k0 = 0; k0 < P.K; k0 += 64) {
            // unconditional loads at clamped indices, then a select: loads inside a bounds branch are issued one at a time
            float av[4], bv[4];
            int sa[4], sb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = t + 256 * e;
                int mm, ka, nn, kb;
                if (P.sak == 1) { ka = idx & 63; mm = idx >> 6; } else { mm = idx & 15; ka = idx >> 4; }
                if (P.sbk == 1) { kb = idx & 63; nn = idx >> 6; } else { nn = idx & 15; kb = idx >> 4; }
                const bool ina = m0 + mm < P.M && k0 + ka < P.K, inb = n0 + nn < P.N && k0 + kb < P.K;
                av[e] = A[(int64_t)min(m0 + mm, P.M - 1) * P.sam + (int64_t)min(k0 + ka, P.K - 1) * P.sak];
                bv[e] = B[(int64_t)min(k0 + kb, P.K - 1) * P.sbk + (int64_t)min(n0 + nn, P.N - 1) * P.sbn];
                if (!ina) av[e] = 0.f;
                if (!inb) bv[e] = 0.f;
                sa[e] = ka * 17 + mm;
                sb[e] = kb * 17 + nn;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { (&As[0][0])[sa[e]] = av[e]; (&Bs[0][0])[sb[e]] = bv[e]; }
            __syncthreads();
            // rows past K hold zeros: a full slice adds the same sum
#pragma unroll 16
            for (int kk = 0; kk < 64; ++kk) acc = fmaf(As[kk][tm], Bs[kk][tn], acc);
            __syncthreads();
        }
    };
    term(P.A, P.B);
    if (P.A2 != nullptr) term(P.A2, P.B2);
    if (m0 + tm < P.M && n0 + tn < P.N) {
        float* out = P.C + (int64_t)(m0 + tm) * P.ldc + n0 + tn;
        *out = P.accumulate ? *out + acc : acc;
    }
}

// validates the problems and lays out the block ranges; returns the number of workgroups, or a negative error
static inline int fill_small_mm(const snag_small_mm* problems, int n_problems, SmallMMSet& s, const char* who) {
    SNAG_REQUIRE(n_problems >= 0 && n_problems <= 4 && (n_problems == 0 || problems != nullptr), SNAG_ERR_ARG, "%s: 0..4 small products", who);
    int blocks = 0;
    for (int i = 0; i < n_problems; ++i) {
        const snag_small_mm& P = problems[i];
        SNAG_REQUIRE(P.M >= 0 && P.N >= 0 && P.K >= 0 && (P.M == 0 || P.N == 0 || (P.C && P.ldc >= P.N && (P.K == 0 || (P.A && P.B)))), SNAG_ERR_ARG,
                     "%s: bad small product %d", who, i);
        SNAG_REQUIRE((P.A2 == nullptr) == (P.B2 == nullptr), SNAG_ERR_ARG, "%s: small product %d has half a second term", who, i);
        SNAG_REQUIRE((int64_t)P.M * P.N <= (1 << 22), SNAG_ERR_UNSUPPORTED, "%s: small product %d has more than 4M outputs (use snag_gemm_f32)", who, i);
        s.p[i] = P;
        s.first[i] = blocks;
        blocks += ceil_div(P.M, 16) * ceil_div(P.N, 16);
    }
    s.first[n_problems] = blocks;
    s.n = n_problems;
    return blocks;
}

}  // namespace snag
