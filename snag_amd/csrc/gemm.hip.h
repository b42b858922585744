// Exact-fp32 MFMA GEMM core (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 157 TFLOP/s peak).
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )
//
// Each operand is either "K-contiguous" (element (r,k) at ptr[r*ld + k]: activations [rows,K],
// nn.Linear weights W[out,in]) or "row-contiguous" (element (r,k) at ptr[k*ld + r]: the transposed
// uses in the backward GEMMs).  Both land in LDS k-major (tile[k][row]) so that an MFMA operand
// fetch is one conflict-free ds_read_b32 per lane; K-contiguous operands are transposed on the
// way in (row stride = tile + 2 makes the four ds_write_b32 of a float4 conflict-free).
// 256 threads = 4 waves in a 2x2 grid, double-buffered LDS, global loads for tile t+1 in flight
// while tile t feeds the matrix pipe.
#pragma once
#include "common.hip.h"

namespace snag {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// LDS-qualified pointers: through a generic float* the compiler falls back to flat_load/flat_store, which
// count on vmcnt AND lgkmcnt, so every operand fetch would also wait for the global prefetch in flight.
#if defined(__HIP_DEVICE_COMPILE__)
using lds_float = __attribute__((address_space(3))) float;
using lds_float4 = __attribute__((address_space(3))) float4;
#else   // the host pass only needs the code to parse
using lds_float = float;
using lds_float4 = float4;
#endif

constexpr int kBK = 16;

struct GemmOperands {
    const float* A;
    const float* B;
    int64_t lda, ldb;
    int M, N, K;
    int a_vec, b_vec;    // 16-byte loads legal for this operand (base and ld aligned)
    int k_per_split;     // K range handled by one blockIdx.z (multiple of kBK); == K rounded up when no split
    int dbg;             // tuning only: 1 = skip global loads after the first tile, 2 = also skip LDS restaging
};

// --- global -> register staging ---------------------------------------------------------------
// One "slot" is a float4: along k for K-contiguous operands, along the row index otherwise.
template <int BR, bool KC>
struct Stager {
    static constexpr int SLOTS = BR * kBK / 4 / 256;   // float4 per thread per tile
    static_assert(BR * kBK % 1024 == 0, "tile must split evenly over 256 threads");
    float4 r[SLOTS];

    // FULL: the whole k-tile lies inside [kbeg, kend) -> no k predicates; rows past the edge are clamped to the
    // last valid row (their products land in accumulator rows/cols the epilogue discards).
    template <bool FULL>
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int vec_ok, int row0, int nrows, int k0,
                                         int kend, int tid) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = tid + 256 * s;
            int row, k;
            if (KC) {
                row = row0 + slot / (kBK / 4);
                k = k0 + (slot % (kBK / 4)) * 4;
            } else {
                row = row0 + (slot % (BR / 4)) * 4;
                k = k0 + slot / (BR / 4);
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                if (FULL) {
                    const float* g = P + (int64_t)min(row, nrows - 1) * ld + k;
                    if (vec_ok) v = *reinterpret_cast<const float4*>(g);
                    else v = make_float4(g[0], g[1], g[2], g[3]);
                } else if (row < nrows) {
                    const float* g = P + (int64_t)row * ld + k;
                    if (vec_ok && k + 3 < kend) {
                        v = *reinterpret_cast<const float4*>(g);
                    } else {
                        if (k + 0 < kend) v.x = g[0];
                        if (k + 1 < kend) v.y = g[1];
                        if (k + 2 < kend) v.z = g[2];
                        if (k + 3 < kend) v.w = g[3];
                    }
                }
            } else {
                if (FULL || k < kend) {
                    const float* g = P + (int64_t)k * ld + row;
                    if (vec_ok && row + 3 < nrows) {
                        v = *reinterpret_cast<const float4*>(g);
                    } else {
                        if (row + 0 < nrows) v.x = g[0];
                        if (row + 1 < nrows) v.y = g[1];
                        if (row + 2 < nrows) v.z = g[2];
                        if (row + 3 < nrows) v.w = g[3];
                    }
                }
            }
            r[s] = v;
        }
    }

    // tile is k-major: tile[k * LD + row]
    template <int LD>
    __device__ __forceinline__ void store(lds_float* tile, int tid) const {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = tid + 256 * s;
            if (KC) {
                const int row = slot / (kBK / 4);
                const int k = (slot % (kBK / 4)) * 4;
                tile[(k + 0) * LD + row] = r[s].x;
                tile[(k + 1) * LD + row] = r[s].y;
                tile[(k + 2) * LD + row] = r[s].z;
                tile[(k + 3) * LD + row] = r[s].w;
            } else {
                const int row = (slot % (BR / 4)) * 4;
                const int k = slot / (BR / 4);
                *reinterpret_cast<lds_float4*>(tile + k * LD + row) = r[s];
            }
        }
    }
};

template <int BR, bool KC>
struct TileLd {
    static constexpr int value = KC ? BR + 2 : BR + 4;
};

// Main loop.  acc[mi][ni] is the 32x32 tile (mi, ni) of this wave's WM x WN patch:
//   row = wm0 + 32*mi + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5),  col = wn0 + 32*ni + (lane & 31).
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
struct GemmCore {
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int LDA = TileLd<BM, A_KC>::value, LDB = TileLd<BN, B_KC>::value;
    static constexpr int TILE_FLOATS = 2 * kBK * (LDA + LDB);
    static constexpr int STAGE_FLOATS = 4 * 32 * WN;   // epilogue staging, 32 rows x WN per wave
    static constexpr int SMEM_FLOATS = TILE_FLOATS > STAGE_FLOATS ? TILE_FLOATS : STAGE_FLOATS;
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per block");

    f32x16 acc[TM][TN];

    __device__ __forceinline__ void run(const GemmOperands& p, float* __restrict__ smem, int m0, int n0, int kbeg, int kend) {
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = wave_id_uniform();
        const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
        lds_float* const sm = (lds_float*)smem;
        constexpr int A_BUF = kBK * LDA, B_BUF = kBK * LDB, B_BASE = 2 * kBK * LDA;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        Stager<BM, A_KC> sa;
        Stager<BN, B_KC> sb;
        const int nk = (kend - kbeg + kBK - 1) / kBK;
        if (nk <= 0) return;
        if (kbeg + kBK <= kend) {
            sa.template load<true>(p.A, p.lda, p.a_vec, m0, p.M, kbeg, kend, tid);
            sb.template load<true>(p.B, p.ldb, p.b_vec, n0, p.N, kbeg, kend, tid);
        } else {
            sa.template load<false>(p.A, p.lda, p.a_vec, m0, p.M, kbeg, kend, tid);
            sb.template load<false>(p.B, p.ldb, p.b_vec, n0, p.N, kbeg, kend, tid);
        }
        sa.template store<LDA>(sm, tid);
        sb.template store<LDB>(sm + B_BASE, tid);
        __syncthreads();
        const int li = lane & 31, lh = lane >> 5;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk && p.dbg == 0) {
                const int k0 = kbeg + (kt + 1) * kBK;
                if (k0 + kBK <= kend) {             // wave-uniform: only the last tile of a ragged K takes the slow path
                    sa.template load<true>(p.A, p.lda, p.a_vec, m0, p.M, k0, kend, tid);
                    sb.template load<true>(p.B, p.ldb, p.b_vec, n0, p.N, k0, kend, tid);
                } else {
                    sa.template load<false>(p.A, p.lda, p.a_vec, m0, p.M, k0, kend, tid);
                    sb.template load<false>(p.B, p.ldb, p.b_vec, n0, p.N, k0, kend, tid);
                }
            }
            const lds_float* a = sm + cur * A_BUF + lh * LDA + wm0 + li;
            const lds_float* b = sm + B_BASE + cur * B_BUF + lh * LDB + wn0 + li;
            // all operand fragments of the k-tile first (one burst of ds_reads), then the MFMA chain
            float av[kBK / 2][TM], bv[kBK / 2][TN];
#pragma unroll
            for (int s = 0; s < kBK / 2; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[s][i] = a[2 * s * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[s][j] = b[2 * s * LDB + 32 * j];
            }
#pragma unroll
            for (int s = 0; s < kBK / 2; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][i], bv[s][j], acc[i][j], 0, 0, 0);
            if (kt + 1 < nk && p.dbg < 2) {
                sa.template store<LDA>(sm + (cur ^ 1) * A_BUF, tid);
                sb.template store<LDB>(sm + B_BASE + (cur ^ 1) * B_BUF, tid);
            }
            __syncthreads();
        }
    }

    // Visit the block tile four columns at a time: f(row, col, float4).  The accumulators take a trip through
    // LDS (the tile buffers are free by now) so that 16 lanes cover 256 contiguous bytes of one output row:
    // the epilogue's loads and the C store are dwordx4 and fully coalesced, and the accumulator indices stay
    // static (a runtime-indexed f32x16 would be demoted to scratch).
    template <class F>
    __device__ __forceinline__ void for_each(float* __restrict__ smem, int m0, int n0, F&& f) const {
        static_assert(WN == 64, "sixteen float4 lanes per row of the wave patch");
        static_assert(4 * 32 * WN <= SMEM_FLOATS, "staging fits in the tile buffers");
        const int lane = threadIdx.x & 63;
        const int wave = wave_id_uniform();
        const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
        lds_float* stage = (lds_float*)smem + wave * (32 * WN);
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * WN + 32 * j + li] = acc[i][j][e];
            __syncthreads();
            const int row0 = m0 + wm0 + 32 * i + (lane >> 4);
            const int col = n0 + wn0 + (lane & 15) * 4;
#pragma unroll 2
            for (int r = 0; r < 32; r += 4) {
                const float4 v = *reinterpret_cast<const lds_float4*>(stage + (r + (lane >> 4)) * WN + (lane & 15) * 4);
                f(row0 + r, col, v);
            }
        }
    }
};

// One block = one BM x BN output tile (x one K split): main loop, then either the epilogue functor
// epi(row, col, float4) or, for split-K, a raw store into this split's slab.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, class Epi>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmOperands p, Epi epi, float* __restrict__ slab) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    Core core;
    core.run(p, smem, m0, n0, kbeg, kend);
    if (slab == nullptr) {
        core.for_each(smem, m0, n0, epi);
    } else {
        float* s = slab + (int64_t)blockIdx.z * p.M * p.N;
        const int M = p.M, N = p.N;
        const bool vec = (N % 4) == 0;
        core.for_each(smem, m0, n0, [=](int m, int n, float4 v) {
            if (m >= M || n >= N) return;
            float* o = s + (int64_t)m * N + n;
            if (vec) {
                *reinterpret_cast<float4*>(o) = v;
            } else {
                o[0] = v.x;
                if (n + 1 < N) o[1] = v.y;
                if (n + 2 < N) o[2] = v.z;
                if (n + 3 < N) o[3] = v.w;
            }
        });
    }
}

}  // namespace snag
