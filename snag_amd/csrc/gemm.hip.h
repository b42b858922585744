// Exact-fp32 MFMA GEMM core (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 157 TFLOP/s peak).
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )
//
// Each operand is either "K-contiguous" (element (r,k) at ptr[r*ld + k]: activations [rows,K],
// nn.Linear weights W[out,in]) or "row-contiguous" (element (r,k) at ptr[k*ld + r]: the transposed
// uses in the backward GEMMs).  A float4 fetched from global memory goes to LDS as ONE ds_write_b128:
//   * K-contiguous: the tile keeps its row-major shape, [row][16 k] = four 16-byte chunks per row, with the chunk
//     index XOR-swizzled by (row >> 2) & 3.  A lane fetches its MFMA operands as ds_read_b128: lane half h reads
//     chunk 2cp + h of its row and feeds four consecutive MFMA k-steps from it, i.e. within a k-tile the sum runs
//     over k in the order 8cp + 4h + j -- a permutation of k, applied identically to both operands (the swizzle
//     makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots: conflict-free);
//   * row-contiguous: the tile is [k][row]; operands are conflict-free ds_read_b32 at the same permuted k.
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

#ifndef SNAG_GEMM_BK
#define SNAG_GEMM_BK 16
#endif
constexpr int kBK = SNAG_GEMM_BK;          // k-tile depth: 16, or 32 (A/B switch: half the barriers per MFMA, twice the LDS)
static_assert(kBK == 16 || kBK == 32, "k-tile depth");
constexpr int kCPR = kBK / 4;               // 16-byte chunks per K-contiguous tile row
// XOR swizzle of the chunk index: 16 consecutive rows x one chunk column must hit 16 distinct 16-byte bank slots
__host__ __device__ constexpr int swz(int row) { return kBK == 16 ? ((row >> 2) & 3) : ((row >> 1) & 7); }

// Build-time structure switches (A/B'ed on one device):
//   SNAG_GEMM_SWZ  1: K-contiguous tiles stay row-major in LDS with XOR-swizzled 16-byte chunks (b128 in, b128 out)
//                  0: K-contiguous tiles are transposed to k-major on the way in (4x ds_write_b32, b32 fragment reads)
//   SNAG_GEMM_PIPE 1: software pipeline (fragments of tile t+1 and global loads of tile t+2 issued before the MFMAs of t)
//                  0: load fragments, MFMA chain, stage tile t+1
#ifndef SNAG_GEMM_SWZ
#define SNAG_GEMM_SWZ 1
#endif
#ifndef SNAG_GEMM_PIPE
#define SNAG_GEMM_PIPE 0
#endif
//   SNAG_GEMM_DMA  1: the unpredicated (VEC) kernels fetch full k-tiles with global_load_lds_dwordx4 (LDS-DMA: no VGPR
//                  staging, no ds_write); needs SNAG_GEMM_SWZ=1 (dense tiles: the DMA writes 64 lanes x 16 B linearly)
#ifndef SNAG_GEMM_DMA
#define SNAG_GEMM_DMA 1
#endif
//   SNAG_GEMM_BF16X 0: v_mfma_f32_32x32x2_f32 on the f32 operands (157 TF peak)
//                   6: every operand value is split EXACTLY into three bf16 pieces a = h + m + l (round-to-nearest each time, the
//                      two remainders are exact in f32) in registers, and a product a*b becomes the six bf16 MFMA products with weight
//                      >= 2^-16: h*h, h*m, m*h, m*m, h*l, l*h (v_mfma_f32_32x32x16_bf16, f32 accumulation; the three dropped
//                      products m*l, l*m, l*l are below 2^-23 of a*b -- the size of f32's own product rounding).  The bf16 matrix
//                      rate is 16x the f32 one, so six products still cost 3/8 of the f32 MFMA time.
//                   9: all nine products
#ifndef SNAG_GEMM_BF16X
#define SNAG_GEMM_BF16X 6
#endif
constexpr int kBf16x = SNAG_GEMM_BF16X;
static_assert(kBf16x == 0 || kBf16x == 6 || kBf16x == 9, "SNAG_GEMM_BF16X");
constexpr bool kDma = (SNAG_GEMM_DMA != 0) && (SNAG_GEMM_SWZ != 0);
constexpr bool kSwz = SNAG_GEMM_SWZ != 0;
constexpr bool kPipe = SNAG_GEMM_PIPE != 0;

// LDS leading dimension of a tile: swizzled row-major tiles are dense; k-major tiles are padded (+2 makes the four
// transposing ds_write_b32 of a float4 conflict-free, +4 keeps ds_write_b128 rows 16-byte aligned)
template <int BR, bool KC>
struct TileLd {
    static constexpr int value = KC ? (kSwz ? kBK : BR + 2) : (kSwz ? BR : BR + 4);
    static constexpr int floats = KC ? (kSwz ? BR * kBK : kBK * (BR + 2)) : (kSwz ? kBK * BR : kBK * (BR + 4));
};

struct GemmOperands {
    const float* A;
    const float* B;
    int64_t lda, ldb;
    int M, N, K;
    int a_vec, b_vec;    // 16-byte loads legal for this operand (base and ld aligned)
    // the VEC kernel variant additionally needs: row-contiguous operands with a row count that is a multiple of 4
    int k_per_split;     // K range handled by one blockIdx.z (multiple of kBK); == K rounded up when no split
    int dbg;             // tuning only: nonzero = stop fetching after the first two k-tiles (compute-only ceiling)
    int xcd;             // 1: XCD-aware workgroup -> tile mapping (xcd_swizzle)
};

// Workgroups are dealt round-robin over the 8 XCDs (observed, never relied on for correctness): ids b and b + 8 share an L2.
// The bijective remap below gives each of the 8 groups a CONTIGUOUS range of tile indices, so tiles that share an operand panel
// -- and, for row pitches that are not a multiple of 128 B, the cache lines straddling a tile boundary -- stay in one L2.
__device__ __forceinline__ int xcd_swizzle(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = orig & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (orig >> 3);
}

// --- global -> register staging ---------------------------------------------------------------
// One "slot" is a float4: along k for K-contiguous operands, along the row index otherwise.
template <int BR, bool KC>
struct Stager {
    static constexpr int TOTAL = BR * kBK / 4;                 // float4 slots per tile
    static constexpr int SLOTS = (TOTAL + 255) / 256;          // per thread; the last one may be half used (BR = 160: waves 0-1 only)
    static_assert(TOTAL % 128 == 0, "a partly used slot must still split on a wave boundary");
    static constexpr bool used(int s, int slot) { return (s + 1) * 256 <= TOTAL || slot < TOTAL; }
    float4 r[SLOTS];
    const float* base[SLOTS];      // per-slot address of k = 0, rows clamped (hot-path loader only)

    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = min(tid + 256 * s, TOTAL - 1);
            if (KC) {
                const int row = min(row0 + slot / (kBK / 4), nrows - 1);
                base[s] = P + (int64_t)row * ld + (slot % (kBK / 4)) * 4;
            } else {
                const int row = max(0, min(row0 + (slot % (BR / 4)) * 4, nrows - 4));
                base[s] = P + (int64_t)(slot / (BR / 4)) * ld + row;
            }
        }
    }

    // LDS-DMA fetch of one full k-tile: lane L of wave w writes the 16-byte chunk at tile position 256*s + 64*w + L, so
    // it must READ the chunk that belongs there.  Row-contiguous tiles are linear in slot order already; K-contiguous
    // tiles store chunk c of a row at position c ^ ((row >> 2) & 3), an involution, so position p holds chunk p ^ swz.
    __device__ __forceinline__ void init_dma(const float* __restrict__ P, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = min(tid + 256 * s, TOTAL - 1);
            if (KC) {
                const int lrow = slot / (kBK / 4);
                const int c = (slot % (kBK / 4)) ^ swz(lrow);
                base[s] = P + (int64_t)min(row0 + lrow, nrows - 1) * ld + 4 * c;
            } else {
                const int row = max(0, min(row0 + (slot % (BR / 4)) * 4, nrows - 4));
                base[s] = P + (int64_t)(slot / (BR / 4)) * ld + row;
            }
        }
    }

    __device__ __forceinline__ void dma(lds_float* tile, int64_t ld, int k0, int wave) const {
        [[maybe_unused]] const int64_t off = KC ? (int64_t)k0 : (int64_t)k0 * ld;      // (the host pass compiles the body away)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (!used(s, 256 * s + 64 * wave)) continue;        // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base[s] + off),
                                             (__attribute__((address_space(3))) void*)(tile + (256 * s + 64 * wave) * 4), 16, 0, 0);
#endif
        }
    }

    // FULL: the whole k-tile lies inside [kbeg, kend) -> no k predicates; rows past the edge are clamped to the
    // last valid row (their products land in accumulator rows/cols the epilogue discards).
    template <bool FULL, bool VEC>
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int vec_ok, int row0, int nrows, int k0,
                                         int kend, int tid) {
        if (FULL && VEC) {
            // hot path: whole tile in range along k, 16-byte loads legal (decided on the host): no predicates, and the
            // only per-tile address arithmetic is one wave-uniform offset.  Rows past the edge were clamped in init();
            // their products land in accumulator rows/cols the epilogue never stores.
            const int64_t off = KC ? (int64_t)k0 : (int64_t)k0 * ld;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (used(s, tid + 256 * s)) r[s] = *reinterpret_cast<const float4*>(base[s] + off);
            return;
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = tid + 256 * s;
            if (!used(s, slot)) continue;
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

    // SWZ & KC: tile[row][16] with swizzled 16-byte chunks; otherwise tile[k][LD]
    __device__ __forceinline__ void store(lds_float* tile, int tid) const {
        constexpr int LD = TileLd<BR, KC>::value;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = tid + 256 * s;
            if (!used(s, slot)) continue;
            if (KC) {
                const int row = slot / (kBK / 4);
                const int c = slot % (kBK / 4);
                if (kSwz) {
                    *reinterpret_cast<lds_float4*>(tile + row * kBK + 4 * (c ^ swz(row))) = r[s];
                } else {
                    tile[(4 * c + 0) * LD + row] = r[s].x;
                    tile[(4 * c + 1) * LD + row] = r[s].y;
                    tile[(4 * c + 2) * LD + row] = r[s].z;
                    tile[(4 * c + 3) * LD + row] = r[s].w;
                }
            } else {
                const int row = (slot % (BR / 4)) * 4;
                const int k = slot / (BR / 4);
                *reinterpret_cast<lds_float4*>(tile + k * LD + row) = r[s];
            }
        }
    }
};

// operand fragments of one k-tile for NF 32-row fragments: f[t][n] feeds MFMA k-step t.
// SWZ: t = 4*cp + j <-> k = 8cp + 4h + j (a permutation of k shared by both operands); else k = 2t + h.
template <int BR, bool KC, int NF>
__device__ __forceinline__ void load_fragments(const lds_float* tile, int row0, int li, int lh, float (&f)[kBK / 2][NF]) {
    constexpr int LD = TileLd<BR, KC>::value;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int row = row0 + 32 * n + li;
        if (kSwz) {
#pragma unroll
            for (int cp = 0; cp < kBK / 8; ++cp) {
                if (KC) {
                    const int c = 2 * cp + lh;
                    const float4 v = *reinterpret_cast<const lds_float4*>(tile + row * kBK + 4 * (c ^ swz(li)));
                    f[4 * cp + 0][n] = v.x;
                    f[4 * cp + 1][n] = v.y;
                    f[4 * cp + 2][n] = v.z;
                    f[4 * cp + 3][n] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[4 * cp + j][n] = tile[(8 * cp + 4 * lh + j) * LD + row];
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < kBK / 2; ++t) f[t][n] = tile[(2 * t + lh) * LD + row];
        }
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
#else
struct bf16x8 { unsigned short v[8]; };
#endif
// a = h + m + l exactly (8 + 8 + 8 significand bits); f[j] is the lane's value for element j of the 32x32x16 operand
struct Split3 {
    bf16x8 h, m, l;
    __device__ __forceinline__ void from(const float (&f)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 hj = (__bf16)f[j];
            const float r1 = f[j] - (float)hj;
            const __bf16 mj = (__bf16)r1;
            const float r2 = r1 - (float)mj;
            h[j] = hj;
            m[j] = mj;
            l[j] = (__bf16)r2;
        }
#endif
    }
};

// Main loop.  acc[mi][ni] is the 32x32 tile (mi, ni) of this wave's WM x WN patch:
//   row = wm0 + 32*mi + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5),  col = wn0 + 32*ni + (lane & 31).
// SYM (split-bf16 builds): C(a, b) == C(b, a) bit for bit.  The six products are not symmetric one by one (h*m <-> m*h, h*l <-> l*h
// under an exchange of the operands) and f32 addition is not associative, so the two mirrored families accumulate in two extra
// accumulator sets that exchange roles with the operands, and meet in one commutative addition at the end.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, bool SYM = false>
struct GemmCore {
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_BUF = TileLd<BM, A_KC>::floats, B_BUF = TileLd<BN, B_KC>::floats, B_BASE = 2 * A_BUF;
    static constexpr int TILE_FLOATS = 2 * (A_BUF + B_BUF);
    static constexpr int CW = (WN % 64 == 0) ? 64 : 32;   // epilogue staging chunk: 32 rows x CW columns per wave at a time
    static constexpr int STAGE_FLOATS = 4 * 32 * CW;
    static constexpr int SMEM_FLOATS = TILE_FLOATS > STAGE_FLOATS ? TILE_FLOATS : STAGE_FLOATS;
    static_assert((BM / WM) * (BN / WN) == 4, "four waves per block");

    f32x16 acc[TM][TN];
    static constexpr bool kSym = SYM && kBf16x != 0;
    f32x16 accx[kSym ? TM : 1][kSym ? TN : 1], accy[kSym ? TM : 1][kSym ? TN : 1];      // (a small, b large) / (a large, b small)

    // one k-tile's worth of matrix work: av[s][i] / bv[s][j] are this lane's operand values of k-step s
    __device__ __forceinline__ void mfma_chain(const float (&av)[kBK / 2][TM], const float (&bv)[kBK / 2][TN]) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kBf16x == 0) {
#pragma unroll
            for (int s = 0; s < kBK / 2; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][i], bv[s][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int g = 0; g < kBK / 16; ++g) {
                Split3 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = av[8 * g + e][i];
                    a[i].from(f);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = bv[8 * g + e][j];
                    b[j].from(f);
                }
                // smallest products first; one pass over the accumulators per product keeps dependent MFMAs apart
#define SNAG_MFMA_TERM(ACC, X, Y)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                     \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].X, b[j].Y, ACC[i][j], 0, 0, 0);
                if constexpr (kSym) {
                    if constexpr (kBf16x == 9) {
                        SNAG_MFMA_TERM(acc, l, l)
                        SNAG_MFMA_TERM(accx, l, m)
                        SNAG_MFMA_TERM(accy, m, l)
                    }
                    SNAG_MFMA_TERM(accx, l, h)
                    SNAG_MFMA_TERM(accy, h, l)
                    SNAG_MFMA_TERM(acc, m, m)
                    SNAG_MFMA_TERM(accx, m, h)
                    SNAG_MFMA_TERM(accy, h, m)
                    SNAG_MFMA_TERM(acc, h, h)
                } else {
                    if constexpr (kBf16x == 9) {
                        SNAG_MFMA_TERM(acc, l, l)
                        SNAG_MFMA_TERM(acc, l, m)
                        SNAG_MFMA_TERM(acc, m, l)
                    }
                    SNAG_MFMA_TERM(acc, l, h)
                    SNAG_MFMA_TERM(acc, h, l)
                    SNAG_MFMA_TERM(acc, m, m)
                    SNAG_MFMA_TERM(acc, m, h)
                    SNAG_MFMA_TERM(acc, h, m)
                    SNAG_MFMA_TERM(acc, h, h)
                }
#undef SNAG_MFMA_TERM
            }
        }
#endif
    }

    __device__ __forceinline__ void run(const GemmOperands& p, float* __restrict__ smem, int m0, int n0, int kbeg, int kend) {
        if constexpr (kSym) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) accx[i][j][e] = accy[i][j][e] = 0.f;
        }
        run_loop(p, smem, m0, n0, kbeg, kend);
        if constexpr (kSym) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += accx[i][j][e] + accy[i][j][e];
        }
    }

    __device__ __forceinline__ void run_loop(const GemmOperands& p, float* __restrict__ smem, int m0, int n0, int kbeg, int kend) {
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = wave_id_uniform();
        const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
        lds_float* const sm = (lds_float*)smem;
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
        const int li = lane & 31, lh = lane >> 5;
        if (VEC) {
            sa.init(p.A, p.lda, m0, p.M, tid);
            sb.init(p.B, p.ldb, n0, p.N, tid);
        }
        // wave-uniform choice between the predicate-free loader and the ragged-K one (last tile only)
        auto fetch = [&](int kt) {
            const int k0 = kbeg + kt * kBK;
            if (k0 + kBK <= kend) {
                sa.template load<true, VEC>(p.A, p.lda, p.a_vec, m0, p.M, k0, kend, tid);
                sb.template load<true, VEC>(p.B, p.ldb, p.b_vec, n0, p.N, k0, kend, tid);
            } else {
                sa.template load<false, false>(p.A, p.lda, p.a_vec, m0, p.M, k0, kend, tid);
                sb.template load<false, false>(p.B, p.ldb, p.b_vec, n0, p.N, k0, kend, tid);
            }
        };
        auto stage = [&](int buf) {
            sa.store(sm + buf * A_BUF, tid);
            sb.store(sm + B_BASE + buf * B_BUF, tid);
        };

        float av[kBK / 2][TM], bv[kBK / 2][TN];
        if (kDma && VEC) {
            // LDS-DMA main loop: the copy of tile kt+1 runs while tile kt feeds the matrix pipe; the only wait is the
            // vmcnt(0) in front of the barrier that publishes it.  A ragged last tile takes the register path.
            sa.init_dma(p.A, p.lda, m0, p.M, tid);
            sb.init_dma(p.B, p.ldb, n0, p.N, tid);
            auto full = [&](int kt) { return kbeg + (kt + 1) * kBK <= kend; };
            auto bring = [&](int kt, int buf) {        // returns true when the tile went through registers
                if (full(kt)) {
                    sa.dma(sm + buf * A_BUF, p.lda, kbeg + kt * kBK, wave);
                    sb.dma(sm + B_BASE + buf * B_BUF, p.ldb, kbeg + kt * kBK, wave);
                    return false;
                }
                sa.template load<false, false>(p.A, p.lda, p.a_vec, m0, p.M, kbeg + kt * kBK, kend, tid);
                sb.template load<false, false>(p.B, p.ldb, p.b_vec, n0, p.N, kbeg + kt * kBK, kend, tid);
                return true;
            };
            if (bring(0, 0)) stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                const bool more = (kt + 1 < nk) && p.dbg == 0;
                bool via_regs = false;
                if (more) via_regs = bring(kt + 1, cur ^ 1);
                load_fragments<BM, A_KC, TM>(sm + cur * A_BUF, wm0, li, lh, av);
                load_fragments<BN, B_KC, TN>(sm + B_BASE + cur * B_BUF, wn0, li, lh, bv);
                mfma_chain(av, bv);
                if (via_regs) stage(cur ^ 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (p.dbg != 2) __syncthreads();       // dbg 2: ablation without the per-k-tile barrier (results are wrong)
            }
            return;
        }
        if (!kPipe) {
            fetch(0);
            stage(0);
            __syncthreads();
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                const bool more = (kt + 1 < nk) && p.dbg == 0;
                if (more) fetch(kt + 1);
                load_fragments<BM, A_KC, TM>(sm + cur * A_BUF, wm0, li, lh, av);
                load_fragments<BN, B_KC, TN>(sm + B_BASE + cur * B_BUF, wn0, li, lh, bv);
                mfma_chain(av, bv);
                if (more) stage(cur ^ 1);
                __syncthreads();
            }
            return;
        }
        // Software pipeline, one barrier per k-tile.  At the top of iteration kt: av/bv hold the operand fragments of
        // tile kt, LDS buffer (kt+1)&1 holds tile kt+1.  The iteration issues the global loads of tile kt+2 and the LDS
        // fragment reads of tile kt+1 BEFORE its MFMA chain, so a wave covers both latencies with its own matrix work
        // instead of relying on other waves; tile kt+2 is written to LDS after the chain.
        fetch(0);
        stage(0);
        __syncthreads();
        load_fragments<BM, A_KC, TM>(sm, wm0, li, lh, av);
        load_fragments<BN, B_KC, TN>(sm + B_BASE, wn0, li, lh, bv);
        if (nk > 1) {
            fetch(1);
            stage(1);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int nxt = (kt + 1) & 1;
            const bool more2 = (kt + 2 < nk) && p.dbg == 0;
            if (more2) fetch(kt + 2);
            float an[kBK / 2][TM], bn[kBK / 2][TN];
            if (kt + 1 < nk) {
                load_fragments<BM, A_KC, TM>(sm + nxt * A_BUF, wm0, li, lh, an);
                load_fragments<BN, B_KC, TN>(sm + B_BASE + nxt * B_BUF, wn0, li, lh, bn);
            }
            mfma_chain(av, bv);
            if (more2) stage(kt & 1);
            __syncthreads();
            if (kt + 1 < nk) {
#pragma unroll
                for (int s = 0; s < kBK / 2; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[s][i] = an[s][i];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[s][j] = bn[s][j];
                }
            }
        }
    }

    // Visit the block tile four columns at a time: f(row, col, float4).  The accumulators take a trip through
    // LDS (the tile buffers are free by now) so that 16 lanes cover 256 contiguous bytes of one output row:
    // the epilogue's loads and the C store are dwordx4 and fully coalesced, and the accumulator indices stay
    // static (a runtime-indexed f32x16 would be demoted to scratch).
    template <class F>
    __device__ __forceinline__ void for_each(float* __restrict__ smem, int m0, int n0, F&& f) const {
        static_assert(WN % CW == 0 && STAGE_FLOATS <= SMEM_FLOATS, "staging fits in the tile buffers");
        constexpr int LPR = CW / 4;            // lanes per staged row (float4 each)
        constexpr int RPP = 64 / LPR;          // rows per pass
        const int lane = threadIdx.x & 63;
        const int wave = wave_id_uniform();
        const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
        lds_float* stage = (lds_float*)smem + wave * (32 * CW);
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int c = 0; c < WN / CW; ++c) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < CW / 32; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * CW + 32 * j + li] = acc[i][c * (CW / 32) + j][e];
                __syncthreads();
                const int row0 = m0 + wm0 + 32 * i + lane / LPR;
                const int col = n0 + wn0 + c * CW + (lane % LPR) * 4;
#pragma unroll 2
                for (int r = 0; r < 32; r += RPP) {
                    const float4 v = *reinterpret_cast<const lds_float4*>(stage + (r + lane / LPR) * CW + (lane % LPR) * 4);
                    f(row0 + r, col, v);
                }
            }
        }
    }

    // Visit the block tile TRANSPOSED, one element per lane: f(col, row, value) with the 32 lanes of a half-wave on 32 consecutive
    // rows of one column, i.e. 128 contiguous bytes of the transposed matrix per half-wave.  The accumulators go through LDS with a
    // padded pitch (CW + 1: the column-wise reads are conflict-free).  smem must hold kTransposeStageFloats.
    static constexpr int kTransposeStageFloats = 4 * 32 * (CW + 1);
    template <class F>
    __device__ __forceinline__ void for_each_transposed(float* __restrict__ smem, int m0, int n0, F&& f) const {
        constexpr int P = CW + 1;
        const int lane = threadIdx.x & 63;
        const int wave = wave_id_uniform();
        const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
        lds_float* stage = (lds_float*)smem + wave * (32 * P);
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int c = 0; c < WN / CW; ++c) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < CW / 32; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * P + 32 * j + li] = acc[i][c * (CW / 32) + j][e];
                __syncthreads();
                const int row = m0 + wm0 + 32 * i + li;
#pragma unroll 4
                for (int it = 0; it < CW / 2; ++it) {
                    const int cc = 2 * it + lh;
                    f(n0 + wn0 + c * CW + cc, row, stage[li * P + cc]);
                }
            }
        }
    }
};

// One block = one BM x BN output tile (x one K split): main loop, then either the epilogue functor
// epi(row, col, float4) or, for split-K, a raw store into this split's slab (split index z).
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__device__ __forceinline__ void gemm_tile_body(const GemmOperands& p, const Epi& epi, float* __restrict__ slab, int z, float* smem) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd) {
        const int id = xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        bx = id % gridDim.x;
        by = id / gridDim.x;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    Core core;
    core.run(p, smem, m0, n0, kbeg, kend);
    if (slab == nullptr) {
        core.for_each(smem, m0, n0, epi);
    } else {
        float* s = slab + (int64_t)z * p.M * p.N;
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

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmOperands p, Epi epi, float* __restrict__ slab) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    gemm_tile_body<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p, epi, slab, blockIdx.z, smem);
}

// TWO problems of the same shape and layout in one launch (blockIdx.z = problem * splits + split): the step's many
// [15000 x 128 x 128]-sized GEMMs are launch / prologue-bound grids of 235 workgroups (< 1 per CU); a pair of them --
// the two modalities' second projections, their backward, the two D x D weight gradients of the read-out token --
// fills the chip once instead of twice.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(256) void gemm_tile_pair_kernel(GemmOperands p0, Epi e0, float* __restrict__ slab0, GemmOperands p1, Epi e1,
                                                             float* __restrict__ slab1, int splits) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    const int z = blockIdx.z % splits;
    if (blockIdx.z < splits) gemm_tile_body<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p0, e0, slab0, z, smem);
    else gemm_tile_body<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p1, e1, slab1, z, smem);
}

// Stream-K: G persistent blocks (512 = two per CU by default) share the (tile, k-tile) iteration space evenly, so every CU gets the same MFMA work whatever
// tiles x splits would have been modulo the CU count (235 tiles x 5 splits = 4.59 blocks per CU otherwise runs at the pace of
// the CUs that got 5).  Block b owns iterations [total*b/G, total*(b+1)/G) of the tile-major order; a tile it covers completely
// goes straight through the epilogue, a partial one is parked as a raw BM x BN tile in slab[(b*max_seg + seg)] and summed in
// block order by streamk_fixup_kernel (deterministic).
struct StreamKPlan {
    int tiles_n;        // output tiles along N
    int nkt;            // k-tiles per output tile
    int64_t total;      // tiles * nkt
    int max_seg;        // slab slots per block
};

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(256) void gemm_streamk_kernel(GemmOperands p, Epi epi, float* __restrict__ slab, StreamKPlan plan) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    const int64_t G = gridDim.x, b = p.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;   // virtual block id: slab slots and the fix-up use it too
    int64_t it = plan.total * b / G;
    const int64_t it_end = plan.total * (b + 1) / G;
    Core core;
    for (int seg = 0; it < it_end; ++seg) {
        const int tile = (int)(it / plan.nkt), kt0 = (int)(it % plan.nkt);
        const int kt1 = (int)min<int64_t>(plan.nkt, kt0 + (it_end - it));
        const int m0 = (tile / plan.tiles_n) * BM, n0 = (tile % plan.tiles_n) * BN;
        core.run(p, smem, m0, n0, kt0 * kBK, min(p.K, kt1 * kBK));
        if (kt0 == 0 && kt1 == plan.nkt) {
            core.for_each(smem, m0, n0, epi);
        } else {
            float* dst = slab + ((int64_t)b * plan.max_seg + seg) * (BM * BN);
            core.for_each(smem, m0, n0, [=](int m, int n, float4 v) { *reinterpret_cast<float4*>(dst + (m - m0) * BN + (n - n0)) = v; });
        }
        it += kt1 - kt0;
        __syncthreads();        // the staging area of for_each is the next segment's tile buffer
    }
}

// gridDim.y blocks per output tile: sum the parked partial tiles of the blocks that touched it (ascending block index) and run the epilogue.
template <int BM, int BN, class Epi>
__global__ __launch_bounds__(256) void streamk_fixup_kernel(const float* __restrict__ slab, Epi epi, StreamKPlan plan, int G) {
    const int tile = blockIdx.x;
    const int64_t i0 = (int64_t)tile * plan.nkt, i1 = i0 + plan.nkt - 1;
    const int b_first = (int)(((i0 + 1) * G + plan.total - 1) / plan.total) - 1;
    const int b_last = (int)(((i1 + 1) * G + plan.total - 1) / plan.total) - 1;
    if (b_first == b_last) return;       // one block covered the whole tile and already ran the epilogue
    const int m0 = (tile / plan.tiles_n) * BM, n0 = (tile % plan.tiles_n) * BN;
    for (int q = blockIdx.y * 256 + threadIdx.x; q < BM * BN / 4; q += 256 * gridDim.y) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = b_first; b <= b_last; ++b) {
            const int seg = tile - (int)((plan.total * b / G) / plan.nkt);
            const float4 v = reinterpret_cast<const float4*>(slab + ((int64_t)b * plan.max_seg + seg) * (BM * BN))[q];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        epi(m0 + (q * 4) / BN, n0 + (q * 4) % BN, acc);
    }
}

}  // namespace snag
