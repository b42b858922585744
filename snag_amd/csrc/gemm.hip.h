// fp32 GEMM core on the gfx950 matrix pipe.
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )          fp32 in, fp32 accumulate, fp32 out
//
// Each operand is either "K-contiguous" (element (r,k) at ptr[r*ld + k]: activations [rows,K], nn.Linear weights W[out,in]) or
// "row-contiguous" (element (r,k) at ptr[k*ld + r]: the transposed uses in the projections and the backward GEMMs).
//
// Two main loops live in GemmCore::run_presplit, chosen at compile time (DESIGN.md par. 4):
//   * row-private (kRowPrivate: waves stacked along M; long K): f32 values are multiplied as six bf16 MFMA products of exactly split
//     operands.  A wave's 32 rows of A go from global memory to its registers
//     and are split there; B is split once per workgroup into an LDS image in MFMA operand order.  Two forms: 128 x 128, four
//     waves; 256 x 128, eight waves in two groups running half an iteration apart (kPingPong: outputs one column tile wide).
//   * 64 x 128 tile (2 x 2 waves; short K, small outputs; the 128 x 128 Gram tile likewise): both operands are split once per workgroup
//     into LDS images.
// Within a k-tile the sum runs over a fixed permutation of k, the same for both operands.  256 threads = 4 waves (512 = 8 in the ping-pong form).
#pragma once
#include <type_traits>
#include <utility>
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

constexpr int kBK = 16;          // k-tile depth
constexpr int kCPR = kBK / 4;               // 16-byte chunks per K-contiguous tile row

#ifndef SNAG_GEMM_SK_WAVES
#define SNAG_GEMM_SK_WAVES 2             // A/B: waves per SIMD the stream-K kernel is compiled for (4: at most 128 registers)
#endif
// fp32 products on the bf16 matrix pipe: every operand value is split EXACTLY into three bf16 pieces a = h + m + l (round-to-nearest each
// time, the two remainders are exact in f32), and a product a*b becomes the six bf16 MFMA products of weight >= 2^-16: h*h, h*m, m*h, m*m,
// h*l, l*h (v_mfma_f32_32x32x16_bf16, f32 accumulation; the three dropped products m*l, l*m, l*l are below 2^-23 of a*b -- the size of f32's
// own product rounding).  The bf16 matrix rate is 16x the f32 one, so six products cost 3/8 of the f32 MFMA time.  (The f32-MFMA loop, the
// nine-product form, 32-deep k-tiles and the un-swizzled / software-pipelined / LDS-DMA f32 tile loops of rounds 1-3 are gone: no shape of the
// build reached them; HISTORY.md has their measurements.)
constexpr int kBf16x = 6;

struct GemmOperands {
    const float* A;
    const float* B;
    int64_t lda, ldb;
    int M, N, K;
    int a_vec, b_vec;    // 16-byte loads legal for this operand (base and ld aligned)
    // the VEC kernel variant additionally needs: row-contiguous operands with a row count that is a multiple of 4
    int k_per_split;     // K range handled by one blockIdx.z (multiple of kBK); == K rounded up when no split
    int dbg;             // tuning only: nonzero = stop fetching after the first two k-tiles (compute-only ceiling)
    int xcd;             // bit 0: XCD-aware workgroup -> tile mapping (xcd_swizzle); bits 4..: row tiles per group of the grouped order (0 = row-major)
};

// Workgroups are dealt round-robin over the 8 XCDs (observed, never relied on for correctness): ids b and b + 8 share an L2.
// The bijective remap below gives each of the 8 groups a CONTIGUOUS range of tile indices, so tiles that share an operand panel
// -- and, for row pitches that are not a multiple of 128 B, the cache lines straddling a tile boundary -- stay in one L2.
__device__ __forceinline__ int xcd_swizzle(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = orig & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (orig >> 3);
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

// ---- pre-split staging (split-bf16 builds, both operands row-contiguous) -------------------------------------------------------
// The in-fragment split above costs ~5.5 VALU instructions per operand value PER WAVE that consumes it; with 2 x 2 waves every value
// is split twice, and at 11 VALU instructions per MFMA the loop is VALU-bound (compute-only ablation: 128 us for the visual
// projection against 38 us of bf16 MFMA time).  Here every value is split ONCE per workgroup on its way from global memory to the
// LDS, which then holds the three bf16 planes in MFMA operand order:
//     image[plane p][k-half q][row r] = 16 bytes = the 8 bf16 pieces p of row r for k = 8q .. 8q+7 of the 16-deep k-tile
// so a lane's 32x32x16 operand is ONE conflict-free ds_read_b128 per plane (16 consecutive rows = 256 contiguous bytes).
// Thread t owns row t % BR and the KPT = 16 BR / 256 consecutive k starting at KPT (t / BR): KPT dword loads per k-tile, each
// wave-instruction covering 64 consecutive rows of one k (256 contiguous bytes; the k index is wave-uniform, so its bound check is too
// and ragged K needs no second loader), two register sets so that every load has two k-tiles of matrix work to land.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
    return r;
}
// two values -> their packed h, m, l pieces (11 VALU instructions)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// One weight image of RoWeightImages: SplitStager<128, K-contiguous>'s thread mapping and split_pair -- the bits the GEMM kernels put into
// their LDS images -- written to memory instead.
__device__ __forceinline__ void ro_weight_image_body(const RoWeightImages& w, int image) {
    const int s = image >> 3, kt = image & 7, tid = threadIdx.x;
    const int row = tid >> 2, c = tid & 3;
    unsigned char* im = w.wimg + (size_t)image * (6 * 128 * 16) + ((c & 1) * 128 + row) * 16 + (c >> 1) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float f[4];
        if (s < 8) {
            const float4 v = *reinterpret_cast<const float4*>(w.W[s] + (int64_t)(row + 64 * i) * w.ldw + kt * 16 + c * 4);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {        // the transposed use: the k of the image runs over the weight's rows (SplitStager<128, row-contiguous> keeps the same chunks)
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = w.Wt[(int64_t)(kt * 16 + 4 * c + j) * w.ldw + row + 64 * i];
        }
        uint32_t h0, m0, l0, h1, m1, l1;
        split_pair(f[0], f[1], h0, m0, l0);
        split_pair(f[2], f[3], h1, m1, l1);
        *reinterpret_cast<uint2*>(im + i * 64 * 16) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(im + i * 64 * 16 + 2 * 128 * 16) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(im + i * 64 * 16 + 4 * 128 * 16) = make_uint2(l0, l1);
    }
}

// VK: a K-contiguous operand is fetched by 16-byte loads as a COMPILE-TIME choice; otherwise the choice is the run-time `vec`.  With the
// run-time branch around the loads the compiler cannot count them and the waits of the ping-pong loop degrade to one vmcnt(0) every
// fourth iteration (all register sets landed) instead of vmcnt(9) in each: use_vec() in gemm.hip picks VK whenever it is legal.
template <int BR, bool KC, int NT = 256, bool VK = true>
struct SplitStager {
    static constexpr int KPT = 16 * BR / NT;             // values per thread and k-tile (4: BR = 64, or 128 rows over 512 threads; 8: BR = 128)
    static_assert(KPT == 4 || KPT == 8, "pre-split staging: 64 or 128 rows per operand tile");
    static constexpr int IMAGE_BYTES = 6 * BR * 16;
    // The unit of work is a CHUNK: 4 consecutive k (4c .. 4c+3, c = 0..3) of one row = 16 bytes of f32 = 8 bytes per bf16 plane.
    // Chunk c lives in k-half c & 1 of the image, in bytes 8 (c >> 1) .. of the row's 16-byte slot: a lane's MFMA operand (one
    // ds_read_b128) is chunks {h, h + 2} of its row, for both operands alike -- a fixed permutation of k inside the k-tile.
    //   row-contiguous, BR = 64 : thread = (row t % 64, chunk t / 64)                    4 dword loads (k = 4c + j)
    //   row-contiguous, BR = 128: thread = (row t % 128, chunks h and h + 2, h = t / 128) 8 dword loads
    //   K-contiguous,   BR = 64 : thread = (row t / 4, chunk t % 4)                      1 dwordx4 load
    //   K-contiguous,   BR = 128: thread = (rows t / 4 and t / 4 + 64, chunk t % 4)      2 dwordx4 loads
    // (row-contiguous: a wave-instruction covers 64 consecutive rows of one k, 256 contiguous bytes; K-contiguous: 4 adjacent
    // lanes cover the 64 contiguous bytes a row has in this k-tile.)
    float r[4 < 4 ? 4 : 4][KPT];                 // register sets: two in flight (tiles t+1, t+2) in the four-wave loops, four in the ping-pong loop
    const char* ubase;               // wave-uniform.  row-contiguous: element (k = kbeg, row 0); K-contiguous: element (row0, k = kbeg)
    int64_t ldb;                     // pitch in bytes
    uint32_t voff, voff2;            // the lane's part of the address, bytes (rows past the edge clamped: their products are never stored)
    int c;                           // first chunk of this thread (wave-uniform when row-contiguous)
    int wr_off;                      // byte offset of this thread's (first) slot in plane 0 of the image
    int vec;                         // K-contiguous: 16-byte loads legal
    int ablate;                      // tuning only (SNAG_GEMM_DBG): k-tiles >= 2 read through an empty descriptor (no memory traffic, zeros)

    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int vec_ok, int row0, int nrows, int kbeg, int tid) {
        ldb = ld * 4;
        vec = vec_ok;
        int row;
        if constexpr (KC) {
            row = tid >> 2, c = tid & 3;
            ubase = reinterpret_cast<const char*>(P) + (int64_t)row0 * ldb + (int64_t)kbeg * 4;
            voff = (uint32_t)(min(row0 + row, nrows - 1) - row0) * (uint32_t)ldb + (uint32_t)c * 16u;
            voff2 = (uint32_t)(min(row0 + row + 64, nrows - 1) - row0) * (uint32_t)ldb + (uint32_t)c * 16u;
            wr_off = ((c & 1) * BR + row) * 16 + (c >> 1) * 8;
        } else {
            row = tid % BR, c = __builtin_amdgcn_readfirstlane(tid / BR);
            ubase = reinterpret_cast<const char*>(P) + (int64_t)kbeg * ldb;
            voff = voff2 = (uint32_t)min(row0 + row, nrows - 1) * 4u;
            wr_off = KPT == 8 ? (c * BR + row) * 16 : ((c & 1) * BR + row) * 16 + (c >> 1) * 8;
        }
    }
    // k (inside the k-tile) of value j of this thread
    __device__ __forceinline__ int k_of(int j) const { return KC ? 4 * c + (j & 3) : (KPT == 8 ? 4 * c + (j & 3) + 8 * (j >> 2) : 4 * c + j); }
    // address of value j of k-tile kt
    __device__ __forceinline__ const char* addr_of(int kt, int j) const {
        if constexpr (KC) return ubase + (j < 4 ? voff : voff2) + (int64_t)kt * 64 + (j & 3) * 4;
        else return ubase + (int64_t)(kt * 16 + k_of(j)) * ldb + voff;
    }
    // FULL k-tile kt (relative to kbeg) -> register set s, by buffer loads: scalar descriptor + scalar offset + the lane's 32-bit
    // offset, no per-lane address arithmetic.  A tile at or past kleft / 16 reads zeros through an empty descriptor (the hardware's
    // range check covers the vector offset only, not the scalar one: it cannot cut a ragged tile; that is load_tail()'s).
    __device__ __forceinline__ void load(int s, int kt, int kleft /* kend - kbeg */) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int range = __builtin_amdgcn_readfirstlane(kt * 16 + 16 <= kleft && !(ablate && kt >= 2) ? (int)0xffffffffu : 0);
        if constexpr (KC) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase), 0, range, 0x00020000);
            const uint32_t so = (uint32_t)kt * 64u;
            if (VK || vec) {
#pragma unroll
                for (int i = 0; i < KPT / 4; ++i) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i ? voff2 : voff, so, 0);
                    r[s][4 * i + 0] = __uint_as_float(v[0]);
                    r[s][4 * i + 1] = __uint_as_float(v[1]);
                    r[s][4 * i + 2] = __uint_as_float(v[2]);
                    r[s][4 * i + 3] = __uint_as_float(v[3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < KPT; ++j) r[s][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (j < 4 ? voff : voff2) + 4 * (j & 3), so, 0));
            }
        } else {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase + (int64_t)kt * 16 * ldb), 0, range, 0x00020000);
            const uint32_t l = (uint32_t)ldb;
#pragma unroll
            for (int j = 0; j < KPT; ++j) r[s][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (uint32_t)k_of(j) * l, 0));
        }
#endif
    }
    // the ragged last k-tile (kt = kleft / 16): element by element, k >= kend reads as zero
    __device__ __forceinline__ void load_tail(int s, int kt, int kleft) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) r[s][j] = kt * 16 + k_of(j) < kleft ? *reinterpret_cast<const float*>(addr_of(kt, j)) : 0.f;
    }
    // SNAG_GEMM_STREAM_ONLY: consume register set s with one VALU instruction per value (keeps the loads and their waits alive)
    __device__ __forceinline__ void sink(int s, uint32_t& acc) const {
#pragma unroll
        for (int j = 0; j < KPT; ++j) acc ^= __float_as_uint(r[s][j]);
    }
    __device__ __forceinline__ void split_store(int s, unsigned char* image) const {
#if defined(__HIP_DEVICE_COMPILE__)
        using lds_u4 = __attribute__((address_space(3))) uint4;
        using lds_u2 = __attribute__((address_space(3))) uint2;
        uint32_t h[KPT / 2], m[KPT / 2], l[KPT / 2];
#pragma unroll
        for (int j = 0; j < KPT / 2; ++j) split_pair(r[s][2 * j], r[s][2 * j + 1], h[j], m[j], l[j]);
        auto* im = (__attribute__((address_space(3))) unsigned char*)image + wr_off;
        if constexpr (KPT == 8 && !KC) {          // chunks c and c + 2 of one row: the whole 16-byte slot
            *(lds_u4*)(im) = make_uint4(h[0], h[1], h[2], h[3]);
            *(lds_u4*)(im + 2 * BR * 16) = make_uint4(m[0], m[1], m[2], m[3]);
            *(lds_u4*)(im + 4 * BR * 16) = make_uint4(l[0], l[1], l[2], l[3]);
        } else {
#pragma unroll
            for (int i = 0; i < KPT / 4; ++i) {   // K-contiguous, BR = 128: the same chunk of rows row and row + 64
                *(lds_u2*)(im + i * 64 * 16) = make_uint2(h[2 * i], h[2 * i + 1]);
                *(lds_u2*)(im + i * 64 * 16 + 2 * BR * 16) = make_uint2(m[2 * i], m[2 * i + 1]);
                *(lds_u2*)(im + i * 64 * 16 + 4 * BR * 16) = make_uint2(l[2 * i], l[2 * i + 1]);
            }
        }
#endif
    }
};

// An operand whose fragment belongs to ONE wave (waves stacked along M: the wave's 32 rows are nobody else's) skips the LDS: the lane
// loads its own 8 values of the k-tile -- chunks {h, h + 2} of its row, the same permutation of k as the images -- straight into
// registers and splits them there (no redundancy: no other wave needs them).
//   row-contiguous: 8 dword loads; a wave-instruction covers 32 rows of two k = two contiguous 128-byte runs
//   K-contiguous:   2 dwordx4 loads (bytes 16 h and 32 + 16 h of the row's 64 bytes in this k-tile)
template <bool KC, bool VK = true>
struct FragLoader {
    float r[4 < 4 ? 4 : 4][8];
    const char* ubase;
    int64_t ldb;
    uint32_t voff;
    int lh, vec, ablate;

    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int vec_ok, int row0 /* of the wave */, int nrows, int kbeg, int lane) {
        ldb = ld * 4;
        vec = vec_ok;
        lh = lane >> 5;
        const int li = lane & 31;
        if constexpr (KC) {
            const int base = min(row0, nrows - 1);          // a wave wholly past the edge of the last tile reads the last row
            ubase = reinterpret_cast<const char*>(P) + (int64_t)base * ldb + (int64_t)kbeg * 4;
            voff = (uint32_t)(min(row0 + li, nrows - 1) - base) * (uint32_t)ldb + (uint32_t)lh * 16u;
        } else {
            ubase = reinterpret_cast<const char*>(P) + (int64_t)kbeg * ldb;
            voff = (uint32_t)min(row0 + li, nrows - 1) * 4u + (uint32_t)(4 * lh) * (uint32_t)ldb;
        }
    }
    __device__ __forceinline__ int k_of(int j) const { return 4 * lh + (j & 3) + 8 * (j >> 2); }
    __device__ __forceinline__ void load(int s, int kt, int kleft) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int range = __builtin_amdgcn_readfirstlane(kt * 16 + 16 <= kleft && !(ablate && kt >= 2) ? (int)0xffffffffu : 0);
        if constexpr (KC) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase), 0, range, 0x00020000);
            const uint32_t so = (uint32_t)kt * 64u;
            if (VK || vec) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, so + 32u * i, 0);
                    r[s][4 * i + 0] = __uint_as_float(v[0]);
                    r[s][4 * i + 1] = __uint_as_float(v[1]);
                    r[s][4 * i + 2] = __uint_as_float(v[2]);
                    r[s][4 * i + 3] = __uint_as_float(v[3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[s][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so + 4u * (j & 3) + 32u * (j >> 2), 0));
            }
        } else {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase + (int64_t)kt * 16 * ldb), 0, range, 0x00020000);
            const uint32_t l = (uint32_t)ldb;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[s][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (uint32_t)((j & 3) + 8 * (j >> 2)) * l, 0));
        }
#endif
    }
    __device__ __forceinline__ void load_tail(int s, int kt, int kleft) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kt * 16 + k_of(j);
            const char* g = KC ? ubase + voff + (int64_t)kt * 64 + 4 * (j & 3) + 32 * (j >> 2) : ubase + voff + (int64_t)(kt * 16 + (j & 3) + 8 * (j >> 2)) * ldb;
            r[s][j] = k < kleft ? *reinterpret_cast<const float*>(g) : 0.f;
        }
    }
    __device__ __forceinline__ void sink(int s, uint32_t& acc) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= __float_as_uint(r[s][j]);
    }
    __device__ __forceinline__ void split(int s, Split3& f) const {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(r[s][2 * j], r[s][2 * j + 1], h[j], m[j], l[j]);
        using u4 = __attribute__((ext_vector_type(4))) uint32_t;
        f.h = __builtin_bit_cast(bf16x8, (u4){h[0], h[1], h[2], h[3]});
        f.m = __builtin_bit_cast(bf16x8, (u4){m[0], m[1], m[2], m[3]});
        f.l = __builtin_bit_cast(bf16x8, (u4){l[0], l[1], l[2], l[3]});
#endif
    }
};

// a lane's three operand pieces of one 32-row fragment: rows row0 .. row0+31 of an image with BR rows
template <int BR>
__device__ __forceinline__ void load_split_fragment(const unsigned char* image, int row, int lh, Split3& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_b8 = __attribute__((address_space(3))) bf16x8;
    const auto* im = (const __attribute__((address_space(3))) unsigned char*)image + (lh * BR + row) * 16;
    f.h = *(const lds_b8*)(im);
    f.m = *(const lds_b8*)(im + 2 * BR * 16);
    f.l = *(const lds_b8*)(im + 4 * BR * 16);
#endif
}

// one plane (0 = h, 1 = m, 2 = l) of the same fragment
template <int BR>
__device__ __forceinline__ bf16x8 load_split_plane(const unsigned char* image, int row, int lh, int plane) {
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_b8 = __attribute__((address_space(3))) bf16x8;
    return *(const lds_b8*)((const __attribute__((address_space(3))) unsigned char*)image + (lh * BR + row) * 16 + plane * 2 * BR * 16);
#else
    return bf16x8{};
#endif
}

// Main loop.  acc[mi][ni] is the 32x32 tile (mi, ni) of this wave's WM x WN patch:
//   row = wm0 + 32*mi + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5),  col = wn0 + 32*ni + (lane & 31).
// (With split-bf16 products C(a, b) and C(b, a) differ in their last bits: h*m <-> m*h, h*l <-> l*h under an exchange of the operands,
// and f32 addition is not associative.  snag_gram_f32 therefore mirrors the upper triangle of its diagonal tiles instead of trusting
// both halves.)
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC>
struct GemmCore {
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int CW = (WN % 64 == 0) ? 64 : 32;   // epilogue staging chunk: 32 rows x CW columns per wave at a time
    static constexpr int NW = (BM / WM) * (BN / WN), NT = 64 * NW;       // waves / threads per workgroup
    static constexpr int STAGE_FLOATS = NW * 32 * CW;
    // pre-split staging loop: split-bf16 builds, 64 / 128-row operand tiles (the 128 x 160 tile keeps the in-fragment split)
    static constexpr bool kPresplit = kBf16x != 0 && (BM == 64 || BM == 128 || (BM == 256 && WM == 32 && WN == BN)) && (BN == 64 || BN == 128) && kBK == 16 &&
                                      (1 != 0 || (!A_KC && !B_KC));
    // waves stacked along M (each wave's 32 rows are its own): the A operand never goes through the LDS (FragLoader)
    static constexpr bool kRowPrivate = kPresplit && WN == BN && WM == 32;
    static constexpr int PS_STAGE_FLOATS = 6 * ((kRowPrivate ? 0 : BM) + BN) * 16 / 4;          // one stage: both images (B's only when row-private)
    // (the eight-wave ping-pong loop keeps THREE B images: a group reads the fragments of tile kt inside its matrix half, while the other
    // group is already writing tile kt+2 -- SNAG_GEMM_PP_LATE_READS)
    static constexpr int PS_IMAGES = (kRowPrivate && (BM / WM) * (BN / WN) == 8 && 1) ? 3 : 2;
    static constexpr int PS_FLOATS = kPresplit ? PS_IMAGES * PS_STAGE_FLOATS : 0;
    static constexpr int SMEM_FLOATS = PS_FLOATS > STAGE_FLOATS ? PS_FLOATS : STAGE_FLOATS;
    static_assert(NW == 4 || NW == 8, "four or eight waves per block");
    // eight waves stacked along M: two groups of four run the k-loop half an iteration apart (ping-pong, run_presplit)
    static constexpr bool kPingPong = kRowPrivate && NW == 8;
    static_assert(NW == 4 || kPingPong, "eight waves: the row-private ping-pong loop only");

    f32x16 acc[TM][TN];

    // the six products of one k-tile on pre-split fragments (smallest first; one pass over the accumulators per product keeps dependent MFMAs apart)
    __device__ __forceinline__ void mfma_split(const Split3 (&a)[TM], const Split3 (&b)[TN]) {
#if defined(__HIP_DEVICE_COMPILE__)
#define SNAG_MFMA_TERM(ACC, X, Y)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                     \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].X, b[j].Y, ACC[i][j], 0, 0, 0);
        SNAG_MFMA_TERM(acc, l, h)
        SNAG_MFMA_TERM(acc, h, l)
        SNAG_MFMA_TERM(acc, m, m)
        SNAG_MFMA_TERM(acc, m, h)
        SNAG_MFMA_TERM(acc, h, m)
        SNAG_MFMA_TERM(acc, h, h)
#undef SNAG_MFMA_TERM
#endif
    }

    __device__ __forceinline__ void run_presplit(const GemmOperands& p, float* __restrict__ smem, int m0, int n0, int kbeg, int kend) {
        const int tid = threadIdx.x, lane = tid & 63, wave = wave_id_uniform();
        const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
        const int li = lane & 31, lh = lane >> 5;
        unsigned char* const sm = reinterpret_cast<unsigned char*>(smem);
        constexpr int STAGE = PS_STAGE_FLOATS * 4, A_IMG = kRowPrivate ? 0 : 6 * BM * 16;
        [[maybe_unused]] constexpr int kPPImages = PS_IMAGES;          // B images of the ping-pong loop (3 with late reads)
        const int kleft = kend - kbeg;
        const int nk = kleft / 16;            // full k-tiles; a ragged last one is the tail's
        if constexpr (kRowPrivate) {
            // A: registers only.  Iteration kt multiplies tile kt (A pieces split during iteration kt-1, B fragments out of buffer
            // kt & 1), splits A of tile kt+1 and stages B of tile kt+1 (register sets (kt+1) & 1), fetches tile kt+3 into those sets.
            static_assert(TM == 1, "row-private: one 32-row fragment per wave");
            FragLoader<A_KC, VEC> fa;
            SplitStager<BN, B_KC, NT, VEC> sb;
            fa.init(p.A, p.lda, p.a_vec, m0 + wm0, p.M, kbeg, lane);
            sb.init(p.B, p.ldb, p.b_vec, n0, p.N, kbeg, tid);
            fa.ablate = p.dbg & 1;
            sb.ablate = p.dbg & 2;
            Split3 a_cur[1], a_nxt[1];
            [[maybe_unused]] auto iteration = [&](int kt, auto S) {
                constexpr int set = decltype(S)::value;
                const int cur = kt & 1;
                Split3 b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) load_split_fragment<BN>(sm + cur * STAGE, wn0 + 32 * j + li, lh, b[j]);
                fa.split(set, a_nxt[0]);
                fa.load(set, kt + 3, kleft);
                sb.split_store(set, sm + (cur ^ 1) * STAGE);
                sb.load(set, kt + 3, kleft);
                mfma_split(a_cur, b);
                a_cur[0] = a_nxt[0];
                __syncthreads();
            };
            // Eight waves: PING-PONG.  The two groups of four waves (one wave of each group per SIMD) run the same loop half an
            // iteration apart, and an iteration is cut by a barrier into its memory half (fragment reads, the A split, the group's
            // share of the next B image, the loads) and its matrix half (24 MFMAs): while one group multiplies, the other one
            // feeds -- the matrix pipe of a SIMD always has exactly one wave on it.  Group 1 starts one barrier late and group 0
            // leaves one barrier late, so both pass the same number of barriers.  Four register sets (tiles kt+1 .. kt+4 in flight: twice
            // the bytes in flight per CU of the four-wave loops).  B image of tile kt+1: group 0's half is written
            // in interval 2 kt, group 1's in interval 2 kt + 1, both before anyone reads it (group 0 in interval 2 kt + 2); the
            // buffer it replaces was last read in intervals 2 kt - 2 / 2 kt - 1.
            // In-kernel stamps (tools/stamp_gemm.py) of the first form of this loop -- fragment reads at the END of the memory half,
            // behind the split's VALU block -- showed the memory half (780-1,100 cycles with every load landed) longer than the matrix
            // half (~860): the intervals ran at the pace of the feeding group.  The fragment reads therefore moved INTO the matrix half
            // (SNAG_GEMM_PP_LATE_READS): twelve ds_read_b128 in the shadow of 24 MFMAs cost next to nothing there (2 per MFMA gap are
            // free, MI355X_MICROARCH.md "LDS"), the memory half is the split and the loads alone.  A group then reads tile kt one
            // interval later than before -- while the other group writes tile kt+2 -- hence three B images instead of two.
            constexpr int kPPSets = 4;
            static_assert(kPPSets == 4 || kPPSets == 6 || kPPSets == 8, "ping-pong register sets");
            [[maybe_unused]] int rd_off = 0, wr_off = STAGE;               // byte offsets of the image being read (tile kt) / written (tile kt+1)
            [[maybe_unused]] uint32_t stream_sink = 0;
            [[maybe_unused]] auto pp_iteration = [&](int kt, auto S) {
                constexpr int set = decltype(S)::value;
                [[maybe_unused]] Split3 b[TN];
                constexpr bool do_asplit = true, do_bsplit = true, do_load = true;
                if (do_asplit) fa.split(set, a_nxt[0]);
                if (do_load) fa.load(set, kt + 1 + kPPSets, kleft);
                if (do_bsplit) sb.split_store(set, sm + wr_off);
                if (do_load) sb.load(set, kt + 1 + kPPSets, kleft);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
                // (tile kt's image was completed by the barrier that ended the PREVIOUS interval: both groups wrote their halves one and two
                // intervals ago; nobody overwrites it before tile kt+3 is staged)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j].h = load_split_plane<BN>(sm + rd_off, wn0 + 32 * j + li, lh, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    b[j].l = load_split_plane<BN>(sm + rd_off, wn0 + 32 * j + li, lh, 2);
                    b[j].m = load_split_plane<BN>(sm + rd_off, wn0 + 32 * j + li, lh, 1);
                }
                mfma_split(a_cur, b);
                a_cur[0] = a_nxt[0];
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#endif
                rd_off = wr_off;
                wr_off = wr_off + STAGE == kPPImages * STAGE ? 0 : wr_off + STAGE;
            };
            if (nk > 0) {
                fa.load(0, 0, kleft);
                sb.load(0, 0, kleft);
                fa.load(1, 1, kleft);
                sb.load(1, 1, kleft);
                if constexpr (kPingPong) {          // SETS register sets: tiles kt+1 .. kt+SETS in flight (tile t in set t % SETS)
#pragma unroll
                    for (int q = 2; q < kPPSets; ++q) {
                        fa.load(q, q, kleft);
                        sb.load(q, q, kleft);
                    }
                }
                fa.split(0, a_cur[0]);
                fa.load(0, kPingPong ? kPPSets : 2, kleft);
                sb.split_store(0, sm);
                sb.load(0, kPingPong ? kPPSets : 2, kleft);
                __syncthreads();
                if constexpr (kPingPong) {
#if defined(__HIP_DEVICE_COMPILE__)
                    const bool late = wave >= 4;
                    if (late) __builtin_amdgcn_s_barrier();
                    for (int kt = 0; kt < nk; kt += kPPSets) {       // iteration kt consumes tile kt + 1: set (kt + 1) % SETS
                        pp_iteration(kt, std::integral_constant<int, 1>{});
                        pp_iteration(kt + 1, std::integral_constant<int, 2>{});
                        pp_iteration(kt + 2, std::integral_constant<int, 3>{});
                        pp_iteration(kt + 3, std::integral_constant<int, 4 % kPPSets>{});
                        if constexpr (kPPSets >= 6) {
                            pp_iteration(kt + 4, std::integral_constant<int, 5 % kPPSets>{});
                            pp_iteration(kt + 5, std::integral_constant<int, 6 % kPPSets>{});
                        }
                        if constexpr (kPPSets >= 8) {
                            pp_iteration(kt + 6, std::integral_constant<int, 7 % kPPSets>{});
                            pp_iteration(kt + 7, std::integral_constant<int, 0>{});
                        }
                    }
                    if (!late) __builtin_amdgcn_s_barrier();
                    __syncthreads();
#endif
                } else {
                    for (int kt = 0; kt < nk; kt += 2) {
                        iteration(kt, std::integral_constant<int, 1>{});
                        iteration(kt + 1, std::integral_constant<int, 0>{});
                    }
                }
            }
            if (kleft % 16 != 0) {
                fa.load_tail(0, nk, kleft);
                sb.load_tail(0, nk, kleft);
                fa.split(0, a_cur[0]);
                sb.split_store(0, sm);
                __syncthreads();
                Split3 b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) load_split_fragment<BN>(sm, wn0 + 32 * j + li, lh, b[j]);
                mfma_split(a_cur, b);
                __syncthreads();
            }
            return;
        } else {
        SplitStager<BM, A_KC, 256, VEC> sa;
        SplitStager<BN, B_KC, 256, VEC> sb;
        sa.init(p.A, p.lda, p.a_vec, m0, p.M, kbeg, tid);
        sb.init(p.B, p.ldb, p.b_vec, n0, p.N, kbeg, tid);
        sa.ablate = p.dbg & 1;            // SNAG_GEMM_DBG: 1 = no A fetch, 2 = no B fetch, 3 = neither (results are wrong)
        sb.ablate = p.dbg & 2;
        auto fetch = [&](int s, int kt) {
            sa.load(s, kt, kleft);
            sb.load(s, kt, kleft);
        };
        auto stage = [&](int s, int buf) {
            sa.split_store(s, sm + buf * STAGE);
            sb.split_store(s, sm + buf * STAGE + A_IMG);
        };
        auto multiply = [&](int buf) {
            Split3 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) load_split_fragment<BM>(sm + buf * STAGE, wm0 + 32 * i + li, lh, a[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) load_split_fragment<BN>(sm + buf * STAGE + A_IMG, wn0 + 32 * j + li, lh, b[j]);
            mfma_split(a, b);
        };
        // one k-tile: fragments of tile kt out of buffer kt & 1; tile kt+1 (register set S) split into the other buffer; tile kt+3 fetched
        // into set S; the matrix work of tile kt; barrier.  Set S ^ 1 holds tile kt+2, in flight across the whole iteration.
        // Staging and fetching are unconditional (a tile past the end reads as zeros and lands in the buffer nobody reads again): a
        // fetch the compiler cannot count on forces its vmcnt waits down to 0, i.e. to wait for the NEWER set as well.
        auto iteration = [&](int kt, auto S) {
            constexpr int set = decltype(S)::value;
            const int cur = kt & 1;
            Split3 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) load_split_fragment<BM>(sm + cur * STAGE, wm0 + 32 * i + li, lh, a[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) load_split_fragment<BN>(sm + cur * STAGE + A_IMG, wn0 + 32 * j + li, lh, b[j]);
            stage(set, cur ^ 1);
            fetch(set, kt + 3);
            mfma_split(a, b);
            __syncthreads();
        };
        if (nk > 0) {
            fetch(0, 0);
            fetch(1, 1);
            stage(0, 0);
            fetch(0, 2);
            __syncthreads();
            // tile kt+1 sits in set (kt+1) & 1.  Always whole pairs: with an odd count the second iteration of the last pair multiplies
            // a tile of zeros (at most one k-tile in 59+ per stream-K segment), which keeps the loop a single straight-line body.
            for (int kt = 0; kt < nk; kt += 2) {
                iteration(kt, std::integral_constant<int, 1>{});
                iteration(kt + 1, std::integral_constant<int, 0>{});
            }
        }
        if (kleft % 16 != 0) {
            sa.load_tail(0, nk, kleft);
            sb.load_tail(0, nk, kleft);
            stage(0, 0);
            __syncthreads();
            multiply(0);
            __syncthreads();
        }
        }
    }

    __device__ __forceinline__ void run(const GemmOperands& p, float* __restrict__ smem, int m0, int n0, int kbeg, int kend) {
        static_assert(kPresplit, "every tile of the build multiplies pre-split bf16 images");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        if (kend > kbeg) run_presplit(p, smem, m0, n0, kbeg, kend);
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

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__device__ __forceinline__ void gemm_tile_at(const GemmOperands& p, const Epi& epi, float* __restrict__ slab, int z, int m0, int n0, float* smem);

// One block = one BM x BN output tile (x one K split): main loop, then either the epilogue functor
// epi(row, col, float4) or, for split-K, a raw store into this split's slab (split index z).
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__device__ __forceinline__ void gemm_tile_body(const GemmOperands& p, const Epi& epi, float* __restrict__ slab, int z, float* smem) {
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd) {
        const int id = xcd_swizzle(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        const int gm = p.xcd >> 4;
        if (gm > 0) {
            // grouped order inside the XCD's range: gm row tiles deep, rows fastest -- the ~32 workgroups an XCD runs side by side
            // cover gm x (32 / gm) tiles instead of 1 x 32, and (moving through K in step) share every A and B k-slab through the L2
            const int per = gm * (int)gridDim.x, g = id / per, first = g * gm;
            const int gsz = min((int)gridDim.y - first, gm), in = id - g * per;
            by = first + in % gsz;
            bx = in / gsz;
        } else {
            bx = id % gridDim.x;
            by = id / gridDim.x;
        }
    }
    gemm_tile_at<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p, epi, slab, z, by * BM, bx * BN, smem);
}

// the tile at (m0, n0) of K split z
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__device__ __forceinline__ void gemm_tile_at(const GemmOperands& p, const Epi& epi, float* __restrict__ slab, int z, int m0, int n0, float* smem) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
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

// Two problems of DIFFERENT shape (same layouts, no K split) in one launch: a linear tile index, problem 0's tiles first.  The
// query projection of the read-out token (235 tiles) fills the drain of the key / value projection (1,408 tiles) instead of paying
// its own launch, first-load latency and drain.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(256) void gemm_tile_group2_kernel(GemmOperands p0, Epi e0, GemmOperands p1, Epi e1, int tiles0, int tiles_n0, int tiles_n1) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    int b = blockIdx.x;
    const bool first = b < tiles0;
    if (!first) b -= tiles0;
    const GemmOperands& p = first ? p0 : p1;
    const int tn = first ? tiles_n0 : tiles_n1;
    const int m0 = (b / tn) * BM, n0 = (b % tn) * BN;
    Core core;
    core.run(p, smem, m0, n0, 0, p.K);
    if (first) core.for_each(smem, m0, n0, e0);
    else core.for_each(smem, m0, n0, e1);
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_tile_kernel(GemmOperands p, Epi epi, float* __restrict__ slab) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    gemm_tile_body<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p, epi, slab, blockIdx.z, smem);
}

// TWO problems of the same shape and layout in one launch (blockIdx.z = problem * splits + split): the step's many
// [15000 x 128 x 128]-sized GEMMs are launch / prologue-bound grids of 235 workgroups (< 1 per CU); a pair of them --
// the two modalities' second projections, their backward, the two D x D weight gradients of the read-out token --
// fills the chip once instead of twice.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_tile_pair_kernel(GemmOperands p0, Epi e0, float* __restrict__ slab0, GemmOperands p1, Epi e1,
                                                             float* __restrict__ slab1, int splits) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    const int z = blockIdx.z % splits;
    if (blockIdx.z < splits) gemm_tile_body<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p0, e0, slab0, z, smem);
    else gemm_tile_body<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(p1, e1, slab1, z, smem);
}

// UP TO SIX problems of any shapes (same layouts and loader) in one launch, each with its own K split and slab: a linear block index,
// problem by problem, inside a problem split by split.  The backward of the read-out layer has six independent [15000 x 128 x 128..384]-
// class products behind the attention backward; one by one each is a 235-workgroup grid that pays its own launch, first-load latency
// and drain (16-27 us each for 1-3 us of matrix work).
constexpr int kGroupMax = 8;
static_assert(kGroupMax == SNAG_GEMM_GROUP_MAX, "include/snag_hip.h");
template <class Epi>
struct GemmGroup {
    GemmOperands p[kGroupMax];
    Epi e[kGroupMax];
    float* slab[kGroupMax];          // nullptr: no K split
    int first[kGroupMax + 1];        // block ranges
    int tiles[kGroupMax], tiles_n[kGroupMax];
    int n;
    int xcd;                         // > 0: XCD-aware block -> (problem, split, tile) mapping for the first `xcd` blocks of the launch
    unsigned a_kc_mask;              // gemm_tile_group_mixed_kernel: bit q = problem q's A operand is K-contiguous
};
// one workgroup of a grouped launch (block index `block`)
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__device__ __forceinline__ void gemm_group_block(const GemmGroup<Epi>& g, int block, float* smem) {
    // Linear ids run problem by problem, split by split, tile by tile: neighbours share the k-range of their split -- the same slice of
    // the narrow operand B.  Workgroups are dealt round-robin over the eight XCDs, so without the remap the tiles of one split land on
    // eight different L2s and each fetches its own copy of that slice; with it every XCD owns a contiguous run of ids.
    // (Set by the host for the weight-gradient groups only -- both operands row-contiguous, K = the entity rows, equal k-ranges per
    // block: 40.8 -> 37.0 us.  Launches whose problems differ in work per block lose by it: an XCD that owns the short problem's run idles
    // at the end -- first projections 110.5 -> 117.8 us, input-gradient group 38.0 -> 44.7.)
    const int id = block < g.xcd ? xcd_swizzle(block, g.xcd) : block;
    int q = 0;
    while (q + 1 < g.n && id >= g.first[q + 1]) ++q;
    const int b = id - g.first[q];
    const int z = b / g.tiles[q], t = b % g.tiles[q];
    gemm_tile_at<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(g.p[q], g.e[q], g.slab[q], z, (t / g.tiles_n[q]) * BM, (t % g.tiles_n[q]) * BN, smem);
}
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_tile_group_kernel(GemmGroup<Epi> g) {
    using Core = GemmCore<BM, BN, WM, WN, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    gemm_group_block<BM, BN, WM, WN, A_KC, B_KC, VEC, Epi>(g, (int)blockIdx.x, smem);
}

// The same with the A operand's layout chosen PER PROBLEM (round 5): the read-out layer's backward issues four weight gradients (A = an
// activation gradient read row-contiguously, K = the entity rows) and three input gradients (A K-contiguous, K = 128 .. 384) that all depend on
// the attention backward alone -- two launches of 896 and 705 workgroups, each with its own ramp and tail (37 + 38 us), become one.  Every
// problem runs the very code of its own instantiation: the same bits.
template <int BM, int BN, int WM, int WN, bool B_KC, bool VEC, class Epi>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_tile_group_mixed_kernel(GemmGroup<Epi> g) {
    using CoreK = GemmCore<BM, BN, WM, WN, true, B_KC, VEC>;
    using CoreR = GemmCore<BM, BN, WM, WN, false, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[CoreK::SMEM_FLOATS > CoreR::SMEM_FLOATS ? CoreK::SMEM_FLOATS : CoreR::SMEM_FLOATS];
    const int id = (int)blockIdx.x < g.xcd ? xcd_swizzle((int)blockIdx.x, g.xcd) : (int)blockIdx.x;
    int q = 0;
    while (q + 1 < g.n && id >= g.first[q + 1]) ++q;
    const int b = id - g.first[q];
    const int z = b / g.tiles[q], t = b % g.tiles[q];
    if ((g.a_kc_mask >> q) & 1u) gemm_tile_at<BM, BN, WM, WN, true, B_KC, VEC, Epi>(g.p[q], g.e[q], g.slab[q], z, (t / g.tiles_n[q]) * BM, (t % g.tiles_n[q]) * BN, smem);
    else gemm_tile_at<BM, BN, WM, WN, false, B_KC, VEC, Epi>(g.p[q], g.e[q], g.slab[q], z, (t / g.tiles_n[q]) * BM, (t % g.tiles_n[q]) * BN, smem);
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
__global__ __launch_bounds__(256, SNAG_GEMM_SK_WAVES) void gemm_streamk_kernel(GemmOperands p, Epi epi, float* __restrict__ slab, StreamKPlan plan) {
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
