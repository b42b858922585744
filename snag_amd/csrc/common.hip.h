// Shared device/host helpers for libsnag_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/snag_hip.h"

namespace snag {

constexpr int kWave = 64;

// ---- error reporting (host) -------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

#define SNAG_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) return ::snag::fail((code), __VA_ARGS__); \
    } while (0)

// ---- cross-lane reductions --------------------------------------------------
// DPP row rotations stay inside a 16-lane row and fuse into v_add_f32_dpp: no LDS traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// all-reduce over each aligned group of 16 lanes; every lane ends with the bitwise-same sum
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f32<0x128>(v);  // row_ror:8
    v += dpp_f32<0x124>(v);  // row_ror:4
    v += dpp_f32<0x122>(v);  // row_ror:2
    v += dpp_f32<0x121>(v);  // row_ror:1
    return v;
}

// DPP with explicit row mask: rows whose bit is clear receive `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32_rows(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// all-reduce over the whole 64-lane wave, no LDS traffic: four row rotations give every lane its row's sum, row_bcast:15
// folds row 0 into 1 and row 2 into 3, row_bcast:31 folds rows 0+1 into 2 and 3, lane 63 then holds the total and a
// v_readlane hands it to every lane through an SGPR.
__device__ __forceinline__ float sum64(float v) {
    v = sum16(v);
    v += dpp_f32_rows<0x142, 0xa>(0.f, v);   // row_bcast:15 -> rows 1, 3
    v += dpp_f32_rows<0x143, 0xc>(0.f, v);   // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// the same for 32-bit integers (lane 63's total, handed to every lane)
__device__ __forceinline__ int isum64(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_f32<0x128>(v));
    v = fmaxf(v, dpp_f32<0x124>(v));
    v = fmaxf(v, dpp_f32<0x122>(v));
    v = fmaxf(v, dpp_f32<0x121>(v));
    return v;
}

__device__ __forceinline__ float max64(float v) {
    v = max16(v);
    v = fmaxf(v, dpp_f32_rows<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_f32_rows<0x143, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// One LayerNorm row held by a wave, column c = lane + 64 u in x[u] (0 where c >= D): biased variance, eps inside the sqrt
// (nn.LayerNorm(eps=1e-12); model/SNAG_tools.py:215,221).  ONE statement of the arithmetic for every kernel that normalises a row
// (ln_fwd_kernel, the GEMM launch that normalises its own output rows), so that they agree bit for bit.
template <int NU>
__device__ __forceinline__ void ln_row_forward(const float (&x)[NU], int D, int lane, const float* __restrict__ g, const float* __restrict__ b, float eps,
                                               float (&o)[NU], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) s += x[u];
    mean = sum64(s) / (float)D;
    float vs = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const float dlt = (lane + 64 * u < D) ? x[u] - mean : 0.f;
        vs = fmaf(dlt, dlt, vs);
    }
    rstd = 1.0f / sqrtf(sum64(vs) / (float)D + eps);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = lane + 64 * u;
        const float gg = c < D ? g[c] : 0.f, bb = c < D ? b[c] : 0.f;
        o[u] = (x[u] - mean) * rstd * gg + bb;
    }
}

__device__ __forceinline__ int wave_id_uniform() {
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// F.normalize denominator: max(sqrt(ss), 1e-12)
__device__ __forceinline__ float l2_denominator(float ss) { return fmaxf(sqrtf(ss), 1e-12f); }

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// tuning switches read once from the environment (0 when unset)
static inline int env_int(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

// The two slab reductions of a layer's backward (results nobody reads before the optimiser): y = 0 -> the LayerNorm slab [nb3][3][D]
// into dgamma / dbeta / dbias, y = 1 -> the attention slab [nbq][3D] into the stacked q/k/v bias gradient.  Column block x covers 64 columns.
// Sixteen partial sums per column (slab rows split into sixteen runs), then their sum in order (the arithmetic of the launch this used to be).
struct SlabColsum {
    const float* part3; int nb3; int D;
    float *a, *b, *c;
    const float* partq; int nbq;
    float* outq;
    int accumulate;
};
// 256 threads = 16 runs x 16 columns (column block bx covers 16 columns): every run of a column has its own thread, as in a launch of
// sixteen waves, so the role is as short as that launch was (6.8 us) and hides behind the reduce it rides with.
// (tid: the thread's index inside its 256-thread block -- a 512-thread workgroup runs two blocks side by side, each with its own `red`)
__device__ __forceinline__ void slab_colsum_body(const SlabColsum& s, int bx, int by, float (*red)[16], int tid) {
    const int run = tid >> 4, cl = tid & 15;
    const int c = bx * 16 + cl;
    const int ncol = 3 * s.D;
    const bool ln = by == 0;
    const float* part = ln ? s.part3 : s.partq;
    const int nblocks = ln ? s.nb3 : s.nbq;
    const int64_t stride = ncol;
    const int per = (nblocks + 15) / 16;
    const int b0 = run * per, b1 = min(nblocks, b0 + per);
    float sum = 0.f;
    if (c < ncol) {
        int blk = b0;
        for (; blk + 8 <= b1; blk += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(blk + u) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
        for (; blk < b1; ++blk) sum += part[(int64_t)blk * stride + c];
    }
    red[run][cl] = sum;
    __syncthreads();
    if (run == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][cl];
        float* dst = ln ? (c < s.D ? s.a + c : (c < 2 * s.D ? s.b + (c - s.D) : s.c + (c - 2 * s.D))) : s.outq + c;
        *dst = s.accumulate ? *dst + t : t;
    }
}

// The weights of the one-launch read-out forward (csrc/readout_fused.hip) as split k-tile images in consumption order: image 8 s + kt =
// k-tile kt of product s, 12 KB each.  One workgroup (256 threads) per image.
// Images 64 .. 71 (optional, Wt != nullptr): a ninth weight as the B operand of its TRANSPOSED use, B(n, k) = Wt[k * ldw + n] -- the output
// projection's weight in the backward's d ctx = dz Wo.
struct RoWeightImages {
    const float* W[8];         // 128 weight rows each, K-contiguous, pitch ldw floats
    const float* Wt;
    int ldw;
    int n_images;              // 64 or 72
    unsigned char* wimg;
};

// gemm.hip
int gemm_check(const snag_gemm_desc* d);
int gemm_dispatch(const snag_gemm_desc* d, hipStream_t st);
int gemm_dispatch_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, hipStream_t st);
// Work that depends on nothing the group computes and that nobody reads before the group is done: deferred K-split reduces (+ a layer's slab
// column sums).  Handed to a group of eight-wave 256-row tiles it RIDES in that launch, on the CUs the launch leaves idle (one 512-thread workgroup
// of that kernel fills a CU: a 247-workgroup launch idles 9 of 256) -- otherwise it goes out as the reduce launch it would have been, in front.
struct GroupTail {
    const snag_gemm_desc* const* reduce;
    int n_reduce;
    const SlabColsum* colsum;
};
int gemm_group_checked(const snag_gemm_desc* const* descs, int n, hipStream_t st, const GroupTail* tail);      // snag_gemm_f32_group's body (+ an optional tail)
int gemm_dispatch_group(const snag_gemm_desc* const* ds, int n, hipStream_t st, const GroupTail* tail = nullptr);   // 2..8 problems, one launch when their plans allow (A layouts may differ per problem on the 64 x 128 tile)
// finishes problems launched with defer_reduce (+ optionally up to four small products in the same launch)
// (+ optionally a layer's slab reductions, independent of everything the launch reduces: a 7 us launch of their own otherwise)
int gemm_reduce_multi(const snag_gemm_desc* const* ds, int n, hipStream_t st, const snag_small_mm* small = nullptr, int n_small = 0, const SlabColsum* colsum = nullptr,
                      const RoWeightImages* images = nullptr);      // (+ optionally the 64 weight images of the read-out forward)
int64_t gemm_workspace_floats(int M, int N, int K);
// C = epilogue(A B) AND y = LayerNorm(C) (rows of y D = N wide, dense) in ONE launch, when the product runs as unsplit 64 x 128 tiles that hold
// whole rows (N <= 128; K < 512 or M < 1024; no workspace, accumulate, pre_act, mirror or transposed store).  Returns 1 -- nothing launched --
// when the problem does not qualify: the caller then runs gemm_dispatch + ln_fwd_kernel, which give the same bits.  stats[m] = (mean, rstd).
int gemm_dispatch_ln(const snag_gemm_desc* d, const float* gamma, const float* beta, float eps, float* y, float* stats, hipStream_t st);

}  // namespace snag
