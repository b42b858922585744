// snag_gemm_f32: the dense contraction behind the projections (a5), the Mformer linears (a2-a4),
// their backward passes, and (with the distance epilogue in eval.hip) the similarity GEMM (a8).
#include <cstdio>
#include <stdlib.h>

#include <algorithm>

#include "gemm.hip.h"
#include "small_mm.hip.h"

namespace snag {

struct Epilogue {
    float* C;
    int64_t ldc;
    const float* bias;      // [N]
    float alpha;
    int act;                // 0 none, 1 relu, 2 gelu(erf), 3 v *= gelu'(aux), 4 v *= (aux > 0)
    const float* aux;       // [M,N] ld = ldaux, for act 3/4 (backward through an activation)
    int64_t ldaux;
    float* pre;             // optional copy of alpha*acc + bias (before activation), same ld as C
    const uint8_t* mask;    // optional keep mask [M, N], row stride ldm bytes
    int64_t ldm;
    float mscale;
    const float* add1;
    int64_t ld1;
    const float* add2;
    int64_t ld2;
    int accumulate;         // C += result instead of C = result
    int M, N;
    int ct;                 // C is stored TRANSPOSED: value (m, n) goes to C[n * ldc + m] (element stores; no pre / mirror)
    int fast;               // host-decided: the dwordx4 path of operator()(m, n, float4) is legal
    float* mirror;          // optional second copy of the final values, rows scaled where mirror_mask[m] != 0
    int64_t ldmirror;
    const uint8_t* mirror_mask;
    float mirror_scale;

    __device__ __forceinline__ float one(int m, int n, float v) const {
        v *= alpha;
        if (bias) v += bias[n];
        if (pre && !ct) pre[(int64_t)m * ldc + n] = v;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        else if (act == 3) {
            const float u = aux[(int64_t)m * ldaux + n];
            // d/du [0.5 u (1 + erf(u/sqrt2))] = 0.5 (1 + erf(u/sqrt2)) + u exp(-u^2/2) / sqrt(2 pi)
            v *= 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * expf(-0.5f * u * u) * 0.39894228040143267794f;
        } else if (act == 4) {
            v = aux[(int64_t)m * ldaux + n] > 0.f ? v : 0.f;
        }
        if (mask) v = mask[(int64_t)m * ldm + n] ? v * mscale : 0.f;
        if (add1) v += add1[(int64_t)m * ld1 + n];
        if (add2) v += add2[(int64_t)m * ld2 + n];
        return v;
    }

    __device__ __forceinline__ void operator()(int m, int n, float v) const {
        if (m >= M || n >= N) return;
        v = one(m, n, v);
        float* c = ct ? C + (int64_t)n * ldc + m : C + (int64_t)m * ldc + n;
        v = accumulate ? *c + v : v;
        *c = v;
        if (mirror) mirror[(int64_t)m * ldmirror + n] = (mirror_mask && mirror_mask[m]) ? v * mirror_scale : v;
    }

    // four consecutive columns n..n+3 of row m
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const {
        if (m >= M || n >= N) return;
        if (fast && n + 3 < N) {
            // every operand present is 16-byte addressable (decided on the host): one dwordx4 per operand instead of four
            // scalar passes; per component the operation order is exactly one()'s, so both paths give the same bits
            float x[4] = {v.x * alpha, v.y * alpha, v.z * alpha, v.w * alpha};
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + n);
                x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
            }
            if (pre) *reinterpret_cast<float4*>(pre + (int64_t)m * ldc + n) = make_float4(x[0], x[1], x[2], x[3]);
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], 0.f);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752440f));
            } else if (act == 3 || act == 4) {
                const float4 u4 = *reinterpret_cast<const float4*>(aux + (int64_t)m * ldaux + n);
                const float u[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (act == 3) x[i] *= 0.5f * (1.0f + erff(u[i] * 0.70710678118654752440f)) + u[i] * expf(-0.5f * u[i] * u[i]) * 0.39894228040143267794f;
                    else x[i] = u[i] > 0.f ? x[i] : 0.f;
                }
            }
            if (mask) {
                const uchar4 k = *reinterpret_cast<const uchar4*>(mask + (int64_t)m * ldm + n);
                x[0] = k.x ? x[0] * mscale : 0.f;
                x[1] = k.y ? x[1] * mscale : 0.f;
                x[2] = k.z ? x[2] * mscale : 0.f;
                x[3] = k.w ? x[3] * mscale : 0.f;
            }
            if (add1) {
                const float4 a = *reinterpret_cast<const float4*>(add1 + (int64_t)m * ld1 + n);
                x[0] += a.x; x[1] += a.y; x[2] += a.z; x[3] += a.w;
            }
            if (add2) {
                const float4 a = *reinterpret_cast<const float4*>(add2 + (int64_t)m * ld2 + n);
                x[0] += a.x; x[1] += a.y; x[2] += a.z; x[3] += a.w;
            }
            float4* c = reinterpret_cast<float4*>(C + (int64_t)m * ldc + n);
            if (accumulate) {
                const float4 o = *c;
                x[0] = o.x + x[0]; x[1] = o.y + x[1]; x[2] = o.z + x[2]; x[3] = o.w + x[3];
            }
            *c = make_float4(x[0], x[1], x[2], x[3]);
            if (mirror) {
                const float ms = (mirror_mask && mirror_mask[m]) ? mirror_scale : 1.0f;
                *reinterpret_cast<float4*>(mirror + (int64_t)m * ldmirror + n) = make_float4(x[0] * ms, x[1] * ms, x[2] * ms, x[3] * ms);
            }
            return;
        }
        (*this)(m, n + 0, v.x);
        (*this)(m, n + 1, v.y);
        (*this)(m, n + 2, v.z);
        (*this)(m, n + 3, v.w);
    }
};

// sums the split-K slabs in slab order (deterministic) and applies the epilogue.  QUAD: N % 4 == 0 -- a thread owns four consecutive
// columns: dwordx4 slab reads (the slabs are the whole traffic: 4 x 7.7 MB for the visual projection) and the epilogue's dwordx4 path.
template <bool QUAD>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ slab, int splits, const Epilogue& epi, int64_t first, int64_t stride) {
    const int64_t total = (int64_t)epi.M * epi.N;
    if constexpr (QUAD) {
        const int64_t quads = total / 4;
        for (int64_t q = first; q < quads; q += stride) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int z = 0;
            for (; z + 4 <= splits; z += 4) {           // four independent 16-byte loads in flight, summed in slab order
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = reinterpret_cast<const float4*>(slab + (int64_t)(z + u) * total)[q];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
            }
            for (; z < splits; ++z) {
                const float4 t = reinterpret_cast<const float4*>(slab + (int64_t)z * total)[q];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const int64_t i = q * 4;
            epi((int)(i / epi.N), (int)(i % epi.N), v);
        }
    } else {
        for (int64_t i = first; i < total; i += stride) {
            float v = 0.f;
            int z = 0;
            for (; z + 8 <= splits; z += 8) {           // eight independent loads in flight, summed in slab order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = slab[(int64_t)(z + u) * total + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; z < splits; ++z) v += slab[(int64_t)z * total + i];
            epi((int)(i / epi.N), (int)(i % epi.N), v);
        }
    }
}

// Small outputs with many slabs (the D x D and F x D weight gradients: 16-128 K floats, 56-166 slabs): one workgroup = 64 quads,
// its four waves take every fourth slab each (ascending), the four partial sums meet in LDS in wave order -- a fixed order, four
// times the loads in flight.
// (tid, part: the thread's index inside its 256-thread block and that block's [3][64] staging rows -- a 512-thread workgroup runs two blocks side by side)
__device__ __forceinline__ void splitk_reduce_small_body(const float* __restrict__ slab, int splits, const Epilogue& epi, int64_t block, int tid, float4 (*part)[64]) {
    const int64_t total = (int64_t)epi.M * epi.N, quads = total / 4;
    const int lane = tid & 63, g = tid >> 6;
    const int64_t q = block * 64 + lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < quads) {
        int z = g;
        for (; z + 12 < splits; z += 16) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = reinterpret_cast<const float4*>(slab + (int64_t)(z + 4 * u) * total)[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
        }
        for (; z < splits; z += 4) {
            const float4 t = reinterpret_cast<const float4*>(slab + (int64_t)z * total)[q];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
    }
    if (g > 0) part[g - 1][lane] = v;
    __syncthreads();
    if (g == 0 && q < quads) {
#pragma unroll
        for (int u = 0; u < 3; ++u) { const float4 t = part[u][lane]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        const int64_t i = q * 4;
        epi((int)(i / epi.N), (int)(i % epi.N), v);
    }
}
__global__ __launch_bounds__(256) void splitk_reduce_small_kernel(const float* __restrict__ slab, int splits, Epilogue epi) {
    __shared__ float4 part[3][64];
    splitk_reduce_small_body(slab, splits, epi, blockIdx.x, threadIdx.x, part);
}
__global__ __launch_bounds__(256) void splitk_reduce_small_pair_kernel(const float* __restrict__ slab0, Epilogue e0, const float* __restrict__ slab1, Epilogue e1,
                                                                       int splits) {
    __shared__ float4 part[3][64];
    splitk_reduce_small_body(blockIdx.y ? slab1 : slab0, splits, blockIdx.y ? e1 : e0, blockIdx.x, threadIdx.x, part);
}
static bool reduce_small(int64_t total, int splits) { return total <= (1 << 18) && splits >= 8; }

// Up to four problems of different shape and split count in one reduce launch (snag_gemm_reduce_pair / _group, the Mformer backward's weight
// gradients): blockIdx.y = problem, each reduced by the body -- and in the summation order -- its own snag_gemm_f32 call would have
// used, over its own block count.  mode: 0 = one column per thread, 1 = four columns per thread, 2 = the small-output slab-parallel form
struct ReduceSet {
    const float* slab[4];
    int splits[4], mode[4], nblk[4];
    Epilogue e[4];
};
// A 1-D grid: the riders that are present, then the reduce problems' workgroups one problem after the other.  Riders: the small products of
// snag_gemm_reduce_group_mm (one 16 x 16 output tile per workgroup), a layer's slab reductions (2 x ceil(3 D / 16) workgroups), the 64
// weight images of the one-launch read-out forward.  (A problem-per-row grid padded every row to the longest one: 3,686 empty workgroups
// beside the 64 of the weight images.)
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(ReduceSet rs, int n_reduce, SmallMMSet mm, int mm_blocks, SlabColsum cs, int cs_blocks,
                                                                  RoWeightImages wi, int wi_blocks) {
    int b = (int)blockIdx.x;          // (riders first: latency-bound work starts with the launch and hides behind the reduce's stream)
    if (b < mm_blocks) {
        small_matmul_body(mm, b);
        return;
    }
    b -= mm_blocks;
    if (b < cs_blocks) {
        __shared__ float red[16][16];
        const int ncb = cs_blocks / 2;
        slab_colsum_body(cs, b % ncb, b / ncb, red, threadIdx.x);
        return;
    }
    b -= cs_blocks;
    if (b < wi_blocks) {
        ro_weight_image_body(wi, b);
        return;
    }
    b -= wi_blocks;
    for (int q = 0; q < n_reduce; ++q) {
        if (b < rs.nblk[q]) {
            if (rs.mode[q] == 2) {
                __shared__ float4 part[3][64];
                splitk_reduce_small_body(rs.slab[q], rs.splits[q], rs.e[q], b, threadIdx.x, part);
            } else if (rs.mode[q] == 1) splitk_reduce_body<true>(rs.slab[q], rs.splits[q], rs.e[q], (int64_t)b * 256 + threadIdx.x, (int64_t)rs.nblk[q] * 256);
            else splitk_reduce_body<false>(rs.slab[q], rs.splits[q], rs.e[q], (int64_t)b * 256 + threadIdx.x, (int64_t)rs.nblk[q] * 256);
            return;
        }
        b -= rs.nblk[q];
    }
}

// A group of eight-wave 256-row tiles + a TAIL: blocks [0, n_main) are the group's tiles; the blocks behind them -- as many as the launch leaves
// CUs idle, one 512-thread workgroup of this kernel fills a CU -- walk deferred split-K reduces and a layer's slab column sums that depend on
// nothing the group computes (the read-out layer's weight-gradient reduce beside the first projections' weight-gradient GEMM: a 9 us launch of
// its own otherwise).  A tail workgroup is two 256-thread blocks side by side; every element is summed by the body and in the order of
// splitk_reduce_multi_kernel: the same bits.
struct TailSet {
    ReduceSet rs;
    int n_reduce;
    SlabColsum cs;
    int cs_blocks;
};
template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(512) void gemm_tile_group_tail_kernel(GemmGroup<Epilogue> g, int n_main, TailSet t) {
    using Core = GemmCore<256, 128, 32, 128, A_KC, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float smem[Core::SMEM_FLOATS];
    static_assert(Core::SMEM_FLOATS >= 2 * 256 + 2 * 3 * 64 * 4, "the tail's staging rows fit in the tile buffers");
    if ((int)blockIdx.x < n_main) {
        gemm_group_block<256, 128, 32, 128, A_KC, B_KC, VEC, Epilogue>(g, (int)blockIdx.x, smem);
        return;
    }
    const int r = (int)blockIdx.x - n_main, R = (int)gridDim.x - n_main;
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
    float (*red)[16] = reinterpret_cast<float (*)[16]>(smem + half * 256);
    float4 (*part)[64] = reinterpret_cast<float4 (*)[64]>(smem + 512 + half * 3 * 64 * 4);
    // slab column sums (cs_blocks is even: both halves always hold a block)
    const int ncb = t.cs_blocks / 2;
    for (int vb0 = 2 * r; vb0 < t.cs_blocks; vb0 += 2 * R) {
        const int vb = vb0 + half;
        slab_colsum_body(t.cs, vb % ncb, vb / ncb, red, tid);
        __syncthreads();          // (the staging rows are written again by the next block)
    }
    for (int q = 0; q < t.n_reduce; ++q) {
        const ReduceSet& rs = t.rs;
        if (rs.mode[q] == 2) {
            for (int vb0 = 2 * r; vb0 < rs.nblk[q]; vb0 += 2 * R) {          // (a block index past the end: no quad of its own, the barriers stay)
                splitk_reduce_small_body(rs.slab[q], rs.splits[q], rs.e[q], vb0 + half, tid, part);
                __syncthreads();
            }
        } else if (rs.mode[q] == 1) {
            splitk_reduce_body<true>(rs.slab[q], rs.splits[q], rs.e[q], (int64_t)r * 512 + threadIdx.x, (int64_t)R * 512);
        } else {
            splitk_reduce_body<false>(rs.slab[q], rs.splits[q], rs.e[q], (int64_t)r * 512 + threadIdx.x, (int64_t)R * 512);
        }
    }
}

template <bool QUAD>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, Epilogue epi) {
    splitk_reduce_body<QUAD>(slab, splits, epi, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// one problem + the 64 (72) weight images of the one-launch read-out forward (and backward) in front (the generic multi-problem kernel is ~2 us slower on the
// same reduce: four epilogues in the kernel arguments, indexed at run time)
template <bool QUAD>
__global__ __launch_bounds__(256) void splitk_reduce_images_kernel(const float* __restrict__ slab, int splits, Epilogue epi, RoWeightImages wi) {
    const int ni = wi.n_images;
    if ((int)blockIdx.x < ni) {
        ro_weight_image_body(wi, (int)blockIdx.x);
        return;
    }
    splitk_reduce_body<QUAD>(slab, splits, epi, (int64_t)(blockIdx.x - ni) * 256 + threadIdx.x, (int64_t)(gridDim.x - ni) * 256);
}

// the same for the two problems of gemm_tile_pair_kernel (blockIdx.y = problem)
template <bool QUAD>
__global__ __launch_bounds__(256) void splitk_reduce_pair_kernel(const float* __restrict__ slab0, Epilogue e0, const float* __restrict__ slab1, Epilogue e1,
                                                                 int splits) {
    splitk_reduce_body<QUAD>(blockIdx.y ? slab1 : slab0, splits, blockIdx.y ? e1 : e0, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// S = alpha * Z Z^T (Z [n, k], K-contiguous).  Only tiles on or above the diagonal run the main loop; each also writes its
// transpose (staged through LDS so that both copies leave in full 128-byte lines) -- a diagonal tile the transpose of its own
// upper triangle -- so S is symmetric bit for bit and the MFMA work is (almost) halved.
__global__ __launch_bounds__(256) void gram_tile_kernel(GemmOperands p, float alpha, float* __restrict__ C, int64_t ldc, int T) {
    // 1-D grid over the T (T + 1) / 2 tiles on or above the diagonal, row-major; the XCD remap hands each L2 a contiguous run
    const int k = p.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
    int by = (int)(((2.0f * T + 1.0f) - sqrtf((2.0f * T + 1.0f) * (2.0f * T + 1.0f) - 8.0f * k)) * 0.5f);
    by = max(0, min(by, T - 1));
    while (by > 0 && by * T - by * (by - 1) / 2 > k) --by;
    while (by + 1 < T && (by + 1) * T - (by + 1) * by / 2 <= k) ++by;
    const int bx = by + (k - (by * T - by * (by - 1) / 2));
    using Core = GemmCore<128, 128, 64, 64, true, true, false>;
    constexpr int kSmem = Core::SMEM_FLOATS > Core::kTransposeStageFloats ? Core::SMEM_FLOATS : Core::kTransposeStageFloats;
    __shared__ __attribute__((aligned(16))) float smem[kSmem];
    const int m0 = by * 128, n0 = bx * 128;
    Core core;
    core.run(p, smem, m0, n0, 0, p.K);
    const int n = p.M;
    const bool vec = (ldc % 4 == 0) && ((((uintptr_t)C) & 15) == 0);
    // A diagonal tile holds S[i][j] and S[j][i] as two separately accumulated values (with split-bf16 products a.b and b.a differ in
    // their last bits): only its elements on or above the diagonal are stored, the transposed pass below mirrors them.
    const bool diag = bx == by;
    core.for_each(smem, m0, n0, [=](int r, int c, float4 v) {
        if (r >= n || c >= n) return;
        float* o = C + (int64_t)r * ldc + c;
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        if (diag && c < r) {            // the quad starts left of the diagonal: element-wise, upper part only
            if (c + 1 >= r && c + 1 < n) o[1] = v.y;
            if (c + 2 >= r && c + 2 < n) o[2] = v.z;
            if (c + 3 >= r && c + 3 < n) o[3] = v.w;
            return;
        }
        if (vec && c + 3 < n) {
            *reinterpret_cast<float4*>(o) = v;
        } else {
            o[0] = v.x;
            if (c + 1 < n) o[1] = v.y;
            if (c + 2 < n) o[2] = v.z;
            if (c + 3 < n) o[3] = v.w;
        }
    });
    // the mirror: C[col, row], 128-byte runs along the transposed rows (a diagonal tile mirrors its strict upper part)
    core.for_each_transposed(smem, m0, n0, [=](int c, int r, float v) {
        if (r < n && c < n && (!diag || c > r)) C[(int64_t)c * ldc + r] = alpha * v;
    });
}

// VEC: every fetch is an unpredicated dwordx4 (all shapes of the hot path); otherwise the checked loader
// Measured on one device (tools/tune_gemm.py, A/B builds): the unpredicated loader wins 15-25 % whenever an operand is
// row-contiguous, but loses 5-20 % when both are K-contiguous, so those keep the checked loader.
// The product AND the LayerNorm of its output rows in one launch (gemm_dispatch_ln): a 64 x 128 tile holds whole rows when N <= 128 (the
// output projection of a D <= 128 Mformer layer: dense + dropout + residual, then LayerNorm -- model/SNAG_tools.py:218-222).  The
// accumulators go through the LDS as one [64][132] tile; wave w then owns rows 16 w .. 16 w + 15: the epilogue value of column
// lane + 64 u (Epilogue::one: bias, keep mask, residual), the row written to C (the backward reads the pre-LayerNorm sum), the row's
// LayerNorm by ln_row_forward -- the arithmetic of ln_fwd_kernel, bit for bit -- written to y, (mean, rstd) to stats.
struct LnTail {
    const float *g, *b;
    float eps;
    float* y;
    float* stats;
};
template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(256) void gemm_tile_ln_kernel(GemmOperands p, Epilogue e, LnTail t) {
    using Core = GemmCore<64, 128, 32, 64, A_KC, B_KC, VEC>;
    constexpr int P = 132;
    constexpr int kSmem = Core::SMEM_FLOATS > 64 * P ? Core::SMEM_FLOATS : 64 * P;
    __shared__ __attribute__((aligned(16))) float smem[kSmem];
    const int m0 = blockIdx.x * 64;
    Core core;
    core.run(p, smem, m0, 0, 0, p.K);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int li = lane & 31, lh = lane >> 5, wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
    lds_float* T = (lds_float*)smem;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) T[(wm0 + (q & 3) + 8 * (q >> 2) + 4 * lh) * P + wn0 + 32 * j + li] = core.acc[0][j][q];
    __syncthreads();
    const int N = p.N;
    // Epilogue::one for act = 0 without pre / add2 (gemm_dispatch_ln admits nothing else), in its operation order: alpha, bias, keep mask, add1.
    // Eight rows at a time, the side inputs of all eight requested before any is used: one round trip per group, not one per row.
    const bool hb = e.bias != nullptr, hm = e.mask != nullptr, ha = e.add1 != nullptr;
    float bv[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (hb && lane + 64 * u < N) bv[u] = e.bias[lane + 64 * u];
    for (int r0 = 0; r0 < 16; r0 += 8) {
        float x[8][2], av[8][2];
        uint8_t kv[8][2];
        if (hm) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int m = min(m0 + wave * 16 + r0 + q, p.M - 1), c = min(lane + 64 * u, N - 1);         // (clamped: rows / columns past the edge are never stored)
                    kv[q][u] = e.mask[(int64_t)m * e.ldm + c];
                }
        }
        if (ha) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int m = min(m0 + wave * 16 + r0 + q, p.M - 1), c = min(lane + 64 * u, N - 1);
                    av[q][u] = e.add1[(int64_t)m * e.ld1 + c];
                }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = lane + 64 * u;
                float v = T[(wave * 16 + r0 + q) * P + min(c, 127)] * e.alpha;
                if (hb) v += bv[u];
                if (hm) v = kv[q][u] ? v * e.mscale : 0.f;
                if (ha) v += av[q][u];
                x[q][u] = c < N ? v : 0.f;
            }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = m0 + wave * 16 + r0 + q;
            if (m < p.M) {                        // (wave-uniform)
                float o[2], mean, rstd;
                ln_row_forward<2>(x[q], N, lane, t.g, t.b, t.eps, o, mean, rstd);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = lane + 64 * u;
                    if (c < N) {
                        e.C[(int64_t)m * e.ldc + c] = x[q][u];
                        t.y[(int64_t)m * N + c] = o[u];
                    }
                }
                if (lane == 0) {
                    t.stats[2 * (int64_t)m] = mean;
                    t.stats[2 * (int64_t)m + 1] = rstd;
                }
            }
        }
    }
}

// presplit: the tile runs the pre-split loops (every tile a split-bf16 build instantiates), which look at VEC for K-contiguous operands
// only; the f32-MFMA loops (run_loop: -DSNAG_GEMM_BF16X=0 builds, the 128 x 160 tile) issue unpredicated float4 loads under VEC and
// need the full shape test.
static bool use_vec(const GemmOperands& p, bool a_kc, bool b_kc, bool presplit = kBf16x != 0) {
    // 16-byte loads as a compile-time choice whenever they are legal -- the loop's waits are then counted (vmcnt(9) in the ping-pong
    // loop) instead of one vmcnt(0) every fourth iteration; same box, in the step: 0.6206 / 0.6251 / 0.6268 vs 0.6280 / 0.6303 / 0.6297 ms
    if (presplit) return (!a_kc || p.a_vec) && (!b_kc || p.b_vec);
    const bool legal = p.a_vec && p.b_vec && (a_kc || p.M % 4 == 0) && (b_kc || p.N % 4 == 0) && p.M >= 4 && p.N >= 4;
    return legal && !(a_kc && b_kc);
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg_pair(const GemmOperands& p0, const Epilogue& e0, float* slab0, const GemmOperands& p1, const Epilogue& e1, float* slab1,
                            int splits, bool a_kc, bool b_kc, bool vec, hipStream_t st) {
    dim3 grid(ceil_div(p0.N, BN), ceil_div(p0.M, BM), 2 * splits), block(64 * (BM / WM) * (BN / WN));
#define SNAG_GEMM2(AK, BK_)                                                                                        \
    do {                                                                                                           \
        if (vec) hipLaunchKernelGGL((gemm_tile_pair_kernel<BM, BN, WM, WN, AK, BK_, true, Epilogue>), grid, block, 0, st, p0, e0, slab0, p1, e1, slab1, splits);  \
        else hipLaunchKernelGGL((gemm_tile_pair_kernel<BM, BN, WM, WN, AK, BK_, false, Epilogue>), grid, block, 0, st, p0, e0, slab0, p1, e1, slab1, splits);     \
    } while (0)
    if (a_kc && b_kc) SNAG_GEMM2(true, true);
    else if (a_kc && !b_kc) SNAG_GEMM2(true, false);
    else if (!a_kc && b_kc) SNAG_GEMM2(false, true);
    else SNAG_GEMM2(false, false);
#undef SNAG_GEMM2
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const GemmOperands& p, const Epilogue& e, float* slab, int splits, bool a_kc, bool b_kc, hipStream_t st) {
    dim3 grid(ceil_div(p.N, BN), ceil_div(p.M, BM), splits), block(64 * (BM / WM) * (BN / WN));
    const bool vec = use_vec(p, a_kc, b_kc, GemmCore<BM, BN, WM, WN, true, true, true>::kPresplit);
#define SNAG_GEMM(AK, BK_)                                                                                        \
    do {                                                                                                          \
        if (vec) hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WM, WN, AK, BK_, true, Epilogue>), grid, block, 0, st, p, e, slab);  \
        else hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WM, WN, AK, BK_, false, Epilogue>), grid, block, 0, st, p, e, slab);     \
    } while (0)
    if (a_kc && b_kc) SNAG_GEMM(true, true);
    else if (a_kc && !b_kc) SNAG_GEMM(true, false);
    else if (!a_kc && b_kc) SNAG_GEMM(false, true);
    else SNAG_GEMM(false, false);
#undef SNAG_GEMM
}

static constexpr int kStreamKBlocks = 512;          // 2 per CU (measured: 512 < 768 < 1024 in time; 256 / 384 much worse)

struct StreamKChoice {
    int G;
    StreamKPlan plan;
    int64_t slab_floats;
};
// G persistent blocks over tiles x nkt iterations; each block keeps at least kMinKTiles iterations
static StreamKChoice plan_streamk(int M, int N, int K, int BM, int BN) {
    StreamKChoice c;
    const int tiles_m = ceil_div(M, BM);
    c.plan.tiles_n = ceil_div(N, BN);
    c.plan.nkt = std::max(1, ceil_div(K, kBK));          // K == 0 never takes this route; keep the arithmetic below defined
    c.plan.total = (int64_t)tiles_m * c.plan.tiles_n * c.plan.nkt;
    int64_t G = kStreamKBlocks;
    while (G > 256 && c.plan.total / G < 16) G -= 256;
    c.G = (int)std::min<int64_t>(G, std::max<int64_t>(1, c.plan.total));
    const int64_t len = ceil_div(c.plan.total, (int64_t)c.G);
    c.plan.max_seg = (int)(ceil_div(len, (int64_t)c.plan.nkt) + 1);
    c.slab_floats = (int64_t)c.G * c.plan.max_seg * BM * BN;
    return c;
}

template <int BM, int BN, int WM, int WN>
static int launch_streamk(const GemmOperands& p, const Epilogue& e, float* slab, const StreamKChoice& c, bool a_kc, bool b_kc, hipStream_t st) {
    const bool vec = use_vec(p, a_kc, b_kc, GemmCore<BM, BN, WM, WN, true, true, true>::kPresplit);
    dim3 grid(c.G), block(256);
#define SNAG_SK(AK, BK_)                                                                                                           \
    do {                                                                                                                           \
        if (vec) hipLaunchKernelGGL((gemm_streamk_kernel<BM, BN, WM, WN, AK, BK_, true, Epilogue>), grid, block, 0, st, p, e, slab, c.plan);  \
        else hipLaunchKernelGGL((gemm_streamk_kernel<BM, BN, WM, WN, AK, BK_, false, Epilogue>), grid, block, 0, st, p, e, slab, c.plan);     \
    } while (0)
    if (a_kc && b_kc) SNAG_SK(true, true);
    else if (a_kc && !b_kc) SNAG_SK(true, false);
    else if (!a_kc && b_kc) SNAG_SK(false, true);
    else SNAG_SK(false, false);
#undef SNAG_SK
    int rc = check_launch("snag_gemm_f32(stream-K)");
    if (rc) return rc;
    const int tiles = ceil_div(p.M, BM) * c.plan.tiles_n;
    hipLaunchKernelGGL((streamk_fixup_kernel<BM, BN, Epilogue>), dim3(tiles, BM * BN / 1024), dim3(256), 0, st, (const float*)slab, e, c.plan, c.G);
    return check_launch("snag_gemm_f32(stream-K fix-up)");
}

// Split K when the output tiles alone leave the chip under-filled: with one 4-wave block per CU only one
// 12-16 KB tile is in flight per CU and the loop runs at HBM latency, not bandwidth; ~4 co-resident
// blocks per CU hide it.  Each split keeps at least kMinKTiles k-tiles of work.
static constexpr int kTargetBlocks = 1024, kMinKTiles = 16;

static int pick_splits(int64_t tiles, int K, bool both_kc = false) {
    if (tiles >= kTargetBlocks / 2) return 1;
    if (both_kc && tiles >= 192) return 1;      // measured: X W^T shapes with ~1 tile per CU gain nothing from splitting
    const int64_t want = (kTargetBlocks + tiles - 1) / tiles;
    const int64_t cap = std::max<int64_t>(1, K / (kMinKTiles * 16));
    return (int)std::min(want, cap);
}

static int log_shapes() { static const int v = env_int("SNAG_GEMM_LOG"); return v; }          // diagnostics: one line per launch on stderr

int64_t gemm_workspace_floats(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // the largest fan-out any tile configuration may pick for this shape (the dispatcher clamps to what it is given)
    const int64_t t64 = (int64_t)ceil_div(M, 64) * ceil_div(N, 128), t160 = (int64_t)ceil_div(M, 128) * ceil_div(N, 160);
    const int splits = std::max(pick_splits(t64, K), pick_splits(t160, K));
    const int64_t sk = plan_streamk(M, N, K, 64, 128).slab_floats;
    return splits > 1 ? std::max((int64_t)splits * M * N, sk) : 0;
}

// What the dispatcher decided for one problem (everything the launch needs)
struct GemmPlan {
    GemmOperands p;
    Epilogue e;
    bool big, wide, streamk, pp;
    int splits;
    StreamKChoice sk;
};

static GemmPlan plan_gemm(const snag_gemm_desc* d) {
    GemmPlan pl{};
    GemmOperands& p = pl.p;
    Epilogue& e = pl.e;
    p.A = d->A; p.B = d->B; p.lda = d->lda; p.ldb = d->ldb; p.M = d->M; p.N = d->N; p.K = d->K;
    p.a_vec = (d->lda % 4 == 0) && (((uintptr_t)d->A & 15) == 0);
    p.b_vec = (d->ldb % 4 == 0) && (((uintptr_t)d->B & 15) == 0);
    p.dbg = 0;
    p.xcd = 1;                  // XCD-aware workgroup -> tile mapping (xcd_swizzle)
    e.C = d->C; e.ldc = d->ldc; e.bias = d->bias; e.alpha = d->alpha; e.act = d->act; e.pre = d->pre_act;
    e.mask = d->keep_mask; e.ldm = d->ldmask > 0 ? d->ldmask : d->N; e.mscale = d->keep_scale; e.add1 = d->add1; e.ld1 = d->ld1; e.add2 = d->add2; e.ld2 = d->ld2;
    e.accumulate = d->accumulate; e.M = d->M; e.N = d->N; e.aux = d->aux; e.ldaux = d->ldaux; e.ct = d->c_transposed;
    e.mirror = d->mirror; e.ldmirror = d->ldmirror; e.mirror_mask = d->mirror_rowmask; e.mirror_scale = d->mirror_scale;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    e.fast = (d->ldc % 4 == 0) && al16(d->C) && (!d->bias || al16(d->bias)) && (!d->pre_act || al16(d->pre_act)) &&
             (!d->add1 || (al16(d->add1) && d->ld1 % 4 == 0)) && (!d->add2 || (al16(d->add2) && d->ld2 % 4 == 0)) &&
             (d->act < 3 || (al16(d->aux) && d->ldaux % 4 == 0)) && (!d->keep_mask || (e.ldm % 4 == 0 && ((uintptr_t)d->keep_mask & 3) == 0)) &&
             (!d->mirror || (al16(d->mirror) && d->ldmirror % 4 == 0)) && !d->c_transposed;

    // Tile choice by padded work x a measured per-configuration factor (tools/tune_gemm.py, tools/tune_gemm_mmea.py, one device):
    //   128x128  when its tiles alone fill the chip; the best loop when both operands are K-contiguous;
    //   128x160  (one wave = 32 x 160) pads N = 300 to 320 instead of 384 and is the fastest loop whenever an operand is
    //            row-contiguous (LDS-DMA path); with both K-contiguous its checked loader costs ~18 %;
    //   64x128   otherwise (twice the blocks for small outputs).
    const bool kk = d->a_kcontig && d->b_kcontig;
    const int64_t pm64 = (int64_t)ceil_div(p.M, 64) * 64, pm128 = (int64_t)ceil_div(p.M, 128) * 128;
    const int64_t pn128 = (int64_t)ceil_div(p.N, 128) * 128, pn160 = (int64_t)ceil_div(p.N, 160) * 160;
    const int64_t big_tiles = (pm128 / 128) * (pn128 / 128), wide_tiles = (pm128 / 128) * (pn160 / 160);
    const double c_small = (double)pm64 * pn128 * (kk ? 1.08 : 1.10);
    const double c_big = (double)pm128 * pn128 * (kk ? 1.00 : 1.05);
    const double c_wide = (double)pm128 * pn160 * (kk ? 1.18 : 1.00);
    const int64_t max_splits = d->workspace ? std::max<int64_t>(1, p.K / (kMinKTiles * 16)) : 1;
    bool big = big_tiles >= 512 && c_big <= c_small;
    bool wide = wide_tiles * max_splits >= 512 && c_wide < (big ? c_big : c_small);
    // Split-bf16 builds: the pre-split staging loop (gemm.hip.h) runs the 64 x 128 tile fastest on every shape of the path
    // (forced 64 x 128 vs the choice above, one device: DB15K step 0.715 -> 0.704 ms, MKG-W 0.799 -> 0.788, MMEA step 22.5 -> 21.4 ms,
    // 10,500^2 x 1,200 similarity eval 7.8 -> 7.2 ms); the larger tiles are what -DSNAG_GEMM_BF16X=0 builds run.
    // (128 x 128 for the long-K shapes only, K >= 512 or 2048: step 0.708 / 0.711 vs 0.697 ms -- no.)
    if (kBf16x != 0) big = wide = false;
    // Long K (the projections and their weight gradients): 128 x 128 "row-private" tiles -- four waves stacked along M, each wave's
    // 32 rows of A go straight to its registers, only B goes through the LDS (44 % less LDS traffic per multiply-add, half the
    // re-reads of B per launch).  For K = 128 / 256 the 64 x 128 tile stays (twice the workgroups for the same output).
    // (per shape, in the step: [15000,128,4096] 110 -> 92 us, [4096,128,15000] 94 -> 87, [15000,128,768] 29.7 -> 28.6; but
    // [128,128,15000] x 2 16.4 -> 20.6, [256,128,45000] 30.5 -> 31.6: long K AND many rows only.)
    if (kBf16x != 0 && p.K >= 512 && p.M >= 1024) big = true;
    if (kBK != 16) wide = false;        // the 128x160 tiles of a 32-deep k-tile do not fit the 64 KB static LDS limit
    if (wide) big = false;
    // the eight-wave ping-pong form of the row-private tile (256 rows, one workgroup per CU) where the output is one tile wide, i.e.
    // A is read exactly once (the projections): DB15K step 0.651 vs 0.657 ms, MKG-W 0.727 vs 0.737; with several column tiles
    // (MMEA: N = 300 / 1,200) the four-wave form is faster (MMEA step 20.6 vs 21.8 ms; only for shapes with >= 256 / >= 1,024 tiles
    // of 256 rows, in grouped order: 21.3 / 20.6-20.7 vs 20.2-20.6); the 10,500^2 similarity GEMM, 83 column
    // tiles wide, does gain (eval.hip picks the tile itself)
    const bool pp = big && kBf16x != 0 && p.N <= 128;
    const int64_t tall_tiles = (int64_t)ceil_div(p.M, pp ? 256 : 128) * (pn128 / 128);
    const int64_t tiles = wide ? wide_tiles : big ? (kBf16x != 0 ? tall_tiles : big_tiles) : (pm64 / 64) * (pn128 / 128);
    // split K when the output alone cannot fill 256 CUs (the dW = dY^T X shapes: tiny output, long K)
    int splits = 1;
    if (d->workspace != nullptr) {
        splits = pick_splits(tiles, p.K, d->a_kcontig && d->b_kcontig);
        // row-private tiles: at most ONE resident wave of workgroups -- 512 four-wave ones (two per CU), 256 eight-wave ones (measured
        // with four waves: 590 workgroups 0.7135 ms per step, 472-512 0.702, against 0.720 with 64 x 128 tiles)
        if (d->splits > 0) splits = (int)std::min<int64_t>(d->splits, std::max<int64_t>(1, p.K / (kMinKTiles * 16)));
        else if (kBf16x != 0 && big)
            splits = (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(1, p.K / (kMinKTiles * 16)),
                                                                (pp ? 256 : 512) / std::max<int64_t>(1, tiles)));
        const int64_t cap = d->workspace_floats / ((int64_t)p.M * p.N);
        if (splits > cap) splits = (int)cap;
        if (splits < 1) splits = 1;
    }
    // K == 0 is legal (C = epilogue(0): bias / add1 / accumulate still apply): one split with an empty k-range; the split
    // arithmetic below would otherwise divide by zero
    if (p.K == 0) splits = 1;
    pl.big = big; pl.wide = wide; pl.streamk = false; pl.pp = pp;
    if (splits > 1 && !big && !wide && d->splits <= 0) {      // (an explicit split request is a split-K request)
        // blocks per CU of the split grid vs what the slowest CU gets
        const double per_cu = (double)tiles * splits / 256.0;
        const double balance = per_cu / std::ceil(per_cu);
        const StreamKChoice c = plan_streamk(p.M, p.N, p.K, 64, 128);
        // few tiles shared by many blocks (the dW shapes) keep split-K: their slabs reduce in parallel over the whole output
        if (balance < 0.95 && c.G >= 256 && tiles * 8 >= c.G && c.slab_floats <= d->workspace_floats) {
            p.k_per_split = 0;
            pl.streamk = true;
            pl.sk = c;
            pl.splits = splits;
            return pl;
        }
    }
    const int kps = std::max(kBK, ceil_div(ceil_div(p.K, splits), kBK) * kBK);
    splits = std::max(1, ceil_div(p.K, kps));
    p.k_per_split = kps;
    pl.splits = splits;
    return pl;
}

int gemm_dispatch(const snag_gemm_desc* d, hipStream_t st) {
    const GemmPlan pl = plan_gemm(d);
    const GemmOperands& p = pl.p;
    const Epilogue& e = pl.e;
    if (pl.streamk) {
        if (log_shapes())
            fprintf(stderr, "snag_gemm M=%d N=%d K=%d a_kc=%d b_kc=%d tile=64x128 streamk=%d fast=%d acc=%d\n", p.M, p.N, p.K, (int)d->a_kcontig,
                    (int)d->b_kcontig, pl.sk.G, (int)e.fast, (int)d->accumulate);
        return launch_streamk<64, 128, 32, 64>(p, e, d->workspace, pl.sk, d->a_kcontig, d->b_kcontig, st);
    }
    const int splits = pl.splits;
    float* slab = splits > 1 ? d->workspace : nullptr;
    if (log_shapes())
        fprintf(stderr, "snag_gemm M=%d N=%d K=%d a_kc=%d b_kc=%d tile=%s splits=%d fast=%d acc=%d\n", p.M, p.N, p.K, (int)d->a_kcontig,
                (int)d->b_kcontig, pl.wide ? "128x160" : pl.pp ? "256x128" : pl.big ? "128x128" : "64x128", splits, (int)e.fast, (int)d->accumulate);
    if (pl.big && pl.pp) launch_cfg<256, 128, 32, 128>(p, e, slab, splits, d->a_kcontig, d->b_kcontig, st);
    else if (pl.big) launch_cfg<128, 128, 32, 128>(p, e, slab, splits, d->a_kcontig, d->b_kcontig, st);
    else launch_cfg<64, 128, 32, 64>(p, e, slab, splits, d->a_kcontig, d->b_kcontig, st);
    int rc = check_launch("snag_gemm_f32");
    if (rc) return rc;
    if (splits > 1 && !d->defer_reduce) {
        const int64_t total = (int64_t)p.M * p.N;
        const bool quad = p.N % 4 == 0 && ((uintptr_t)slab & 15) == 0;         // (slab z starts z M N floats in: a multiple of 16 bytes)
        const int blocks = (int)std::min<int64_t>(2048, (total / (quad ? 4 : 1) + 255) / 256);
        if (quad && reduce_small(total, splits))
            hipLaunchKernelGGL(splitk_reduce_small_kernel, dim3((unsigned)((total / 4 + 63) / 64)), dim3(256), 0, st, slab, splits, e);
        else if (quad) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, slab, splits, e);
        else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, slab, splits, e);
        rc = check_launch("snag_gemm_f32(split-K reduce)");
    }
    return rc;
}

int gemm_dispatch_ln(const snag_gemm_desc* d, const float* gamma, const float* beta, float eps, float* y, float* stats, hipStream_t st) {
    if (d->M <= 0 || d->N <= 0 || d->N > 128 || d->workspace || d->accumulate || d->pre_act || d->mirror || d->c_transposed || d->act != 0 || d->add2) return 1;
    const GemmPlan pl = plan_gemm(d);
    if (pl.streamk || pl.big || pl.wide || pl.splits != 1) return 1;
    const bool vec = use_vec(pl.p, d->a_kcontig, d->b_kcontig, GemmCore<64, 128, 32, 64, true, true, true>::kPresplit);
    const LnTail t{gamma, beta, eps, y, stats};
    if (log_shapes())
        fprintf(stderr, "snag_gemm+LN M=%d N=%d K=%d a_kc=%d b_kc=%d tile=64x128\n", pl.p.M, pl.p.N, pl.p.K, (int)d->a_kcontig, (int)d->b_kcontig);
    const dim3 grid(ceil_div(pl.p.M, 64)), block(256);
#define SNAG_GEMM_LN(AK, BK_)                                                                                          \
    do {                                                                                                               \
        if (vec) hipLaunchKernelGGL((gemm_tile_ln_kernel<AK, BK_, true>), grid, block, 0, st, pl.p, pl.e, t);          \
        else hipLaunchKernelGGL((gemm_tile_ln_kernel<AK, BK_, false>), grid, block, 0, st, pl.p, pl.e, t);             \
    } while (0)
    if (d->a_kcontig && d->b_kcontig) SNAG_GEMM_LN(true, true);
    else if (d->a_kcontig && !d->b_kcontig) SNAG_GEMM_LN(true, false);
    else if (!d->a_kcontig && d->b_kcontig) SNAG_GEMM_LN(false, true);
    else SNAG_GEMM_LN(false, false);
#undef SNAG_GEMM_LN
    return check_launch("snag_gemm_f32 + LayerNorm");
}

// Two problems in one launch when the dispatcher decides the same for both (same shape, layouts, tile, split count, loader);
// anything else runs them one after the other.
int gemm_dispatch_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, hipStream_t st) {
    const GemmPlan a = plan_gemm(d0), b = plan_gemm(d1);
    const bool same = !a.streamk && !b.streamk && !a.wide && !b.wide && a.big == b.big && !a.pp && !b.pp && a.splits == b.splits && d0->M == d1->M && d0->N == d1->N &&
                      d0->K == d1->K && d0->a_kcontig == d1->a_kcontig && d0->b_kcontig == d1->b_kcontig && a.p.k_per_split == b.p.k_per_split &&
                      use_vec(a.p, d0->a_kcontig, d0->b_kcontig) == use_vec(b.p, d1->a_kcontig, d1->b_kcontig) &&
                      (a.splits == 1 || (d0->workspace != d1->workspace));
    // different shapes, same layouts, 64 x 128 tiles, no K split on either side: one grouped launch
    const bool group = !same && !a.streamk && !b.streamk && !a.wide && !b.wide && !a.big && !b.big && a.splits == 1 && b.splits == 1 &&
                       d0->K > 0 && d1->K > 0 && d0->a_kcontig == d1->a_kcontig && d0->b_kcontig == d1->b_kcontig &&
                       use_vec(a.p, d0->a_kcontig, d0->b_kcontig) == use_vec(b.p, d1->a_kcontig, d1->b_kcontig);
    if (group) {
        const int tn0 = ceil_div(a.p.N, 128), tn1 = ceil_div(b.p.N, 128);
        const int t0 = tn0 * ceil_div(a.p.M, 64), t1 = tn1 * ceil_div(b.p.M, 64);
        const bool vec = use_vec(a.p, d0->a_kcontig, d0->b_kcontig);
        if (log_shapes())
            fprintf(stderr, "snag_gemm GROUP M=%d N=%d K=%d + M=%d N=%d K=%d a_kc=%d b_kc=%d tile=64x128\n", a.p.M, a.p.N, a.p.K, b.p.M, b.p.N, b.p.K,
                    (int)d0->a_kcontig, (int)d0->b_kcontig);
        GemmOperands pa = a.p, pb = b.p;
        pa.k_per_split = pa.K; pb.k_per_split = pb.K;
#define SNAG_GROUP(AK, BK_)                                                                                                                   \
    do {                                                                                                                                      \
        if (vec) hipLaunchKernelGGL((gemm_tile_group2_kernel<64, 128, 32, 64, AK, BK_, true, Epilogue>), dim3(t0 + t1), dim3(256), 0, st, pa, a.e, pb, b.e, t0, tn0, tn1);   \
        else hipLaunchKernelGGL((gemm_tile_group2_kernel<64, 128, 32, 64, AK, BK_, false, Epilogue>), dim3(t0 + t1), dim3(256), 0, st, pa, a.e, pb, b.e, t0, tn0, tn1);     \
    } while (0)
        if (d0->a_kcontig && d0->b_kcontig) SNAG_GROUP(true, true);
        else if (d0->a_kcontig && !d0->b_kcontig) SNAG_GROUP(true, false);
        else if (!d0->a_kcontig && d0->b_kcontig) SNAG_GROUP(false, true);
        else SNAG_GROUP(false, false);
#undef SNAG_GROUP
        return check_launch("snag_gemm_f32_pair(grouped)");
    }
    if (!same) {
        int rc = gemm_dispatch(d0, st);
        return rc ? rc : gemm_dispatch(d1, st);
    }
    const int splits = a.splits;
    float* s0 = splits > 1 ? d0->workspace : nullptr;
    float* s1 = splits > 1 ? d1->workspace : nullptr;
    const bool vec = use_vec(a.p, d0->a_kcontig, d0->b_kcontig);
    if (log_shapes())
        fprintf(stderr, "snag_gemm PAIR M=%d N=%d K=%d a_kc=%d b_kc=%d tile=%s splits=%d\n", a.p.M, a.p.N, a.p.K, (int)d0->a_kcontig, (int)d0->b_kcontig,
                a.big ? "128x128" : "64x128", splits);
    if (a.big) launch_cfg_pair<128, 128, 32, 128>(a.p, a.e, s0, b.p, b.e, s1, splits, d0->a_kcontig, d0->b_kcontig, vec, st);
    else launch_cfg_pair<64, 128, 32, 64>(a.p, a.e, s0, b.p, b.e, s1, splits, d0->a_kcontig, d0->b_kcontig, vec, st);
    int rc = check_launch("snag_gemm_f32_pair");
    if (rc) return rc;
    if (splits > 1 && !(d0->defer_reduce && d1->defer_reduce)) {
        const int64_t total = (int64_t)a.p.M * a.p.N;
        const bool quad = a.p.N % 4 == 0 && ((uintptr_t)s0 & 15) == 0 && ((uintptr_t)s1 & 15) == 0;
        const int blocks = (int)std::min<int64_t>(1024, (total / (quad ? 4 : 1) + 255) / 256);
        if (quad && reduce_small(total, splits))
            hipLaunchKernelGGL(splitk_reduce_small_pair_kernel, dim3((unsigned)((total / 4 + 63) / 64), 2), dim3(256), 0, st, (const float*)s0, a.e,
                               (const float*)s1, b.e, splits);
        else if (quad) hipLaunchKernelGGL(splitk_reduce_pair_kernel<true>, dim3(blocks, 2), dim3(256), 0, st, (const float*)s0, a.e, (const float*)s1, b.e, splits);
        else hipLaunchKernelGGL(splitk_reduce_pair_kernel<false>, dim3(blocks, 2), dim3(256), 0, st, (const float*)s0, a.e, (const float*)s1, b.e, splits);
        rc = check_launch("snag_gemm_f32_pair(split-K reduce)");
    }
    return rc;
}

static int launch_reduce(const GemmPlan& pl, float* slab, hipStream_t st);

// Up to kGroupMax problems in ONE launch (gemm_tile_group_kernel) when every one of them would run the 64 x 128 tile with plain or split K
// (no stream-K) on the same operand layouts; anything else: one after the other.  Each problem keeps the K split, slab and epilogue its
// own dispatch would use, so the results are the same bits as one-by-one launches.
#ifndef SNAG_GEMM_GROUP_BM
#define SNAG_GEMM_GROUP_BM 256      // rows per tile of the long-K grouped launches: 256 = eight waves in two ping-pong groups; 128 = four waves (A/B)
#endif
static int device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = -1;
    }
    return cus;
}

static int fill_reduce_set(const snag_gemm_desc* const* ds, int n, ReduceSet& rs, GemmPlan* last, float** last_slab);

int gemm_dispatch_group(const snag_gemm_desc* const* ds, int n, hipStream_t st, const GroupTail* tail) {
    GemmPlan pl[kGroupMax];
    bool ok = n >= 2 && n <= kGroupMax;
    bool pp = false;                  // every problem on the eight-wave 256 x 128 tile instead (512 threads per workgroup)
    bool mixed = false;               // the A operands' layouts differ (64 x 128 tiles only): gemm_tile_group_mixed_kernel
    for (int i = 0; ok && i < n; ++i) {
        const snag_gemm_desc* d = ds[i];
        pl[i] = plan_gemm(d);
        if (i == 0) pp = pl[0].big && pl[0].pp;
        // (in a group of eight-wave tiles a problem with an explicit split request rides along even if, alone, its row count would pick
        // the 64 x 128 tile: its K split and epilogue do not depend on the tile)
        const bool pp_ok = (pl[i].big && pl[i].pp) || (d->splits > 0 && kBf16x != 0 && d->K >= 512 && d->N <= 128 && !pl[i].big);
        mixed = mixed || d->a_kcontig != ds[0]->a_kcontig;
        ok = d->M > 0 && d->N > 0 && d->K > 0 && !pl[i].streamk && !pl[i].wide && (pp ? pp_ok : (!pl[i].big && !pl[i].pp)) &&
             (d->a_kcontig == ds[0]->a_kcontig || !pp) && d->b_kcontig == ds[0]->b_kcontig &&
             use_vec(pl[i].p, d->a_kcontig, d->b_kcontig) == use_vec(pl[0].p, ds[0]->a_kcontig, ds[0]->b_kcontig);
        for (int j = 0; ok && j < i; ++j) ok = !(pl[i].splits > 1 && pl[j].splits > 1 && ds[i]->workspace == ds[j]->workspace);
    }
    // the tail rides only in ONE launch of eight-wave tiles that leaves CUs idle (see below); everywhere else it is the reduce launch it would have been
    auto tail_alone = [&]() { return tail ? gemm_reduce_multi(tail->reduce, tail->n_reduce, st, nullptr, 0, tail->colsum) : SNAG_OK; };
    if (!ok) {
        int rc = tail_alone();
        for (int i = 0; !rc && i < n; ++i) rc = gemm_dispatch(ds[i], st);
        return rc;
    }
    GemmGroup<Epilogue> g{};
    g.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = pl[i].p;
        g.p[i].xcd = 0;
        g.e[i] = pl[i].e;
        g.slab[i] = pl[i].splits > 1 ? ds[i]->workspace : nullptr;
        g.tiles_n[i] = ceil_div(pl[i].p.N, 128);
        g.tiles[i] = g.tiles_n[i] * ceil_div(pl[i].p.M, pp ? SNAG_GEMM_GROUP_BM : 64);
        g.first[i] = blocks;
        blocks += g.tiles[i] * pl[i].splits;
        if (ds[i]->a_kcontig) g.a_kc_mask |= 1u << i;
        // the weight-gradient shapes (dW = dY^T X: both operands row-contiguous, K = the entity rows) at the FRONT of the launch get the XCD-aware
        // block order (see gemm_tile_group_kernel); it stops at the first problem of another shape
        if (!pp && !ds[i]->a_kcontig && !ds[i]->b_kcontig && g.xcd == g.first[i]) g.xcd = blocks;
        if (log_shapes())
            fprintf(stderr, "snag_gemm GROUP[%d/%d] M=%d N=%d K=%d a_kc=%d b_kc=%d tile=%s splits=%d\n", i, n, pl[i].p.M, pl[i].p.N, pl[i].p.K, (int)ds[i]->a_kcontig,
                    (int)ds[i]->b_kcontig, pp ? "256x128" : "64x128", pl[i].splits);
    }
    g.first[n] = blocks;
    const bool vec = use_vec(pl[0].p, ds[0]->a_kcontig, ds[0]->b_kcontig);
    if (tail != nullptr) {
        // idle CUs of this launch (one 512-thread workgroup of the eight-wave kernel per CU: 176 registers per lane)
        const int idle = (pp && SNAG_GEMM_GROUP_BM == 256 && !mixed) ? device_cu_count() - blocks : 0;
        TailSet t{};
        t.n_reduce = idle >= 4 ? fill_reduce_set(tail->reduce, tail->n_reduce, t.rs, nullptr, nullptr) : 0;
        if (t.n_reduce < 0) return t.n_reduce;
        if (idle < 4) {
            const int rc = tail_alone();
            if (rc) return rc;
        } else {
            if (tail->colsum) { t.cs = *tail->colsum; t.cs_blocks = 2 * ceil_div(3 * tail->colsum->D, 16); }
            const int riders = std::min(idle, 16);
#define SNAG_TAILN(AK, BK_)                                                                                                                       \
    do {                                                                                                                                          \
        if (vec) hipLaunchKernelGGL((gemm_tile_group_tail_kernel<AK, BK_, true>), dim3(blocks + riders), dim3(512), 0, st, g, blocks, t);          \
        else hipLaunchKernelGGL((gemm_tile_group_tail_kernel<AK, BK_, false>), dim3(blocks + riders), dim3(512), 0, st, g, blocks, t);            \
    } while (0)
            if (ds[0]->a_kcontig && ds[0]->b_kcontig) SNAG_TAILN(true, true);
            else if (ds[0]->a_kcontig && !ds[0]->b_kcontig) SNAG_TAILN(true, false);
            else if (!ds[0]->a_kcontig && ds[0]->b_kcontig) SNAG_TAILN(false, true);
            else SNAG_TAILN(false, false);
#undef SNAG_TAILN
            int rc = check_launch("snag_gemm_f32_group (+ tail)");
            for (int i = 0; !rc && i < n; ++i)
                if (pl[i].splits > 1 && !ds[i]->defer_reduce) rc = launch_reduce(pl[i], ds[i]->workspace, st);
            return rc;
        }
    }
    if (mixed) {
        if (ds[0]->b_kcontig) {
            if (vec) hipLaunchKernelGGL((gemm_tile_group_mixed_kernel<64, 128, 32, 64, true, true, Epilogue>), dim3(blocks), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((gemm_tile_group_mixed_kernel<64, 128, 32, 64, true, false, Epilogue>), dim3(blocks), dim3(256), 0, st, g);
        } else {
            if (vec) hipLaunchKernelGGL((gemm_tile_group_mixed_kernel<64, 128, 32, 64, false, true, Epilogue>), dim3(blocks), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((gemm_tile_group_mixed_kernel<64, 128, 32, 64, false, false, Epilogue>), dim3(blocks), dim3(256), 0, st, g);
        }
    } else {
#define SNAG_GROUPN(AK, BK_)                                                                                                              \
    do {                                                                                                                                  \
        if (pp && vec) hipLaunchKernelGGL((gemm_tile_group_kernel<SNAG_GEMM_GROUP_BM, 128, 32, 128, AK, BK_, true, Epilogue>), dim3(blocks), dim3(2 * SNAG_GEMM_GROUP_BM), 0, st, g);     \
        else if (pp) hipLaunchKernelGGL((gemm_tile_group_kernel<SNAG_GEMM_GROUP_BM, 128, 32, 128, AK, BK_, false, Epilogue>), dim3(blocks), dim3(2 * SNAG_GEMM_GROUP_BM), 0, st, g);      \
        else if (vec) hipLaunchKernelGGL((gemm_tile_group_kernel<64, 128, 32, 64, AK, BK_, true, Epilogue>), dim3(blocks), dim3(256), 0, st, g);     \
        else hipLaunchKernelGGL((gemm_tile_group_kernel<64, 128, 32, 64, AK, BK_, false, Epilogue>), dim3(blocks), dim3(256), 0, st, g);       \
    } while (0)
        if (ds[0]->a_kcontig && ds[0]->b_kcontig) SNAG_GROUPN(true, true);
        else if (ds[0]->a_kcontig && !ds[0]->b_kcontig) SNAG_GROUPN(true, false);
        else if (!ds[0]->a_kcontig && ds[0]->b_kcontig) SNAG_GROUPN(false, true);
        else SNAG_GROUPN(false, false);
#undef SNAG_GROUPN
    }
    int rc = check_launch("snag_gemm_f32_group");
    for (int i = 0; !rc && i < n; ++i)
        if (pl[i].splits > 1 && !ds[i]->defer_reduce) rc = launch_reduce(pl[i], ds[i]->workspace, st);
    return rc;
}

// the reduce launch of one problem (what gemm_dispatch appends to a K split)
static int launch_reduce(const GemmPlan& pl, float* slab, hipStream_t st) {
    const GemmOperands& p = pl.p;
    const int64_t total = (int64_t)p.M * p.N;
    const bool quad = p.N % 4 == 0 && ((uintptr_t)slab & 15) == 0;
    const int blocks = (int)std::min<int64_t>(2048, (total / (quad ? 4 : 1) + 255) / 256);
    if (quad && reduce_small(total, pl.splits))
        hipLaunchKernelGGL(splitk_reduce_small_kernel, dim3((unsigned)((total / 4 + 63) / 64)), dim3(256), 0, st, (const float*)slab, pl.splits, pl.e);
    else if (quad) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, (const float*)slab, pl.splits, pl.e);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, (const float*)slab, pl.splits, pl.e);
    return check_launch("snag_gemm_f32(split-K reduce)");
}

// the problems among ds[0..n) that left K-split slabs behind (defer_reduce), as a ReduceSet; returns their number or a negative error
static int fill_reduce_set(const snag_gemm_desc* const* ds, int n, ReduceSet& rs, GemmPlan* last = nullptr, float** last_slab = nullptr) {
    int cnt = 0;
    for (int i = 0; i < n && i < 4; ++i) {
        const snag_gemm_desc* d = ds[i];
        const GemmPlan pl = plan_gemm(d);
        if (!(d->defer_reduce && !pl.streamk && pl.splits > 1 && d->M > 0 && d->N > 0)) continue;      // finished by its own launch
        for (int j = 0; j < cnt; ++j) SNAG_REQUIRE(rs.slab[j] != d->workspace, SNAG_ERR_ARG, "gemm_reduce: two problems share a workspace");
        const int64_t total = (int64_t)pl.p.M * pl.p.N;
        const bool quad = pl.p.N % 4 == 0 && ((uintptr_t)d->workspace & 15) == 0;
        rs.slab[cnt] = d->workspace;
        rs.splits[cnt] = pl.splits;
        rs.mode[cnt] = quad ? (reduce_small(total, pl.splits) ? 2 : 1) : 0;
        rs.nblk[cnt] = rs.mode[cnt] == 2 ? (int)((total / 4 + 63) / 64) : (int)std::min<int64_t>(2048, (total / (quad ? 4 : 1) + 255) / 256);
        rs.e[cnt] = pl.e;
        if (last) *last = pl;
        if (last_slab) *last_slab = d->workspace;
        ++cnt;
    }
    return cnt;
}

// finishes up to four problems launched with defer_reduce (distinct workspaces): ONE launch for all that left slabs behind
int gemm_reduce_multi(const snag_gemm_desc* const* ds, int n, hipStream_t st, const snag_small_mm* small, int n_small, const SlabColsum* colsum,
                      const RoWeightImages* images) {
    const int cs_blocks = colsum ? 2 * ceil_div(3 * colsum->D, 16) : 0;
    const SlabColsum cs = colsum ? *colsum : SlabColsum{};
    const int wi_blocks = images ? images->n_images : 0;
    const RoWeightImages wi = images ? *images : RoWeightImages{};
    SmallMMSet mm{};
    int mm_blocks = 0;
    if (n_small > 0) {
        mm_blocks = fill_small_mm(small, n_small, mm, "gemm_reduce_group_mm");
        if (mm_blocks < 0) return mm_blocks;
    }
    ReduceSet rs{};
    GemmPlan last{};
    float* last_slab = nullptr;
    const int cnt = fill_reduce_set(ds, n, rs, &last, &last_slab);
    if (cnt < 0) return cnt;
    if (cnt == 0 && mm_blocks == 0 && cs_blocks == 0 && wi_blocks == 0) return SNAG_OK;
    if (cnt == 1 && mm_blocks == 0 && cs_blocks == 0 && wi_blocks == 0) return launch_reduce(last, last_slab, st);
    if (cnt == 1 && mm_blocks == 0 && cs_blocks == 0 && rs.mode[0] != 2) {          // one problem + the weight images
        if (rs.mode[0] == 1) hipLaunchKernelGGL(splitk_reduce_images_kernel<true>, dim3(rs.nblk[0] + wi_blocks), dim3(256), 0, st, rs.slab[0], rs.splits[0], rs.e[0], wi);
        else hipLaunchKernelGGL(splitk_reduce_images_kernel<false>, dim3(rs.nblk[0] + wi_blocks), dim3(256), 0, st, rs.slab[0], rs.splits[0], rs.e[0], wi);
        return check_launch("snag_gemm_reduce_group");
    }
    int total_blocks = mm_blocks + cs_blocks + wi_blocks;
    for (int i = 0; i < cnt; ++i) total_blocks += rs.nblk[i];
    hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, rs, cnt, mm, mm_blocks, cs, cs_blocks, wi, wi_blocks);
    return check_launch("snag_gemm_reduce_group");
}

int gemm_reduce_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, hipStream_t st) {
    const snag_gemm_desc* ds[2] = {d0, d1};
    return gemm_reduce_multi(ds, 2, st);
}

int gemm_check(const snag_gemm_desc* d) {
    SNAG_REQUIRE(d != nullptr, SNAG_ERR_ARG, "gemm: null descriptor");
    SNAG_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, SNAG_ERR_ARG, "gemm: negative size M=%d N=%d K=%d", d->M, d->N, d->K);
    if (d->M == 0 || d->N == 0) return SNAG_OK;
    SNAG_REQUIRE(d->C != nullptr, SNAG_ERR_ARG, "gemm: null C");
    SNAG_REQUIRE(d->K == 0 || (d->A && d->B), SNAG_ERR_ARG, "gemm: null A or B");
    SNAG_REQUIRE(d->ldc >= (d->c_transposed ? d->M : d->N), SNAG_ERR_ARG, "gemm: ldc=%ld < %s=%d", (long)d->ldc, d->c_transposed ? "M" : "N", d->c_transposed ? d->M : d->N);
    SNAG_REQUIRE(!d->c_transposed || (!d->pre_act && !d->mirror), SNAG_ERR_UNSUPPORTED, "gemm: a transposed C takes no pre_act / mirror copy");
    SNAG_REQUIRE(d->lda >= (d->a_kcontig ? d->K : d->M), SNAG_ERR_ARG, "gemm: lda=%ld too small", (long)d->lda);
    SNAG_REQUIRE(d->ldb >= (d->b_kcontig ? d->K : d->N), SNAG_ERR_ARG, "gemm: ldb=%ld too small", (long)d->ldb);
    // the staged loaders address one operand tile with 32-bit byte offsets from a 64-bit base: 128 rows (or 16 k rows) x pitch < 4 GB
    SNAG_REQUIRE(d->lda < (1 << 22) && d->ldb < (1 << 22), SNAG_ERR_UNSUPPORTED, "gemm: operand pitch lda=%ld / ldb=%ld above 4M elements", (long)d->lda,
                 (long)d->ldb);
    SNAG_REQUIRE(d->act >= 0 && d->act <= 4, SNAG_ERR_ARG, "gemm: act=%d unknown", d->act);
    SNAG_REQUIRE(d->act < 3 || (d->aux && d->ldaux >= d->N), SNAG_ERR_ARG, "gemm: act=%d needs aux [M,N]", d->act);
    SNAG_REQUIRE(!d->add1 || d->ld1 >= d->N, SNAG_ERR_ARG, "gemm: ld1 too small");
    SNAG_REQUIRE(!d->add2 || d->ld2 >= d->N, SNAG_ERR_ARG, "gemm: ld2 too small");
    SNAG_REQUIRE(d->ldmask == 0 || d->ldmask >= d->N, SNAG_ERR_ARG, "gemm: ldmask=%ld < N=%d", (long)d->ldmask, d->N);
    SNAG_REQUIRE(!d->mirror || d->ldmirror >= d->N, SNAG_ERR_ARG, "gemm: ldmirror=%ld < N=%d", (long)d->ldmirror, d->N);
    return SNAG_OK;
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_gemm_f32(const snag_gemm_desc* d, void* stream) {
    int rc = gemm_check(d);
    if (rc) return rc;
    if (d->M == 0 || d->N == 0) return SNAG_OK;
    return gemm_dispatch(d, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_gemm_f32_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, void* stream) {
    int rc = gemm_check(d0);
    if (rc) return rc;
    if ((rc = gemm_check(d1))) return rc;
    const bool e0 = d0->M == 0 || d0->N == 0, e1 = d1->M == 0 || d1->N == 0;
    if (e0 || e1) {
        if (!e0) return gemm_dispatch(d0, (hipStream_t)stream);
        if (!e1) return gemm_dispatch(d1, (hipStream_t)stream);
        return SNAG_OK;
    }
    return gemm_dispatch_pair(d0, d1, (hipStream_t)stream);
}

namespace snag {
// snag_gemm_f32_group's body (checked descriptors, empty problems dropped) with an optional tail (common.hip.h: GroupTail)
int gemm_group_checked(const snag_gemm_desc* const* descs, int n, hipStream_t st, const GroupTail* tail) {
    SNAG_REQUIRE(n >= 0 && n <= kGroupMax && (n == 0 || descs), SNAG_ERR_ARG, "gemm_f32_group: n=%d (0..%d problems per call)", n, kGroupMax);
    const snag_gemm_desc* live[kGroupMax];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        SNAG_REQUIRE(descs[i], SNAG_ERR_ARG, "gemm_f32_group: null descriptor %d", i);
        int rc = gemm_check(descs[i]);
        if (rc) return rc;
        if (descs[i]->M > 0 && descs[i]->N > 0) live[m++] = descs[i];
    }
    if (m >= 2) return gemm_dispatch_group(live, m, st, tail);
    if (tail) {          // nothing to ride in: the reduce launch it would have been
        const int rc = gemm_reduce_multi(tail->reduce, tail->n_reduce, st, nullptr, 0, tail->colsum);
        if (rc) return rc;
    }
    return m == 1 ? gemm_dispatch(live[0], st) : SNAG_OK;
}
}  // namespace snag

extern "C" SNAG_API int snag_gemm_f32_group(const snag_gemm_desc* const* descs, int n, void* stream) {
    return gemm_group_checked(descs, n, (hipStream_t)stream, nullptr);
}

extern "C" SNAG_API int snag_gemm_reduce_pair(const snag_gemm_desc* d0, const snag_gemm_desc* d1, void* stream) {
    int rc = gemm_check(d0);
    if (rc) return rc;
    if ((rc = gemm_check(d1))) return rc;
    return gemm_reduce_pair(d0, d1, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_gemm_reduce_group(const snag_gemm_desc* const* descs, int n, void* stream) {
    SNAG_REQUIRE(n >= 0 && n <= 4 && (n == 0 || descs), SNAG_ERR_ARG, "gemm_reduce_group: n=%d (0..4 problems per launch)", n);
    for (int i = 0; i < n; ++i) {
        SNAG_REQUIRE(descs[i], SNAG_ERR_ARG, "gemm_reduce_group: null descriptor %d", i);
        int rc = gemm_check(descs[i]);
        if (rc) return rc;
    }
    return gemm_reduce_multi(descs, n, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_gemm_reduce_group_mm(const snag_gemm_desc* const* descs, int n, const snag_small_mm* small_products, int n_small, void* stream) {
    SNAG_REQUIRE(n >= 0 && n <= 4 && (n == 0 || descs), SNAG_ERR_ARG, "gemm_reduce_group_mm: n=%d (0..4 problems per launch)", n);
    SNAG_REQUIRE(n_small >= 0 && n_small <= 4 && (n_small == 0 || small_products), SNAG_ERR_ARG, "gemm_reduce_group_mm: n_small=%d (0..4 small products)", n_small);
    for (int i = 0; i < n; ++i) {
        SNAG_REQUIRE(descs[i], SNAG_ERR_ARG, "gemm_reduce_group_mm: null descriptor %d", i);
        int rc = gemm_check(descs[i]);
        if (rc) return rc;
    }
    return gemm_reduce_multi(descs, n, (hipStream_t)stream, small_products, n_small);
}

extern "C" SNAG_API int snag_gram_f32(const float* Z, int64_t ldz, int n, int k, float alpha, float* S, int64_t lds, void* stream) {
    SNAG_REQUIRE(n >= 0 && k >= 0 && ldz >= k && lds >= n, SNAG_ERR_ARG, "gram: bad sizes n=%d k=%d ldz=%ld lds=%ld", n, k, (long)ldz, (long)lds);
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE((Z || k == 0) && S, SNAG_ERR_ARG, "gram: null pointer");
    GemmOperands p;
    p.A = Z; p.B = Z; p.lda = ldz; p.ldb = ldz; p.M = n; p.N = n; p.K = k;
    p.a_vec = p.b_vec = (ldz % 4 == 0) && (((uintptr_t)Z & 15) == 0);
    p.k_per_split = ceil_div(std::max(k, 1), kBK) * kBK;
    p.dbg = 0;
    const int t = ceil_div(n, 128);
    p.xcd = 1;
    hipLaunchKernelGGL(gram_tile_kernel, dim3(t * (t + 1) / 2), dim3(256), 0, (hipStream_t)stream, p, alpha, S, lds, t);
    return check_launch("snag_gram_f32");
}

extern "C" SNAG_API int64_t snag_gemm_workspace_floats(int M, int N, int K) { return gemm_workspace_floats(M, N, K); }

