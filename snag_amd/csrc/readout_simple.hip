// a9 joint_way "atten_weight" (README.md:51) and the README's POOL=1 knob (README.md:68,79): the two row-wise, HBM-bound pieces of the
// MKGC read-out family that need no transformer.
//
// atten_weight -- non-transformer attention over an entity's modality tokens.  The only arithmetic of that name in the reference tree is
// EVA_tools.SelfAttention.forward (model/EVA_tools.py:190-232, dead code there): one learnable vector a[D],
//     pre_m = x_m . a ;  w = softmax_m(relu(pre)) ;  out = sum_m w_m x_m .
// PARITY UNPINNED beyond that formula (the MKGC source is absent, SURVEY.md par. 0.1).
//
// Sixteen lanes own one entity (a wave = four entities): every token row is read as float4 by lane l at quads l, l + 16, ...; the M dot
// products meet by DPP row rotations (sum16), the weighted sum re-reads the rows from the L1 they were just pulled through.
#include "common.hip.h"

namespace snag {

constexpr int kAttenBlocks = 512;      // partial slabs of d a (fixed count: a fixed summation order whatever the grid)

template <int M>
__global__ __launch_bounds__(256) void atten_weight_fwd_kernel(const float* __restrict__ x, int64_t rows, int D, const float* __restrict__ a,
                                                               float* __restrict__ out, int64_t ld_out, float* __restrict__ aw) {
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int nq = D >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    for (int64_t n0 = (int64_t)blockIdx.x * 16; n0 < rows; n0 += (int64_t)gridDim.x * 16) {
        const int64_t n = n0 + g;
        const bool live = n < rows;
        const float4* xr = reinterpret_cast<const float4*>(x + (live ? n : rows - 1) * (int64_t)M * D);
        float dot[M];
#pragma unroll
        for (int m = 0; m < M; ++m) dot[m] = 0.f;
        for (int q = l; q < nq; q += 16) {
            const float4 av = a4[q];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float4 v = xr[m * nq + q];
                dot[m] = fmaf(v.x, av.x, dot[m]);
                dot[m] = fmaf(v.y, av.y, dot[m]);
                dot[m] = fmaf(v.z, av.z, dot[m]);
                dot[m] = fmaf(v.w, av.w, dot[m]);
            }
        }
        float w[M], mx = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            dot[m] = sum16(dot[m]);
            w[m] = fmaxf(dot[m], 0.f);
            mx = m == 0 ? w[m] : fmaxf(mx, w[m]);
        }
        float den = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            w[m] = expf(w[m] - mx);
            den += w[m];
        }
#pragma unroll
        for (int m = 0; m < M; ++m) w[m] = w[m] / den;
        if (live) {
            float4* o4 = reinterpret_cast<float4*>(out + n * ld_out);
            for (int q = l; q < nq; q += 16) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float4 v = xr[m * nq + q];
                    acc.x = fmaf(w[m], v.x, acc.x);
                    acc.y = fmaf(w[m], v.y, acc.y);
                    acc.z = fmaf(w[m], v.z, acc.z);
                    acc.w = fmaf(w[m], v.w, acc.w);
                }
                o4[q] = acc;
            }
            if (l == 0) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    aw[n * 2 * M + m] = w[m];
                    aw[n * 2 * M + M + m] = dot[m];
                }
            }
        }
    }
}

// Backward of the above.  With dw_m = d_out . x_m and t = sum_j w_j dw_j:
//   d pre_m = [pre_m > 0] w_m (dw_m - t) ;  d x_m = w_m d_out + d pre_m a ;  d a = sum_n sum_m d pre_m x_m .
// d a: every thread keeps the columns it owns in registers over its entities, the block's sixteen entity groups meet in LDS in group order,
// block b writes slab b; atten_da_reduce_kernel adds the kAttenBlocks slabs in order (no float atomics: bit-reproducible).
template <int M, int NQ>
__global__ __launch_bounds__(256) void atten_weight_bwd_kernel(const float* __restrict__ x, int64_t rows, int D, const float* __restrict__ a,
                                                               const float* __restrict__ aw, const float* __restrict__ d_out, int64_t ld_dout,
                                                               float* __restrict__ d_x, float* __restrict__ part) {
    __shared__ float red[16][NQ * 64 + 4];
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int nq = D >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    float4 da[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) da[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t n0 = (int64_t)blockIdx.x * 16; n0 < rows; n0 += (int64_t)gridDim.x * 16) {
        const int64_t n = n0 + g;
        const bool live = n < rows;
        const int64_t nc = live ? n : rows - 1;
        const float4* xr = reinterpret_cast<const float4*>(x + nc * (int64_t)M * D);
        const float4* go = reinterpret_cast<const float4*>(d_out + nc * ld_dout);
        float dw[M], w[M], pre[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            dw[m] = 0.f;
            w[m] = aw[nc * 2 * M + m];
            pre[m] = aw[nc * 2 * M + M + m];
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = l + 16 * u;
            if (q < nq) {
                const float4 gv = go[q];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float4 v = xr[m * nq + q];
                    dw[m] = fmaf(v.x, gv.x, dw[m]);
                    dw[m] = fmaf(v.y, gv.y, dw[m]);
                    dw[m] = fmaf(v.z, gv.z, dw[m]);
                    dw[m] = fmaf(v.w, gv.w, dw[m]);
                }
            }
        }
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            dw[m] = sum16(dw[m]);
            t = fmaf(w[m], dw[m], t);
        }
        float dpre[M];
#pragma unroll
        for (int m = 0; m < M; ++m) dpre[m] = (live && pre[m] > 0.f) ? w[m] * (dw[m] - t) : 0.f;
        float4* dxr = reinterpret_cast<float4*>(d_x + nc * (int64_t)M * D);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = l + 16 * u;
            if (q < nq) {
                const float4 gv = go[q];
                const float4 av = a4[q];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float4 v = xr[m * nq + q];
                    da[u].x = fmaf(dpre[m], v.x, da[u].x);
                    da[u].y = fmaf(dpre[m], v.y, da[u].y);
                    da[u].z = fmaf(dpre[m], v.z, da[u].z);
                    da[u].w = fmaf(dpre[m], v.w, da[u].w);
                    if (live) {
                        float4 o;
                        o.x = fmaf(w[m], gv.x, dpre[m] * av.x);
                        o.y = fmaf(w[m], gv.y, dpre[m] * av.y);
                        o.z = fmaf(w[m], gv.z, dpre[m] * av.z);
                        o.w = fmaf(w[m], gv.w, dpre[m] * av.w);
                        dxr[m * nq + q] = o;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int c = 4 * (l + 16 * u);
        red[g][c] = da[u].x; red[g][c + 1] = da[u].y; red[g][c + 2] = da[u].z; red[g][c + 3] = da[u].w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][c];
        part[(int64_t)blockIdx.x * D + c] = s;
    }
}

__global__ __launch_bounds__(256) void atten_da_reduce_kernel(const float* __restrict__ part, int nblocks, int D, float* __restrict__ d_a, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(int64_t)b * D + c];
    d_a[c] = accumulate ? d_a[c] + s : s;
}

// README POOL=1 ("pooling operations are applied to all pre-extracted visual/text features for dimensionality reduction and uniformity
// in dimensions", README.md:68,79): every row of a feature table, viewed as an H x W image, is adaptively average-pooled to oh x ow --
// window [floor(i H / oh), ceil((i + 1) H / oh)) x [floor(j W / ow), ceil((j + 1) W / ow)), sum in row-major order divided by the window
// size (the definition of torch.nn.functional.adaptive_avg_pool2d; H = oh = 1 is the 1-D pool).  Once per table at load time.
__global__ __launch_bounds__(256) void adaptive_avg_pool_rows_kernel(const float* __restrict__ x, int64_t N, int64_t ld, int H, int W, int oh, int ow,
                                                                     float* __restrict__ out, int64_t ld_out) {
    const int64_t per = (int64_t)oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * per; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / per;
        const int o = (int)(i - n * per);
        const int oi = o / ow, oj = o - oi * ow;
        const int h0 = (int)(((int64_t)oi * H) / oh), h1 = (int)((((int64_t)oi + 1) * H + oh - 1) / oh);
        const int w0 = (int)(((int64_t)oj * W) / ow), w1 = (int)((((int64_t)oj + 1) * W + ow - 1) / ow);
        const float* row = x + n * ld;
        float s = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) s += row[(int64_t)h * W + w];
        out[n * ld_out + o] = s / (float)((h1 - h0) * (w1 - w0));
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int grid_rows16(int64_t rows, int cap) {
    const int64_t b = (rows + 15) / 16;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int64_t snag_atten_weight_work_floats(int D) { return (int64_t)kAttenBlocks * (D > 0 ? D : 0); }

extern "C" SNAG_API int snag_atten_weight_fwd(const float* x, int64_t rows, int M, int D, const float* a, float* out, int64_t ld_out, float* aw,
                                              void* stream) {
    SNAG_REQUIRE(rows >= 0 && ld_out >= D, SNAG_ERR_ARG, "atten_weight_fwd: bad sizes");
    SNAG_REQUIRE(M >= 1 && M <= 6 && D >= 4 && D <= 512 && D % 4 == 0, SNAG_ERR_UNSUPPORTED, "atten_weight_fwd: 1 <= M <= 6 tokens of 4 | D <= 512 (got M=%d D=%d)", M, D);
    if (rows == 0) return SNAG_OK;          // (an empty tensor's data pointer is null)
    SNAG_REQUIRE(x && a && out && aw, SNAG_ERR_ARG, "atten_weight_fwd: null pointer");
    SNAG_REQUIRE(ld_out % 4 == 0 && aligned16(x) && aligned16(a) && aligned16(out), SNAG_ERR_UNSUPPORTED,
                 "atten_weight_fwd: x, a, out must be 16-byte aligned and ld_out a multiple of 4 floats (rows are read and written as float4)");
    const dim3 grid(grid_rows16(rows, 1 << 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SNAG_AW_FWD(MM) case MM: hipLaunchKernelGGL(atten_weight_fwd_kernel<MM>, grid, block, 0, st, x, rows, D, a, out, ld_out, aw); break;
    switch (M) { SNAG_AW_FWD(1) SNAG_AW_FWD(2) SNAG_AW_FWD(3) SNAG_AW_FWD(4) SNAG_AW_FWD(5) SNAG_AW_FWD(6) }
#undef SNAG_AW_FWD
    return check_launch("snag_atten_weight_fwd");
}

extern "C" SNAG_API int snag_atten_weight_bwd(const float* x, int64_t rows, int M, int D, const float* a, const float* aw, const float* d_out,
                                              int64_t ld_dout, float* d_x, float* d_a, int accumulate_da, float* work, void* stream) {
    SNAG_REQUIRE(rows >= 0 && ld_dout >= D, SNAG_ERR_ARG, "atten_weight_bwd: bad sizes");
    SNAG_REQUIRE(M >= 1 && M <= 6 && D >= 4 && D <= 512 && D % 4 == 0, SNAG_ERR_UNSUPPORTED, "atten_weight_bwd: 1 <= M <= 6 tokens of 4 | D <= 512 (got M=%d D=%d)", M, D);
    SNAG_REQUIRE(d_a != nullptr, SNAG_ERR_ARG, "atten_weight_bwd: null d_a");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {          // (an empty tensor's data pointer is null) no entity: d a = 0
        if (!accumulate_da) (void)hipMemsetAsync(d_a, 0, sizeof(float) * D, st);
        return check_launch("snag_atten_weight_bwd");
    }
    SNAG_REQUIRE(x && a && aw && d_out && d_x && work, SNAG_ERR_ARG, "atten_weight_bwd: null pointer");
    SNAG_REQUIRE(ld_dout % 4 == 0 && aligned16(x) && aligned16(a) && aligned16(d_out) && aligned16(d_x), SNAG_ERR_UNSUPPORTED,
                 "atten_weight_bwd: x, a, d_out, d_x must be 16-byte aligned and ld_dout a multiple of 4 floats (rows are read and written as float4)");
    const int nb = grid_rows16(rows, kAttenBlocks);
    const dim3 grid(nb), block(256);
#define SNAG_AW_BWD(MM)                                                                                                                      \
    case MM:                                                                                                                                 \
        if (D <= 128) hipLaunchKernelGGL((atten_weight_bwd_kernel<MM, 2>), grid, block, 0, st, x, rows, D, a, aw, d_out, ld_dout, d_x, work); \
        else hipLaunchKernelGGL((atten_weight_bwd_kernel<MM, 8>), grid, block, 0, st, x, rows, D, a, aw, d_out, ld_dout, d_x, work);          \
        break;
    switch (M) { SNAG_AW_BWD(1) SNAG_AW_BWD(2) SNAG_AW_BWD(3) SNAG_AW_BWD(4) SNAG_AW_BWD(5) SNAG_AW_BWD(6) }
#undef SNAG_AW_BWD
    hipLaunchKernelGGL(atten_da_reduce_kernel, dim3((D + 255) / 256), block, 0, st, work, nb, D, d_a, accumulate_da);
    return check_launch("snag_atten_weight_bwd");
}

extern "C" SNAG_API int snag_adaptive_avg_pool_rows(const float* x, int64_t N, int64_t ld, int H, int W, int oh, int ow, float* out, int64_t ld_out,
                                                    void* stream) {
    SNAG_REQUIRE(x && out, SNAG_ERR_ARG, "adaptive_avg_pool_rows: null pointer");
    SNAG_REQUIRE(N >= 0 && H >= 1 && W >= 1 && oh >= 1 && ow >= 1 && ld >= (int64_t)H * W && ld_out >= (int64_t)oh * ow, SNAG_ERR_ARG,
                 "adaptive_avg_pool_rows: bad sizes (H=%d W=%d -> %d x %d)", H, W, oh, ow);
    if (N == 0) return SNAG_OK;
    const int64_t total = N * (int64_t)oh * ow;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(adaptive_avg_pool_rows_kernel, dim3((unsigned)(blocks > (1 << 20) ? (1 << 20) : blocks)), dim3(256), 0, (hipStream_t)stream, x, N, ld, H, W, oh,
                       ow, out, ld_out);
    return check_launch("snag_adaptive_avg_pool_rows");
}
