// BASELINE config 5 (|E| = 1M, dim 256, 128 negatives, bf16 tables, 8 GPUs): what the training step needs besides the bf16 scorer.
// The entity table is stored in bf16 (what the scorer gathers from, replicated); every rank OWNS 1/G of the rows as fp32 master
// weights with their Adam moments (SURVEY.md par. 8e option 2: row-sharded ownership + reduce-scatter of the gradient).
#include <algorithm>

#include "common.hip.h"

namespace snag {

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float x) {
    const __bf16 b = (__bf16)x;            // round-to-nearest-even (v_cvt_pk_bf16_f32); a NaN stays a NaN
    return __builtin_bit_cast(uint16_t, b);
}

// torch.optim.AdamW on a slice of fp32 master weights, four elements per lane, writing the bf16 copy of the updated weights
// next to it (the slice of the replicated bf16 table this rank owns) in the same pass.
__global__ __launch_bounds__(256) void adamw_bf16_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps, float wd,
                                                         float bc1, float bc2, float gscale, const uint64_t* __restrict__ tick,
                                                         uint16_t* __restrict__ bf16_out) {
    if (tick != nullptr) {
        const float t = (float)(tick[0] + 1);
        bc1 = 1.0f - powf(b1, t);
        bc2 = 1.0f - powf(b2, t);
    }
    const float isb2 = 1.0f / sqrtf(bc2), step = lr / bc1, decay = 1.0f - lr * wd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 gi = reinterpret_cast<const float4*>(g)[i];
        float4 pi = reinterpret_cast<const float4*>(p)[i], mi = reinterpret_cast<const float4*>(m)[i], vi = reinterpret_cast<const float4*>(v)[i];
        float* pp = &pi.x; float* mm = &mi.x; float* vv = &vi.x; const float* gg = &gi.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * gscale;
            mm[e] = b1 * mm[e] + (1.0f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.0f - b2) * ge * ge;
            pp[e] = pp[e] * decay - step * (mm[e] / (sqrtf(vv[e]) * isb2 + eps));
        }
        reinterpret_cast<float4*>(p)[i] = pi;
        reinterpret_cast<float4*>(m)[i] = mi;
        reinterpret_cast<float4*>(v)[i] = vi;
        if (bf16_out != nullptr) {
            ushort4 o;
            o.x = f32_to_bf16_bits(pi.x); o.y = f32_to_bf16_bits(pi.y); o.z = f32_to_bf16_bits(pi.z); o.w = f32_to_bf16_bits(pi.w);
            reinterpret_cast<ushort4*>(bf16_out)[i] = o;
        }
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, int64_t n4, uint16_t* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = reinterpret_cast<const float4*>(src)[i];
        ushort4 o;
        o.x = f32_to_bf16_bits(x.x); o.y = f32_to_bf16_bits(x.y); o.z = f32_to_bf16_bits(x.z); o.w = f32_to_bf16_bits(x.w);
        reinterpret_cast<ushort4*>(dst)[i] = o;
    }
}

static int grid4(int64_t n4) { return (int)std::max<int64_t>(1, std::min<int64_t>(8192, (n4 + 255) / 256)); }

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_adamw_step_bf16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                             float beta2, float eps, float weight_decay, int64_t step, const uint64_t* tick, float grad_scale,
                                             uint16_t* bf16_out, void* stream) {
    SNAG_REQUIRE(n >= 0 && (step >= 1 || tick != nullptr), SNAG_ERR_ARG, "adamw_step_bf16: n=%ld step=%ld", (long)n, (long)step);
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE(param && grad && exp_avg && exp_avg_sq, SNAG_ERR_ARG, "adamw_step_bf16: null pointer");
    SNAG_REQUIRE(n % 4 == 0, SNAG_ERR_UNSUPPORTED, "adamw_step_bf16: n=%ld must be a multiple of 4 (whole rows of a dim %% 4 == 0 table)", (long)n);
    auto al = [](const void* q, int a) { return ((uintptr_t)q & (a - 1)) == 0; };
    SNAG_REQUIRE(al(param, 16) && al(grad, 16) && al(exp_avg, 16) && al(exp_avg_sq, 16) && (!bf16_out || al(bf16_out, 8)), SNAG_ERR_UNSUPPORTED,
                 "adamw_step_bf16: buffers must be 16-byte aligned (bf16 copy: 8)");
    const float bc1 = 1.0f - powf(beta1, (float)std::max<int64_t>(step, 1)), bc2 = 1.0f - powf(beta2, (float)std::max<int64_t>(step, 1));
    hipLaunchKernelGGL(adamw_bf16_kernel, dim3(grid4(n / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n / 4, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, grad_scale, tick, bf16_out);
    return check_launch("snag_adamw_step_bf16");
}

extern "C" SNAG_API int snag_f32_to_bf16(const float* src, int64_t n, uint16_t* dst, void* stream) {
    SNAG_REQUIRE(n >= 0 && (n == 0 || (src && dst)), SNAG_ERR_ARG, "f32_to_bf16: bad arguments");
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE(n % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, SNAG_ERR_UNSUPPORTED, "f32_to_bf16: n %% 4 and 16-byte alignment");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid4(n / 4)), dim3(256), 0, (hipStream_t)stream, src, n / 4, dst);
    return check_launch("snag_f32_to_bf16");
}

namespace snag {
// Clear exactly the gradient rows one batch scattered into -- head, positive tail and n_neg negative tails of every triple --
// instead of a memset of the whole [n_ent, dim] buffer: at |E| = 1M x 256 a batch of 1,024 x (128 + 2) rows is 136 MB against 1 GB.
template <typename IdxT>
__global__ __launch_bounds__(256) void zero_touched_rows_kernel(float* __restrict__ d_ent, int64_t ld, int dim, const IdxT* __restrict__ h,
                                                                const IdxT* __restrict__ t, const IdxT* __restrict__ neg_t, int64_t batch, int n_neg) {
    const int64_t b = blockIdx.x;
    const int per = n_neg + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < per; k += 4) {
        const int64_t row = (k == 0) ? (int64_t)h[b] : (k == 1) ? (int64_t)t[b] : (int64_t)neg_t[b * n_neg + (k - 2)];
        float4* p = reinterpret_cast<float4*>(d_ent + row * ld);
        for (int c = lane; c < dim / 4; c += 64) p[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
}  // namespace snag

extern "C" SNAG_API int snag_zero_touched_rows(float* d_ent, int64_t d_ent_ld, int dim, const void* h, const void* t, const void* neg_t, int idx_bytes,
                                               int64_t batch, int n_neg, void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_neg >= 0 && dim >= 4 && dim % 4 == 0 && d_ent_ld % 4 == 0 && d_ent_ld >= dim, SNAG_ERR_ARG,
                 "zero_touched_rows: bad sizes (dim and row stride must be multiples of 4)");
    if (batch == 0) return SNAG_OK;
    SNAG_REQUIRE(d_ent && h && t && (neg_t || n_neg == 0) && (idx_bytes == 4 || idx_bytes == 8), SNAG_ERR_ARG, "zero_touched_rows: null pointer / idx_bytes");
    SNAG_REQUIRE(((uintptr_t)d_ent & 15) == 0, SNAG_ERR_UNSUPPORTED, "zero_touched_rows: d_ent must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (idx_bytes == 4) hipLaunchKernelGGL((zero_touched_rows_kernel<int32_t>), dim3((unsigned)batch), dim3(256), 0, st, d_ent, d_ent_ld, dim, (const int32_t*)h,
                                           (const int32_t*)t, (const int32_t*)neg_t, batch, n_neg);
    else hipLaunchKernelGGL((zero_touched_rows_kernel<int64_t>), dim3((unsigned)batch), dim3(256), 0, st, d_ent, d_ent_ld, dim, (const int64_t*)h,
                            (const int64_t*)t, (const int64_t*)neg_t, batch, n_neg);
    return check_launch("snag_zero_touched_rows");
}
