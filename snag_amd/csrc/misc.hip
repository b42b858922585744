// a6 noise-mask injection (model/SNAG.py:66-98, model/SNAG_tools.py:127-128), column statistics,
// Philox keep-masks for the Bert dropouts, bias-gradient column sums, the hd_mean read-out backward
// and the fused AdamW update that closes a training step.
#include <algorithm>

#include "fusion_kernels.hip.h"
#include "philox.hip.h"

namespace snag {

__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

// four standard normals from one Philox draw (Box-Muller)
__device__ __forceinline__ float4 normal4(uint64_t seed, uint64_t ctr) {
    const Philox4 r = philox4x32_10(seed, ctr);
    const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u01(r.y), &s0, &c0);
    sincosf(6.28318530717958647692f * u01(r.w), &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

// Bernoulli byte mask of the elements [offset, offset + n) of ONE global stream: element g comes from Philox(seed, g / 4), word g % 4 --
// whatever part of the stream a caller asks for (a rank that owns rows [lo, hi) of a table draws exactly what the single rank draws for
// those rows).  out[i] = (u(offset + i) >= p) != invert: invert = 0 is the inverted-dropout keep mask of nn.Dropout(p), invert = 1 the
// "rand < p" row mask of the noise injection (model/SNAG.py:68).
__device__ __forceinline__ void bernoulli_region_body(int64_t vblock, uint64_t seed, const uint64_t* __restrict__ tick, int64_t offset, int64_t n, float p,
                                                      int invert, uint8_t* __restrict__ out, int64_t period = 0, int64_t skip = 0, int64_t take = 0) {
    int64_t q;                                            // one Philox draw = 4 elements of the global stream
    if (period > 0) {                                     // strided: draw `take` elements at `skip` of every `period` (all multiples of 4)
        const int64_t tq = take >> 2, sel = vblock * 256 + threadIdx.x;
        if (sel >= (n / period) * tq) return;
        q = (offset + (sel / tq) * period + skip) / 4 + sel % tq;
    } else {
        q = (offset >> 2) + vblock * 256 + threadIdx.x;
        if (q * 4 >= offset + n) return;
    }
    if (tick != nullptr) seed += tick[0] * 0x9E3779B97F4A7C15ull;    // a fresh stream per training step inside a replayed graph
    const Philox4 r = philox4x32_10(seed, (uint64_t)q);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
    const uint32_t inv = invert ? 1u : 0u;
    const int64_t i0 = q * 4 - offset;                  // local index of the draw's first element
    if (i0 >= 0 && i0 + 3 < n && (((uintptr_t)(out + i0)) & 3) == 0) {          // one 4-byte store per draw
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= ((u01(rr[e]) >= p ? 1u : 0u) ^ inv) << (8 * e);
        *reinterpret_cast<uint32_t*>(out + i0) = w;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i0 + e >= 0 && i0 + e < n) out[i0 + e] = (uint8_t)((u01(rr[e]) >= p ? 1u : 0u) ^ inv);
}

__device__ __forceinline__ void bernoulli_keep_body(int64_t vblock, uint64_t seed, const uint64_t* __restrict__ tick, int64_t n, float p,
                                                    uint8_t* __restrict__ out) {
    bernoulli_region_body(vblock, seed, tick, 0, n, p, 0, out);
}

__global__ __launch_bounds__(256) void bernoulli_keep_kernel(uint64_t seed, const uint64_t* __restrict__ tick, int64_t n, float p,
                                                             uint8_t* __restrict__ out) {
    bernoulli_keep_body(blockIdx.x, seed, tick, n, p, out);
}

// up to SNAG_MAX_MASK_REGIONS regions in one launch: blocks [cum[i-1], cum[i]) draw region i
struct RegionArgs {
    snag_mask_region r[SNAG_MAX_MASK_REGIONS];
    int cum[SNAG_MAX_MASK_REGIONS];
    int n;
};
__device__ __forceinline__ void bernoulli_regions_body(int vblock, const RegionArgs& a, const uint64_t* __restrict__ tick) {
    int i = 0;
    while (i + 1 < a.n && vblock >= a.cum[i]) ++i;
    const int first = i ? a.cum[i - 1] : 0;
    bernoulli_region_body(vblock - first, a.r[i].seed, tick, a.r[i].offset, a.r[i].n, a.r[i].p, a.r[i].invert, a.r[i].out, a.r[i].period, a.r[i].skip, a.r[i].take);
}
__global__ __launch_bounds__(256) void bernoulli_regions_kernel(RegionArgs a, const uint64_t* __restrict__ tick) {
    bernoulli_regions_body(blockIdx.x, a, tick);
}
static int region_blocks(const snag_mask_region& r) {
    if (r.n <= 0) return 0;
    if (r.period > 0) return (int)(((r.n / r.period) * (r.take / 4) + 255) / 256);
    const int64_t q0 = r.offset >> 2, q1 = (r.offset + r.n + 3) >> 2;
    return (int)((q1 - q0 + 255) / 256);
}
// validates and lays out a region table; returns the total block count or a negative error
static int fill_regions(const snag_mask_region* regions, int n_regions, RegionArgs& a, const char* who) {
    if (n_regions < 0 || n_regions > SNAG_MAX_MASK_REGIONS || (n_regions > 0 && regions == nullptr))
        return fail(SNAG_ERR_ARG, "%s: %d mask regions (at most %d)", who, n_regions, SNAG_MAX_MASK_REGIONS);
    a.n = 0;
    int total = 0;
    for (int i = 0; i < n_regions; ++i) {
        const snag_mask_region& r = regions[i];
        if (r.n < 0 || r.offset < 0 || (r.n > 0 && r.out == nullptr) || !(r.p >= 0.f && r.p < 1.f))
            return fail(SNAG_ERR_ARG, "%s: bad mask region %d (n=%ld offset=%ld p=%g)", who, i, (long)r.n, (long)r.offset, r.p);
        if (r.period != 0 && !(r.period > 0 && r.skip >= 0 && r.take > 0 && r.skip + r.take <= r.period && r.n % r.period == 0 &&
                               ((r.offset | r.period | r.skip | r.take) & 3) == 0))
            return fail(SNAG_ERR_ARG, "%s: bad strided mask region %d (period=%ld skip=%ld take=%ld n=%ld offset=%ld: multiples of 4, n a multiple of period)", who, i,
                        (long)r.period, (long)r.skip, (long)r.take, (long)r.n, (long)r.offset);
        const int nb = region_blocks(r);
        if (nb == 0) continue;
        total += nb;
        a.r[a.n] = r;
        a.cum[a.n] = total;
        ++a.n;
    }
    return total;
}

// out[n, c] = mean[c] + std[c] * z    (model/SNAG.py:96)
__global__ __launch_bounds__(256) void gaussian_fill_kernel(uint64_t seed, int64_t N, int F, const float* __restrict__ mean,
                                                            const float* __restrict__ std_, float* __restrict__ out) {
    const int64_t total = N * (int64_t)F;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= total) return;
    const float4 z = normal4(seed, (uint64_t)q);
    const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = q * 4 + e;
        if (i < total) {
            const int c = (int)(i % F);
            out[i] = mean[c] + std_[c] * zz[e];
        }
    }
}

// Row-level Gaussian noise injection (model/SNAG.py:66-75):
//   masked row: out = (1 - eps) x + eps (mean + std * z) ; other rows: out = x.
// z comes from the caller (compact [n_sel, F], row sel_index[n]) or from Philox(seed, n*F + c).
__global__ __launch_bounds__(256) void noise_apply_kernel(const float* __restrict__ x, int64_t N, int F,
                                                          const float* __restrict__ mean, const float* __restrict__ std_,
                                                          const uint8_t* __restrict__ row_mask, const int32_t* __restrict__ sel_index,
                                                          const float* __restrict__ z, uint64_t seed, float eps,
                                                          float* __restrict__ out) {
    const int64_t total = N * (int64_t)F;
    // rows of whole quads, 16-byte addressable: one dwordx4 per operand and ONE Philox draw per quad (the element-wise form below draws
    // the quad's four normals for each of its elements and keeps one) -- the same values, element for element
    if ((F & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)z) | ((uintptr_t)mean) | ((uintptr_t)std_)) & 15) == 0) {
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q * 4 < total; q += (int64_t)gridDim.x * 256) {
            const int64_t i = q * 4, n = i / F;
            const int c = (int)(i % F);
            float4 v = *reinterpret_cast<const float4*>(x + i);
            if (row_mask[n]) {
                const float4 g = z != nullptr ? *reinterpret_cast<const float4*>(z + (int64_t)sel_index[n] * F + c) : normal4(seed, (uint64_t)q);
                const float4 mu = *reinterpret_cast<const float4*>(mean + c), sd = *reinterpret_cast<const float4*>(std_ + c);
                v.x = (1.0f - eps) * v.x + eps * (mu.x + sd.x * g.x);
                v.y = (1.0f - eps) * v.y + eps * (mu.y + sd.y * g.y);
                v.z = (1.0f - eps) * v.z + eps * (mu.z + sd.z * g.z);
                v.w = (1.0f - eps) * v.w + eps * (mu.w + sd.w * g.w);
            }
            *reinterpret_cast<float4*>(out + i) = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / F;
        const int c = (int)(i % F);
        float v = x[i];
        if (row_mask[n]) {
            float zv;
            if (z != nullptr) {
                zv = z[(int64_t)sel_index[n] * F + c];
            } else {
                const float4 g = normal4(seed, (uint64_t)(i >> 2));
                const int e = (int)(i & 3);
                zv = e == 0 ? g.x : (e == 1 ? g.y : (e == 2 ? g.z : g.w));
            }
            v = (1.0f - eps) * v + eps * (mean[c] + std_[c] * zv);
        }
        out[i] = v;
    }
}

// column statistics, two passes over fixed-order partial sums; rows with exclude[n] != 0 are skipped
__global__ __launch_bounds__(256) void colstat_part_kernel(const float* __restrict__ x, int64_t R, int C,
                                                           const uint8_t* __restrict__ exclude, const float* __restrict__ mean,
                                                           float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C) {
        const float mu = mean ? mean[c] : 0.f;
        for (int64_t r = (int64_t)blockIdx.y * 4 + wave; r < R; r += (int64_t)gridDim.y * 4) {
            if (exclude && exclude[r]) continue;
            const float v = x[r * C + c];
            s += mean ? (v - mu) * (v - mu) : v;
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// mode 0: out = sum / n ; mode 1: out = sqrt(sum / (n - 1)).  One workgroup per 64 columns: sixteen waves sum runs of consecutive
// partial blocks (eight loads in flight), the runs are added in wave order -- a fixed order, 16 x shorter than one thread's serial walk
// over all blocks (59 us for a 128-column table as a single 128-thread block).
__global__ __launch_bounds__(1024) void colstat_finish_kernel(const float* __restrict__ part, int nblocks, int C, float n, int mode,
                                                              float* __restrict__ out) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    const int per = (nblocks + 15) / 16;
    const int b0 = wave * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < C) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(b + u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < b1; ++b) s += part[(int64_t)b * C + c];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        out[c] = mode == 0 ? t / n : sqrtf(t / (n - 1.0f));
    }
}

// Refresh of one modality's noisy feature table, incremental (model/SNAG.py:66-75 once per epoch, :86-92):
//   row in new_mask          : noisy = (1 - eps) x + eps (mean + std * z),  z = Philox(seed, ((row0 + n) F + c) / 4): keyed on the GLOBAL row
//   row in old_mask only     : noisy = x   (the previous epoch's noise is taken back)
//   row in neither           : untouched -- it already equals x
// One workgroup walks rows blockIdx.x, + gridDim.x, ...; the masks are wave-uniform, so an untouched row costs two byte loads.
__global__ __launch_bounds__(256) void noise_refresh_rows_kernel(const float* __restrict__ x, int64_t N, int F, const float* __restrict__ mean,
                                                                 const float* __restrict__ std_, const uint8_t* __restrict__ new_mask,
                                                                 const uint8_t* __restrict__ old_mask, uint64_t seed, int64_t row0, float eps,
                                                                 float* __restrict__ out) {
    const bool quad = (F & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)mean) | ((uintptr_t)std_)) & 15) == 0;
    for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
        const int nm = new_mask[n], om = old_mask != nullptr ? old_mask[n] : 1;
        if (!(nm | om)) continue;
        const float* xr = x + n * F;
        float* o = out + n * F;
        const int64_t g0 = (row0 + n) * (int64_t)F;                 // global element index of the row's first column
        if (quad) {
            for (int c = threadIdx.x * 4; c < F; c += 1024) {
                float4 v = *reinterpret_cast<const float4*>(xr + c);
                if (nm) {
                    const float4 g = normal4(seed, (uint64_t)((g0 + c) >> 2));
                    const float4 mu = *reinterpret_cast<const float4*>(mean + c), sd = *reinterpret_cast<const float4*>(std_ + c);
                    v.x = (1.0f - eps) * v.x + eps * (mu.x + sd.x * g.x);
                    v.y = (1.0f - eps) * v.y + eps * (mu.y + sd.y * g.y);
                    v.z = (1.0f - eps) * v.z + eps * (mu.z + sd.z * g.z);
                    v.w = (1.0f - eps) * v.w + eps * (mu.w + sd.w * g.w);
                }
                *reinterpret_cast<float4*>(o + c) = v;
            }
        } else {
            for (int c = threadIdx.x; c < F; c += 256) {
                float v = xr[c];
                if (nm) {
                    const int64_t gi = g0 + c;
                    const float4 g = normal4(seed, (uint64_t)(gi >> 2));
                    const int e = (int)(gi & 3);
                    const float zv = e == 0 ? g.x : (e == 1 ? g.y : (e == 2 ? g.z : g.w));
                    v = (1.0f - eps) * v + eps * (mean[c] + std_[c] * zv);
                }
                o[c] = v;
            }
        }
    }
}

// The same refresh for engines that also keep the table TRANSPOSED ([F, ld_t], rows of the table along the contiguous axis): 64 x 64
// tiles through LDS (pitch 65), every element of the transposed copy rewritten, the row-major copy only where a mask touches it.
__global__ __launch_bounds__(256) void noise_refresh_tiles_kernel(const float* __restrict__ x, int64_t N, int F, const float* __restrict__ mean,
                                                                  const float* __restrict__ std_, const uint8_t* __restrict__ new_mask,
                                                                  const uint8_t* __restrict__ old_mask, uint64_t seed, int64_t row0, float eps,
                                                                  float* __restrict__ out, float* __restrict__ out_t, int64_t ld_t) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = c0 + tx;
    const float mu = c < F ? mean[c] : 0.f, sd = c < F ? std_[c] : 0.f;
    for (int i = ty; i < 64; i += 4) {
        const int64_t n = r0 + i;
        if (n >= N || c >= F) continue;
        float v = x[n * F + c];
        const int nm = new_mask[n], om = old_mask != nullptr ? old_mask[n] : 1;
        if (nm) {
            const int64_t gi = (row0 + n) * (int64_t)F + c;
            const float4 g = normal4(seed, (uint64_t)(gi >> 2));
            const int e = (int)(gi & 3);
            const float zv = e == 0 ? g.x : (e == 1 ? g.y : (e == 2 ? g.z : g.w));
            v = (1.0f - eps) * v + eps * (mu + sd * zv);
        }
        if (out != nullptr && (nm | om)) out[n * F + c] = v;
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < F && r0 + tx < N) out_t[(int64_t)(c0 + i) * ld_t + r0 + tx] = tile[tx][i];
}

// d hidden[n, m, :] = d_out[n, :] / M ; zero_src: d_out is overwritten with zeros once read (it is the scorer's scatter-add target)
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(float* __restrict__ d_out, int64_t N, int M, int D,
                                                             float* __restrict__ d_hidden, int zero_src) {
    const int64_t total = N * (int64_t)D;
    const float inv = 1.0f / (float)M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % D);
        const int64_t n = i / D;
        const float v = d_out[i] * inv;
        if (zero_src) d_out[i] = 0.f;
        for (int m = 0; m < M; ++m) d_hidden[(n * M + m) * D + c] = v;
    }
}

// entity-table gradient through the noise blend: d_table[n,:] (=|+=) scale_n * d_tok[n,:], scale_n = 1 - a for masked rows
__global__ __launch_bounds__(256) void scaled_rows_kernel(const float* __restrict__ d_tok, int64_t ld, int64_t N, int D,
                                                          const uint8_t* __restrict__ nmask, float a, float* __restrict__ d_table,
                                                          int accumulate) {
    const int64_t total = N * (int64_t)D;
    if (D % 4 == 0 && ld % 4 == 0 && ((((uintptr_t)d_tok) | ((uintptr_t)d_table)) & 15) == 0) {       // four columns per thread
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q * 4 < total; q += (int64_t)gridDim.x * 256) {
            const int64_t i = q * 4, n = i / D;
            const int c = (int)(i % D);
            float4 v = *reinterpret_cast<const float4*>(d_tok + n * ld + c);
            if (nmask != nullptr && nmask[n]) { v.x *= (1.0f - a); v.y *= (1.0f - a); v.z *= (1.0f - a); v.w *= (1.0f - a); }
            float4* o = reinterpret_cast<float4*>(d_table + i);
            if (accumulate) { const float4 t = *o; v.x = t.x + v.x; v.y = t.y + v.y; v.z = t.z + v.z; v.w = t.w + v.w; }
            *o = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / D;
        const int c = (int)(i % D);
        float v = d_tok[n * ld + c];
        if (nmask != nullptr && nmask[n]) v *= (1.0f - a);
        d_table[i] = accumulate ? d_table[i] + v : v;
    }
}

// torch.optim.AdamW semantics (decoupled weight decay), one launch over a flat parameter buffer
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, float gscale,
                                                    const uint64_t* __restrict__ tick, const int64_t* __restrict__ seg_end,
                                                    const float* __restrict__ seg_wd, const float* __restrict__ seg_lr, int n_seg,
                                                    const float* __restrict__ gscale_dev, int tick_advanced, int vec,
                                                    const float* __restrict__ norm_parts = nullptr, int n_parts = 0, float max_norm = 0.f,
                                                    float* __restrict__ clip_out = nullptr) {
    if (gscale_dev != nullptr) gscale *= gscale_dev[0];      // e.g. the clip coefficient of snag_clip_grad_norm
    if (norm_parts != nullptr) {
        // clip_grad_norm_'s coefficient from the sum-of-squares partials, computed by EVERY workgroup in the same fixed order (4 KB of
        // L2-resident partials): the single-block launch that used to turn them into the coefficient is gone
        __shared__ float red[4];
        float s = 0.f;
        for (int i = threadIdx.x; i < n_parts; i += 256) s += norm_parts[i];
        s = sum64(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float norm = sqrtf(((red[0] + red[1]) + red[2]) + red[3]);
        const float coef = fminf(1.0f, max_norm / (norm + 1e-6f));
        gscale *= coef;
        if (clip_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { clip_out[0] = coef; clip_out[1] = norm; }
    }
    if (tick != nullptr) {          // step number lives on the device (graph replay): t = tick + 1 (tick itself when already advanced)
        const float t = (float)(tick[0] + (tick_advanced ? 0 : 1));
        bc1 = 1.0f - powf(b1, t);
        bc2 = 1.0f - powf(b2, t);
    }
    // four consecutive elements per lane (one dwordx4 per operand); a segment boundary inside a quad is honoured per element
    auto group_of = [&](int64_t i, float& lr_i, float& wd_i) {
        lr_i = lr; wd_i = wd;
        if (n_seg > 0) {                    // parameter groups: segment s covers [seg_end[s-1], seg_end[s])
            int lo = 0, hi = n_seg - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (i < seg_end[mid]) hi = mid;
                else lo = mid + 1;
            }
            lr_i = lr * seg_lr[lo];
            wd_i = seg_wd[lo];
        }
    };
    auto update = [&](float& pi, float gi, float& mi, float& vi, float lr_i, float wd_i) {
        gi *= gscale;
        const float pd = pi * (1.0f - lr_i * wd_i);
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        pi = pd - (lr_i / bc1) * (mi / denom);
    };
    const int64_t n4 = vec ? n / 4 : 0;
    // A thread keeps the parameter group of its last quad: [seg_lo, seg_hi) -- a quad inside it needs no search (the groups are a handful of long
    // runs; two binary searches per quad, eight dependent loads in front of the update, paced the launch: 22 us for 77 MB).  Two quads per
    // iteration, all eight operand loads requested before the first update.
    int64_t seg_lo = 0, seg_hi = n_seg > 0 ? -1 : n;
    float c_lr = lr, c_wd = wd;
    auto locate = [&](int64_t i) {
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (i < seg_end[mid]) hi = mid;
            else lo = mid + 1;
        }
        seg_lo = lo > 0 ? seg_end[lo - 1] : 0;
        seg_hi = seg_end[lo];
        c_lr = lr * seg_lr[lo];
        c_wd = seg_wd[lo];
    };
    auto quad = [&](int64_t q, const float4& g4, float4& p4, float4& m4, float4& v4) {
        if (!(4 * q >= seg_lo && 4 * q < seg_hi)) locate(4 * q);
        float* pp = &p4.x; float* mm = &m4.x; float* vv = &v4.x; const float* gg = &g4.x;
        if (4 * q + 3 < seg_hi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) update(pp[e], gg[e], mm[e], vv[e], c_lr, c_wd);
        } else {                              // a group boundary inside the quad: per element
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lr_e, wd_e;
                group_of(4 * q + e, lr_e, wd_e);
                update(pp[e], gg[e], mm[e], vv[e], lr_e, wd_e);
            }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += 2 * stride) {
        const int64_t q1 = q + stride;
        const bool two = q1 < n4;
        const int64_t qb = two ? q1 : q;
        const float4 ga = reinterpret_cast<const float4*>(g)[q], gb = reinterpret_cast<const float4*>(g)[qb];
        float4 pa = reinterpret_cast<const float4*>(p)[q], ma = reinterpret_cast<const float4*>(m)[q], va = reinterpret_cast<const float4*>(v)[q];
        float4 pb = reinterpret_cast<const float4*>(p)[qb], mb = reinterpret_cast<const float4*>(m)[qb], vb = reinterpret_cast<const float4*>(v)[qb];
        quad(q, ga, pa, ma, va);
        reinterpret_cast<float4*>(p)[q] = pa;
        reinterpret_cast<float4*>(m)[q] = ma;
        reinterpret_cast<float4*>(v)[q] = va;
        if (two) {
            quad(q1, gb, pb, mb, vb);
            reinterpret_cast<float4*>(p)[q1] = pb;
            reinterpret_cast<float4*>(m)[q1] = mb;
            reinterpret_cast<float4*>(v)[q1] = vb;
        }
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float lr_i, wd_i, pi = p[i], mi = m[i], vi = v[i];
        group_of(i, lr_i, wd_i);
        update(pi, g[i], mi, vi, lr_i, wd_i);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

__global__ void tick_kernel(uint64_t* tick) {
    if (threadIdx.x == 0 && blockIdx.x == 0) tick[0] += 1;
}

// structural token: out[n*ld + :] = blend(table[n,:])   (model/SNAG_tools.py:127-128)
__device__ __forceinline__ void entity_token_fwd_body(int64_t vblock, int64_t vgrid, const float* __restrict__ table, int64_t N, int D,
                                                      const float* __restrict__ noise, const uint8_t* __restrict__ nmask,
                                                      float a, float* __restrict__ out, int64_t ld) {
    const int64_t total = N * (int64_t)D;
    const bool quad = D % 4 == 0 && ld % 4 == 0 && ((((uintptr_t)table) | ((uintptr_t)out) | ((uintptr_t)noise)) & 15) == 0;
    if (quad) {                 // four columns per thread: dwordx4 in, dwordx4 out
        for (int64_t q = vblock * 256 + threadIdx.x; q * 4 < total; q += vgrid * 256) {
            const int64_t i = q * 4, n = i / D;
            const int c = (int)(i % D);
            float4 v = *reinterpret_cast<const float4*>(table + i);
            if (nmask != nullptr && nmask[n]) {
                const float4 z = *reinterpret_cast<const float4*>(noise + i);
                v.x = (1.0f - a) * v.x + a * z.x; v.y = (1.0f - a) * v.y + a * z.y; v.z = (1.0f - a) * v.z + a * z.z; v.w = (1.0f - a) * v.w + a * z.w;
            }
            *reinterpret_cast<float4*>(out + n * ld + c) = v;
        }
        return;
    }
    for (int64_t i = vblock * 256 + threadIdx.x; i < total; i += vgrid * 256) {
        const int64_t n = i / D;
        const int c = (int)(i % D);
        float v = table[i];
        if (nmask != nullptr && nmask[n]) v = (1.0f - a) * v + a * noise[i];
        out[n * ld + c] = v;
    }
}

__global__ __launch_bounds__(256) void entity_token_fwd_kernel(const float* __restrict__ table, int64_t N, int D,
                                                               const float* __restrict__ noise, const uint8_t* __restrict__ nmask,
                                                               float a, float* __restrict__ out, int64_t ld) {
    entity_token_fwd_body(blockIdx.x, gridDim.x, table, N, D, noise, nmask, a, out, ld);
}

// On-device negative sampling (in-tree sampler: generate_neg_triples_fast, model/MSNEA_tools.py:59-90).
// One thread per (positive b, slot k).  mode 0 corrupts the tail only (MKGC: NEG_NUM negative tails); mode 1 flips a
// Bernoulli(0.5) coin per POSITIVE (as the reference does, :68) between corrupting the head and the tail.  A candidate
// that forms a known triple is rejected and redrawn, up to 10 tries, the last try is kept whatever it is (:66-76).
// known: sorted int64 keys (h * n_rel + r) * n_ent + t, or NULL for no filtering.
__device__ __forceinline__ void sample_negatives_body(int64_t vblock, uint64_t seed, const uint64_t* __restrict__ tick, const int32_t* __restrict__ h,
                                                      const int32_t* __restrict__ r, const int32_t* __restrict__ t, int64_t B, int K,
                                                      int n_ent, int n_rel, int mode, const int64_t* __restrict__ known, int64_t n_known,
                                                      int32_t* __restrict__ neg_h, int32_t* __restrict__ neg_t) {
    const int64_t i = vblock * 256 + threadIdx.x;
    if (i >= B * K) return;
    if (tick != nullptr) seed += tick[0] * 0x9E3779B97F4A7C15ull;
    const int64_t b = i / K;
    const int hb = h[b], rb = r[b], tb = t[b];
    bool corrupt_head = false;
    if (mode == 1) corrupt_head = (philox4x32_10(seed ^ 0xC0FFEEull, (uint64_t)b).x & 1u) != 0;
    int cand = 0;
    for (int attempt = 0; attempt < 10; ++attempt) {
        const Philox4 p = philox4x32_10(seed, (uint64_t)i * 16 + attempt);
        cand = (int)(((uint64_t)p.x * (uint64_t)n_ent) >> 32);             // uniform in [0, n_ent), no modulo bias to speak of
        if (known == nullptr || attempt == 9) break;
        const int64_t key = corrupt_head ? ((int64_t)cand * n_rel + rb) * n_ent + tb : ((int64_t)hb * n_rel + rb) * n_ent + cand;
        if (!key_known(known, n_known, key)) break;
    }
    if (neg_h != nullptr) neg_h[i] = corrupt_head ? cand : hb;
    neg_t[i] = corrupt_head ? tb : cand;
}

__global__ __launch_bounds__(256) void sample_negatives_kernel(uint64_t seed, const uint64_t* __restrict__ tick, const int32_t* __restrict__ h,
                                                               const int32_t* __restrict__ r, const int32_t* __restrict__ t, int64_t B, int K,
                                                               int n_ent, int n_rel, int mode, const int64_t* __restrict__ known, int64_t n_known,
                                                               int32_t* __restrict__ neg_h, int32_t* __restrict__ neg_t) {
    sample_negatives_body(blockIdx.x, seed, tick, h, r, t, B, K, n_ent, n_rel, mode, known, n_known, neg_h, neg_t);
}

// The three launches that open an MKGC training step -- negative sampling (a dependent chain of ~20 loads per thread in 128
// workgroups), the structural token (7.7 MB copy + noise blend) and the Philox dropout masks -- are independent of each other and
// each far too small to fill 256 CUs: ONE launch whose workgroup ranges run the three bodies side by side (11.9 + 6.7 + 5.8 us of
// serial launches -> the longest of the three).  Same arithmetic, same Philox streams as the separate kernels.
struct PrologueArgs {
    // sampler (blocks [0, nb_s))
    uint64_t s_seed; const uint64_t* tick; const int32_t *h, *r, *t; int64_t B; int K, n_ent, n_rel, mode; const int64_t* known; int64_t n_known;
    int32_t *neg_h, *neg_t;
    // structural token (blocks [nb_s, nb_s + nb_e))
    const float* table; int64_t N; int D; const float* noise; const uint8_t* nmask; float a; float* tok; int64_t tok_ld;
    // Bernoulli mask regions (the rest): the Bert dropout masks, one region per mask tensor
    RegionArgs masks;
    // parameter-sized products (the last blocks): the folded-projection weights Wf_m = Wkv l2_m of the step
    SmallMMSet mm;
    int nb_s, nb_e, nb_d, nb_m;
};

__global__ __launch_bounds__(256) void step_prologue_kernel(PrologueArgs a) {
    // (the two roles that are dependent chains -- the parameter-sized products and the sampler -- take the FIRST workgroups: dispatched
    // behind 8,000 streaming workgroups they would start late and their latency would be the launch's tail)
    int b = blockIdx.x;
    if (b < a.nb_m) {
        small_matmul_body(a.mm, b);
        return;
    }
    b -= a.nb_m;
    if (b < a.nb_s) sample_negatives_body(b, a.s_seed, a.tick, a.h, a.r, a.t, a.B, a.K, a.n_ent, a.n_rel, a.mode, a.known, a.n_known, a.neg_h, a.neg_t);
    else if (b < a.nb_s + a.nb_e) entity_token_fwd_body(b - a.nb_s, a.nb_e, a.table, a.N, a.D, a.noise, a.nmask, a.a, a.tok, a.tok_ld);
    else bernoulli_regions_body(b - a.nb_s - a.nb_e, a.masks, a.tick);
}

static int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, (n + 255) / 256); }
// AdamW over a flat buffer: four workgroups per CU walk it with a stride (a workgroup's preamble -- the clip coefficient from the partials -- is paid 1,024 times, not 2,650)
static int adamw_grid(int64_t n4) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n4 + 255) / 256)); }

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_dropout_keep_mask(uint64_t seed, const uint64_t* tick, int64_t n, float p, uint8_t* out, void* stream) {
    SNAG_REQUIRE(n >= 0 && (out || n == 0) && p >= 0.f && p < 1.f, SNAG_ERR_ARG, "dropout_keep_mask: bad arguments (n=%ld p=%g)", (long)n, p);
    if (n == 0) return SNAG_OK;
    hipLaunchKernelGGL(bernoulli_keep_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, seed, tick, n, p, out);
    return check_launch("snag_dropout_keep_mask");
}

extern "C" SNAG_API int snag_bernoulli_regions(const snag_mask_region* regions, int n_regions, const uint64_t* tick, void* stream) {
    RegionArgs a{};
    const int total = fill_regions(regions, n_regions, a, "bernoulli_regions");
    if (total < 0) return total;
    if (total == 0) return SNAG_OK;
    hipLaunchKernelGGL(bernoulli_regions_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a, tick);
    return check_launch("snag_bernoulli_regions");
}

extern "C" SNAG_API int snag_gaussian_fill(uint64_t seed, int64_t N, int F, const float* mean, const float* std_, float* out, void* stream) {
    SNAG_REQUIRE(N >= 0 && F >= 1 && mean && std_ && (out || N == 0), SNAG_ERR_ARG, "gaussian_fill: bad arguments");
    if (N == 0) return SNAG_OK;
    const int64_t total = N * (int64_t)F;
    hipLaunchKernelGGL(gaussian_fill_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, seed, N, F, mean, std_, out);
    return check_launch("snag_gaussian_fill");
}

extern "C" SNAG_API int snag_noise_apply(const float* x, int64_t N, int F, const float* mean, const float* std_, const uint8_t* row_mask,
                                         const int32_t* sel_index, const float* z, uint64_t seed, float mask_ratio, float* out,
                                         void* stream) {
    SNAG_REQUIRE(N >= 0 && F >= 1, SNAG_ERR_ARG, "noise_apply: bad sizes");
    if (N == 0) return SNAG_OK;
    SNAG_REQUIRE(x && mean && std_ && row_mask && out, SNAG_ERR_ARG, "noise_apply: null pointer");
    SNAG_REQUIRE((z == nullptr) == (sel_index == nullptr), SNAG_ERR_ARG, "noise_apply: z and sel_index go together");
    hipLaunchKernelGGL(noise_apply_kernel, dim3(grid_for(N * (int64_t)F)), dim3(256), 0, (hipStream_t)stream, x, N, F, mean, std_, row_mask,
                       sel_index, z, seed, mask_ratio, out);
    return check_launch("snag_noise_apply");
}

extern "C" SNAG_API int snag_noise_refresh(const float* x, int64_t N, int F, const float* mean, const float* std_, const uint8_t* new_mask,
                                           const uint8_t* old_mask, uint64_t seed, int64_t row0, float mask_ratio, float* noisy, float* noisy_t,
                                           int64_t ld_t, void* stream) {
    SNAG_REQUIRE(N >= 0 && F >= 1 && row0 >= 0, SNAG_ERR_ARG, "noise_refresh: bad sizes");
    if (N == 0) return SNAG_OK;
    SNAG_REQUIRE(x && mean && std_ && new_mask && (noisy || noisy_t), SNAG_ERR_ARG, "noise_refresh: null pointer");
    SNAG_REQUIRE(noisy_t == nullptr || ld_t >= N, SNAG_ERR_ARG, "noise_refresh: transposed pitch %ld < %ld rows", (long)ld_t, (long)N);
    hipStream_t st = (hipStream_t)stream;
    if (noisy_t != nullptr) {
        SNAG_REQUIRE(ceil_div(N, 64) <= 65535, SNAG_ERR_UNSUPPORTED, "noise_refresh: more than 4M rows with a transposed copy");
        hipLaunchKernelGGL(noise_refresh_tiles_kernel, dim3(ceil_div(F, 64), ceil_div(N, 64)), dim3(256), 0, st, x, N, F, mean, std_, new_mask, old_mask,
                           seed, row0, mask_ratio, noisy, noisy_t, ld_t);
    } else {
        hipLaunchKernelGGL(noise_refresh_rows_kernel, dim3((unsigned)std::min<int64_t>(N, 8192)), dim3(256), 0, st, x, N, F, mean, std_, new_mask, old_mask,
                           seed, row0, mask_ratio, noisy);
    }
    return check_launch("snag_noise_refresh");
}

extern "C" SNAG_API int64_t snag_colstats_work_floats(int C) { return (int64_t)256 * C; }

extern "C" SNAG_API int snag_colstats(const float* x, int64_t R, int C, const uint8_t* exclude, int64_t n_valid, float* mean, float* std_,
                                      float* work, void* stream) {
    SNAG_REQUIRE(x && mean && work && R >= 1 && C >= 1, SNAG_ERR_ARG, "colstats: bad arguments");
    SNAG_REQUIRE(n_valid >= 1 && n_valid <= R, SNAG_ERR_ARG, "colstats: n_valid=%ld outside [1, %ld]", (long)n_valid, (long)R);
    hipStream_t st = (hipStream_t)stream;
    const int by = (int)std::min<int64_t>(256, (R + 3) / 4);
    const dim3 grid(ceil_div(C, 64), by), fgrid(ceil_div(C, 64));
    hipLaunchKernelGGL(colstat_part_kernel, grid, dim3(256), 0, st, x, R, C, exclude, (const float*)nullptr, work);
    hipLaunchKernelGGL(colstat_finish_kernel, fgrid, dim3(1024), 0, st, (const float*)work, by, C, (float)n_valid, 0, mean);
    if (std_ != nullptr) {
        hipLaunchKernelGGL(colstat_part_kernel, grid, dim3(256), 0, st, x, R, C, exclude, (const float*)mean, work);
        hipLaunchKernelGGL(colstat_finish_kernel, fgrid, dim3(1024), 0, st, (const float*)work, by, C, (float)n_valid, 1, std_);
    }
    return check_launch("snag_colstats");
}

extern "C" SNAG_API int snag_colsum(const float* x, int64_t R, int C, int64_t ld, float* out, int accumulate, float* work, void* stream) {
    SNAG_REQUIRE((x || R == 0) && out && work && R >= 0 && C >= 1 && ld >= C, SNAG_ERR_ARG, "colsum: bad arguments");     // (no rows: zeros)
    hipStream_t st = (hipStream_t)stream;
    const int by = (int)std::max<int64_t>(1, std::min<int64_t>(256, (R + 3) / 4));
    hipLaunchKernelGGL(colsum_part_kernel, dim3(ceil_div(C, 64), by), dim3(256), 0, st, x, R, C, ld, work);
    hipLaunchKernelGGL(slab_colsum_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, st, (const float*)work, by, (int64_t)C, C, out, accumulate);
    return check_launch("snag_colsum");
}

namespace snag {
// two column-sum problems of one shape in one pair of launches (the two modalities' projection bias gradients): blockIdx.z = problem
__global__ __launch_bounds__(256) void colsum_part_pair_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int64_t R, int C, int64_t ld,
                                                               float* __restrict__ part) {
    __shared__ float red[4][64];
    const float* x = blockIdx.z ? x1 : x0;
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C)
        for (int64_t r = (int64_t)blockIdx.y * 4 + wave; r < R; r += (int64_t)gridDim.y * 4) s += x[r * ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C)
        part[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}
__global__ __launch_bounds__(1024) void slab_colsum_pair_kernel(const float* __restrict__ part, int nblocks, int ncol, float* __restrict__ out0,
                                                                float* __restrict__ out1, int accumulate) {
    __shared__ float red[16][64];
    part += (int64_t)blockIdx.y * nblocks * ncol;
    float* out = blockIdx.y ? out1 : out0;
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    const int per = (nblocks + 15) / 16;
    const int b0 = wave * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < ncol) {              // the order of slab_colsum_kernel: sixteen runs of consecutive blocks, then the runs in wave order
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(b + u) * ncol + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < b1; ++b) s += part[(int64_t)b * ncol + c];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// dst[c, r] = src[r, c]: 64 x 64 tiles through LDS (pitch 65: conflict-free both ways), 256-byte runs on both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t ld_src, int64_t R, int64_t C,
                                                        float* __restrict__ dst, int64_t ld_dst) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(r0 + i) * ld_src + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < C && r0 + tx < R) dst[(c0 + i) * ld_dst + r0 + tx] = tile[tx][i];
}
}  // namespace snag

// two problems of one shape (R, C, ld) in one pair of launches; results bit-equal to two snag_colsum calls.  work: 2 x 256 x C floats
extern "C" SNAG_API int snag_colsum_pair(const float* x0, const float* x1, int64_t R, int C, int64_t ld, float* out0, float* out1, int accumulate, float* work,
                                         void* stream) {
    SNAG_REQUIRE(((x0 && x1) || R == 0) && out0 && out1 && work && R >= 0 && C >= 1 && ld >= C, SNAG_ERR_ARG, "colsum_pair: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int by = (int)std::max<int64_t>(1, std::min<int64_t>(256, (R + 3) / 4));
    hipLaunchKernelGGL(colsum_part_pair_kernel, dim3(ceil_div(C, 64), by, 2), dim3(256), 0, st, x0, x1, R, C, ld, work);
    hipLaunchKernelGGL(slab_colsum_pair_kernel, dim3(ceil_div(C, 64), 2), dim3(1024), 0, st, (const float*)work, by, C, out0, out1, accumulate);
    return check_launch("snag_colsum_pair");
}

extern "C" SNAG_API int snag_transpose_f32(const float* src, int64_t ld_src, int64_t rows, int64_t cols, float* dst, int64_t ld_dst, void* stream) {
    SNAG_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= rows, SNAG_ERR_ARG, "transpose: bad sizes");
    if (rows == 0 || cols == 0) return SNAG_OK;
    SNAG_REQUIRE(src && dst, SNAG_ERR_ARG, "transpose: null pointer");
    SNAG_REQUIRE(ceil_div(rows, 64) <= 65535, SNAG_ERR_UNSUPPORTED, "transpose: more than 4M rows");
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows, cols, dst, ld_dst);
    return check_launch("snag_transpose_f32");
}

extern "C" SNAG_API int snag_token_mean_bwd(float* d_out, int64_t N, int M, int D, float* d_hidden, int zero_src, void* stream) {
    SNAG_REQUIRE(d_out && d_hidden && N >= 0 && M >= 1 && D >= 1, SNAG_ERR_ARG, "token_mean_bwd: bad arguments");
    if (N == 0) return SNAG_OK;
    hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(grid_for(N * (int64_t)D)), dim3(256), 0, (hipStream_t)stream, d_out, N, M, D, d_hidden, zero_src);
    return check_launch("snag_token_mean_bwd");
}

extern "C" SNAG_API int snag_entity_token_bwd(const float* d_tok, int64_t ld, int64_t N, int D, const uint8_t* noise_mask,
                                              float half_mask_ratio, float* d_table, int accumulate, void* stream) {
    SNAG_REQUIRE(d_tok && d_table && N >= 0 && D >= 1 && ld >= D, SNAG_ERR_ARG, "entity_token_bwd: bad arguments");
    if (N == 0) return SNAG_OK;
    hipLaunchKernelGGL(scaled_rows_kernel, dim3(grid_for((N * (int64_t)D + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_tok, ld, N, D, noise_mask,
                       half_mask_ratio, d_table, accumulate);
    return check_launch("snag_entity_token_bwd");
}

namespace snag {
// torch.nn.utils.clip_grad_norm_ without leaving the device: fixed-order sum of squares in two levels, then
// out[0] = min(1, max_norm / (total_norm + 1e-6)), out[1] = total_norm.
constexpr int kNormBlocks = 1024;
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, int64_t n, float pre_scale, float* __restrict__ part,
                                                         uint64_t* __restrict__ advance = nullptr) {
    if (advance != nullptr && blockIdx.x == 0 && threadIdx.x == 0) advance[0] += 1;        // (nothing in this launch reads the counter)
    __shared__ float red[4];
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {          // dwordx4 body (fixed order per thread: x, y, z, w of its quads), scalar tail
        const int64_t quads = n / 4;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
            const float4 v4 = reinterpret_cast<const float4*>(g)[q];
            const float v[4] = {v4.x * pre_scale, v4.y * pre_scale, v4.z * pre_scale, v4.w * pre_scale};
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(v[e], v[e], s);
        }
        if (blockIdx.x == 0 && threadIdx.x < n - quads * 4) {
            const float v = g[quads * 4 + threadIdx.x] * pre_scale;
            s = fmaf(v, v, s);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const float v = g[i] * pre_scale;
            s = fmaf(v, v, s);
        }
    }
    s = sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ part, int nb, float max_norm, float* __restrict__ out,
                                                         uint64_t* __restrict__ advance) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 1024) s += part[i];
    s = sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 16; ++i) tot += red[i];
        const float norm = sqrtf(tot);
        out[0] = fminf(1.0f, max_norm / (norm + 1e-6f));
        out[1] = norm;
        if (advance) advance[0] += 1;          // single block, single thread: the step counter moves here instead of in its own launch
    }
}
}  // namespace snag

extern "C" SNAG_API int snag_clip_grad_norm(const float* grad, int64_t n, float grad_scale, float max_norm, float* out2, float* work,
                                            uint64_t* advance_tick, void* stream) {
    SNAG_REQUIRE(n >= 0 && max_norm > 0.f && out2 && work && (grad || n == 0), SNAG_ERR_ARG, "clip_grad_norm: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kNormBlocks, (n + 255) / 256));
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nb), dim3(256), 0, st, grad, n, grad_scale, work);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1024), 0, st, (const float*)work, nb, max_norm, out2, advance_tick);
    return check_launch("snag_clip_grad_norm");
}

// dwordx4 path: all four buffers 16-byte aligned
static int adamw_vec_ok(const void* a, const void* b, const void* c, const void* d) {
    return ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0) ? 1 : 0;
}

extern "C" SNAG_API int snag_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                        float beta2, float eps, float weight_decay, int64_t step, const uint64_t* tick, float grad_scale, void* stream) {
    SNAG_REQUIRE(n >= 0 && (step >= 1 || tick != nullptr), SNAG_ERR_ARG, "adamw_step: n=%ld step=%ld", (long)n, (long)step);
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE(param && grad && exp_avg && exp_avg_sq, SNAG_ERR_ARG, "adamw_step: null pointer");
    const float bc1 = 1.0f - powf(beta1, (float)std::max<int64_t>(step, 1)), bc2 = 1.0f - powf(beta2, (float)std::max<int64_t>(step, 1));
    hipLaunchKernelGGL(adamw_kernel, dim3(adamw_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2, grad_scale, tick, (const int64_t*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (const float*)nullptr, 0,
                       adamw_vec_ok(param, grad, exp_avg, exp_avg_sq));
    return check_launch("snag_adamw_step");
}

extern "C" SNAG_API int snag_adamw_step_groups(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                               float beta2, float eps, const int64_t* seg_end, const float* seg_weight_decay,
                                               const float* seg_lr_scale, int n_seg, int64_t step, const uint64_t* tick, float grad_scale,
                                               const float* grad_scale_dev, int tick_advanced, void* stream) {
    SNAG_REQUIRE(n >= 0 && (step >= 1 || tick != nullptr), SNAG_ERR_ARG, "adamw_step_groups: n=%ld step=%ld", (long)n, (long)step);
    SNAG_REQUIRE(n_seg >= 1 && seg_end && seg_weight_decay && seg_lr_scale, SNAG_ERR_ARG, "adamw_step_groups: segment table missing");
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE(param && grad && exp_avg && exp_avg_sq, SNAG_ERR_ARG, "adamw_step_groups: null pointer");
    const float bc1 = 1.0f - powf(beta1, (float)std::max<int64_t>(step, 1)), bc2 = 1.0f - powf(beta2, (float)std::max<int64_t>(step, 1));
    hipLaunchKernelGGL(adamw_kernel, dim3(adamw_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                       eps, 0.0f, bc1, bc2, grad_scale, tick, seg_end, seg_weight_decay, seg_lr_scale, n_seg, grad_scale_dev, tick_advanced,
                       adamw_vec_ok(param, grad, exp_avg, exp_avg_sq));
    return check_launch("snag_adamw_step_groups");
}

extern "C" SNAG_API int snag_clip_adamw_step_groups(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                                    float beta2, float eps, const int64_t* seg_end, const float* seg_weight_decay,
                                                    const float* seg_lr_scale, int n_seg, uint64_t* tick, float grad_scale, float max_norm,
                                                    float* out2, float* work, void* stream) {
    SNAG_REQUIRE(n >= 0 && tick != nullptr && max_norm > 0.f && work != nullptr, SNAG_ERR_ARG, "clip_adamw_step_groups: bad arguments");
    SNAG_REQUIRE(n_seg >= 1 && seg_end && seg_weight_decay && seg_lr_scale, SNAG_ERR_ARG, "clip_adamw_step_groups: segment table missing");
    if (n == 0) return snag_tick(tick, stream);
    SNAG_REQUIRE(param && grad && exp_avg && exp_avg_sq, SNAG_ERR_ARG, "clip_adamw_step_groups: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kNormBlocks, (n + 255) / 256));
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nb), dim3(256), 0, st, grad, n, grad_scale, work, tick);          // also advances the step counter
    hipLaunchKernelGGL(adamw_kernel, dim3(adamw_grid((n + 3) / 4)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, 0.0f, 1.0f, 1.0f,
                       grad_scale, (const uint64_t*)tick, seg_end, seg_weight_decay, seg_lr_scale, n_seg, (const float*)nullptr, 1,
                       adamw_vec_ok(param, grad, exp_avg, exp_avg_sq), (const float*)work, nb, max_norm, out2);
    return check_launch("snag_clip_adamw_step_groups");
}

extern "C" SNAG_API int snag_tick(uint64_t* tick, void* stream) {
    SNAG_REQUIRE(tick != nullptr, SNAG_ERR_ARG, "tick: null counter");
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tick);
    return check_launch("snag_tick");
}

extern "C" SNAG_API int snag_entity_token_fwd(const float* table, int64_t N, int D, const float* noise, const uint8_t* noise_mask,
                                              float half_mask_ratio, float* out, int64_t out_ld, void* stream) {
    SNAG_REQUIRE(table && out && N >= 0 && D >= 1 && out_ld >= D, SNAG_ERR_ARG, "entity_token_fwd: bad arguments");
    SNAG_REQUIRE((noise == nullptr) == (noise_mask == nullptr), SNAG_ERR_ARG, "entity_token_fwd: noise and noise_mask go together");
    if (N == 0) return SNAG_OK;
    hipLaunchKernelGGL(entity_token_fwd_kernel, dim3(grid_for(N * (int64_t)D)), dim3(256), 0, (hipStream_t)stream, table, N, D, noise,
                       noise_mask, half_mask_ratio, out, out_ld);
    return check_launch("snag_entity_token_fwd");
}

namespace snag {
// Filtered-setting mask for tail prediction: mask[b, x] = 1 for every known triple (h_b, r_b, x) except the query's own tail.
// known: ascending keys (h * n_rel + r) * n_ent + t (the sampler's table).  One wave per query: binary search of the (h, r) prefix,
// then the lanes walk the run of keys.
__global__ __launch_bounds__(256) void filter_mask_kernel(const int32_t* __restrict__ h, const int32_t* __restrict__ r, const int32_t* __restrict__ t,
                                                          int64_t batch, int n_ent, int n_rel, const int64_t* __restrict__ known, int64_t n_known,
                                                          uint8_t* __restrict__ mask) {
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (b >= batch) return;
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)h[b] * n_rel + r[b]) * n_ent;          // first key of the (h, r) run
    int64_t lo = 0, hi = n_known;
    while (lo < hi) {                                                       // lower_bound(base)
        const int64_t mid = (lo + hi) >> 1;
        if (known[mid] < base) lo = mid + 1;
        else hi = mid;
    }
    const int64_t tb = t[b];
    for (int64_t i = lo + lane; i < n_known; i += 64) {
        const int64_t x = known[i] - base;
        if (x >= n_ent) break;
        if (x != tb) mask[b * n_ent + x] = 1;
    }
}
}  // namespace snag

extern "C" SNAG_API int snag_build_filter_mask(const int32_t* h, const int32_t* r, const int32_t* t, int64_t batch, int n_ent, int n_rel,
                                               const int64_t* known_sorted, int64_t n_known, uint8_t* mask, void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_ent >= 1 && n_rel >= 1 && n_known >= 0, SNAG_ERR_ARG, "build_filter_mask: bad sizes");
    if (batch == 0) return SNAG_OK;
    SNAG_REQUIRE(h && r && t && mask && (known_sorted || n_known == 0), SNAG_ERR_ARG, "build_filter_mask: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(mask, 0, (size_t)batch * n_ent, st) != hipSuccess) return fail(SNAG_ERR_LAUNCH, "build_filter_mask: memset failed");
    if (n_known > 0)
        hipLaunchKernelGGL(filter_mask_kernel, dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, st, h, r, t, batch, n_ent, n_rel, known_sorted, n_known, mask);
    return check_launch("snag_build_filter_mask");
}

extern "C" SNAG_API int snag_step_prologue(uint64_t sample_seed, const uint64_t* tick, const int32_t* h, const int32_t* r, const int32_t* t, int64_t batch,
                                           int n_neg, int n_ent, int n_rel, const int64_t* known_sorted, int64_t n_known, int32_t* neg_t,
                                           const float* table, int64_t n_rows, int dim, const float* noise, const uint8_t* noise_mask, float half_mask_ratio,
                                           float* tok, int64_t tok_ld, const snag_mask_region* mask_regions, int n_mask_regions,
                                           const snag_small_mm* small_products, int n_small_products, void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_neg >= 0 && n_ent >= 1 && n_rel >= 1 && n_rows >= 0 && dim >= 1, SNAG_ERR_ARG, "step_prologue: bad sizes");
    SNAG_REQUIRE(batch * n_neg == 0 || (h && r && t && neg_t), SNAG_ERR_ARG, "step_prologue: null sampler pointer");
    SNAG_REQUIRE((known_sorted != nullptr) == (n_known > 0), SNAG_ERR_ARG, "step_prologue: known_sorted and n_known go together");
    SNAG_REQUIRE(n_rows == 0 || (table && tok && tok_ld >= dim), SNAG_ERR_ARG, "step_prologue: null structural-token pointer");
    SNAG_REQUIRE((noise == nullptr) == (noise_mask == nullptr), SNAG_ERR_ARG, "step_prologue: noise and noise_mask go together");
    SNAG_REQUIRE((int64_t)n_ent * n_rel < (1LL << 62) / n_ent, SNAG_ERR_UNSUPPORTED, "step_prologue: triple key overflows int64");
    PrologueArgs a{};
    a.s_seed = sample_seed; a.tick = tick; a.h = h; a.r = r; a.t = t; a.B = batch; a.K = n_neg; a.n_ent = n_ent; a.n_rel = n_rel; a.mode = 0;
    a.known = known_sorted; a.n_known = n_known; a.neg_h = nullptr; a.neg_t = neg_t;
    a.table = table; a.N = n_rows; a.D = dim; a.noise = noise; a.nmask = noise_mask; a.a = half_mask_ratio; a.tok = tok; a.tok_ld = tok_ld;
    const int nb_d = fill_regions(mask_regions, n_mask_regions, a.masks, "step_prologue");
    if (nb_d < 0) return nb_d;
    a.nb_s = (int)((batch * n_neg + 255) / 256);
    a.nb_e = n_rows > 0 ? grid_for((n_rows * (int64_t)dim + 3) / 4) : 0;          // (the dwordx4 form covers four elements per thread; the scalar form strides)
    a.nb_d = nb_d;
    const int nb_m = fill_small_mm(small_products, n_small_products, a.mm, "step_prologue");
    if (nb_m < 0) return nb_m;
    a.nb_m = nb_m;
    const int total = a.nb_s + a.nb_e + nb_d + nb_m;
    if (total == 0) return SNAG_OK;
    hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("snag_step_prologue");
}

extern "C" SNAG_API int snag_sample_negatives(uint64_t seed, const uint64_t* tick, const int32_t* h, const int32_t* r, const int32_t* t, int64_t batch,
                                              int n_neg, int n_ent, int n_rel, int mode, const int64_t* known_sorted, int64_t n_known,
                                              int32_t* neg_h, int32_t* neg_t, void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_neg >= 0 && n_ent >= 1 && n_rel >= 1, SNAG_ERR_ARG, "sample_negatives: bad sizes");
    SNAG_REQUIRE(mode == 0 || mode == 1, SNAG_ERR_ARG, "sample_negatives: mode=%d (0 = tails, 1 = head-or-tail)", mode);
    if (batch == 0 || n_neg == 0) return SNAG_OK;
    SNAG_REQUIRE(h && r && t && neg_t, SNAG_ERR_ARG, "sample_negatives: null pointer");
    SNAG_REQUIRE(mode == 0 || neg_h != nullptr, SNAG_ERR_ARG, "sample_negatives: head-or-tail mode needs neg_h");
    SNAG_REQUIRE((known_sorted != nullptr) == (n_known > 0), SNAG_ERR_ARG, "sample_negatives: known_sorted and n_known go together");
    SNAG_REQUIRE((int64_t)n_ent * n_rel < (1LL << 62) / n_ent, SNAG_ERR_UNSUPPORTED, "sample_negatives: triple key overflows int64");
    const int64_t total = batch * n_neg;
    hipLaunchKernelGGL(sample_negatives_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, tick, h, r, t, batch,
                       n_neg, n_ent, n_rel, mode, known_sorted, n_known, neg_h, neg_t);
    return check_launch("snag_sample_negatives");
}
