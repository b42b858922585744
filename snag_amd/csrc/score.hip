// a7: negative-sampled TransE-L2^2 margin scoring (reference: model/MSNEA.py:110-122,151-152).
//
// Forward: one wave per triple.  A row of `dim` floats is spread over a 16-lane DPP row as float4
// chunks (lane sub owns float4 #sub, #sub+16, ...), so one wave-instruction gathers four table rows
// as four fully-used 256-B segments and every reduction is four v_add_f32_dpp (no LDS, no bpermute).
// The head/relation/positive-tail rows are read by all four 16-lane groups (same lines, L1-served);
// negatives stream four per round, up to eight rounds in flight.
//
// Backward: one wave per triple, a row spread over the whole wave (col = lane + 64*i) so that every
// fp32 atomic wave-instruction adds 256 contiguous bytes of one gradient row (the full-rate shape).
#include "common.hip.h"
#include "score_rows.hip.h"

namespace snag {

template <int NV>
struct RowV4 {
    float4 v[NV];
};

template <int NV>
__device__ __forceinline__ RowV4<NV> load_row_v4(const float* __restrict__ base, int64_t row, int64_t ld, int dim, int sub) {
    RowV4<NV> r;
    const float4* p = reinterpret_cast<const float4*>(base + row * ld);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        r.v[j] = (c * 4 < dim) ? p[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return r;
}

template <int NV>
__device__ __forceinline__ float sumsq_v4(const RowV4<NV>& r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        s = fmaf(r.v[j].x, r.v[j].x, s);
        s = fmaf(r.v[j].y, r.v[j].y, s);
        s = fmaf(r.v[j].z, r.v[j].z, s);
        s = fmaf(r.v[j].w, r.v[j].w, s);
    }
    return s;
}

// || q - x / max(||x||, eps) ||^2, reduced over the 16-lane group
template <int NV>
__device__ __forceinline__ float score_row(const RowV4<NV>& q, const RowV4<NV>& x) {
    const float inv = 1.0f / l2_denominator(sum16(sumsq_v4<NV>(x)));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float d;
        d = q.v[j].x - x.v[j].x * inv; s = fmaf(d, d, s);
        d = q.v[j].y - x.v[j].y * inv; s = fmaf(d, d, s);
        d = q.v[j].z - x.v[j].z * inv; s = fmaf(d, d, s);
        d = q.v[j].w - x.v[j].w * inv; s = fmaf(d, d, s);
    }
    return sum16(s);
}

// WPT waves per triple: see score_fwd_bf16_kernel.
template <typename IdxT, int NV, int WPT>
__global__ __launch_bounds__(256) void score_fwd_kernel(
    const float* __restrict__ ent, int64_t eld, const float* __restrict__ rel, int dim,
    const IdxT* __restrict__ h, const IdxT* __restrict__ r, const IdxT* __restrict__ t,
    const IdxT* __restrict__ neg_t, int64_t batch, int n_neg_all, float margin,
    float* __restrict__ pos, float* __restrict__ neg, float* __restrict__ loss_part) {
    __shared__ float part_sum[4];
    const int wave = wave_id_uniform();
    const int64_t b_raw = (int64_t)blockIdx.x * (4 / WPT) + wave / WPT;
    const int part = wave % WPT;
    const bool live = b_raw < batch;
    if (WPT == 1 && !live) return;
    const int64_t b = live ? b_raw : 0;          // WPT > 1: dead waves still reach the barrier at the end, on triple 0 with no negatives
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;

    // Two dependent memory round trips in total: (1) every index this wave needs -- the three scalar ids and one coalesced
    // read of up to 64 negative-tail ids -- (2) every row: head, relation, positive tail and the first 32 negatives
    // (22 dwordx4 per lane in flight).  Only then does arithmetic start.
    const int64_t hi = (int64_t)h[b], ri = (int64_t)r[b], ti = (int64_t)t[b];
    const int per_part = (WPT == 1) ? n_neg_all : ((n_neg_all + WPT - 1) / WPT + 31) / 32 * 32;
    const int k_lo = min(n_neg_all, part * per_part);
    const int n_neg = live ? min(n_neg_all - k_lo, per_part) : 0;       // this wave's share of the negatives: [k_lo, k_lo + n_neg)
    const IdxT* nrow = neg_t + b * (int64_t)n_neg_all + k_lo;
    float* out = neg + b * (int64_t)n_neg_all + k_lo;
    int my = (lane < n_neg) ? (neg_t != nullptr ? (int)nrow[lane] : k_lo + lane) : 0;
    // keep the three scalar ids and the id vector in ONE wait: nothing below may be scheduled above this point
    asm volatile("" : "+v"(my) : "s"((int)hi), "s"((int)ri), "s"((int)ti) : "memory");
    constexpr int kFirst = (NV <= 2) ? 8 : 4;                 // rounds of 4 rows prefetched with the positive (register budget)
    const bool first_block = n_neg >= 4 * kFirst;
    RowV4<NV> x0[kFirst];
    if (first_block) {
#pragma unroll
        for (int u = 0; u < kFirst; ++u) x0[u] = load_row_v4<NV>(ent, __shfl(my, 4 * u + grp), eld, dim, sub);
    }
    RowV4<NV> q = load_row_v4<NV>(ent, hi, eld, dim, sub);
    RowV4<NV> rr = load_row_v4<NV>(rel, ri, dim, dim, sub);
    const RowV4<NV> tt = load_row_v4<NV>(ent, ti, eld, dim, sub);
    {
        const float ih = 1.0f / l2_denominator(sum16(sumsq_v4<NV>(q)));
        const float ir = 1.0f / l2_denominator(sum16(sumsq_v4<NV>(rr)));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            q.v[j].x = q.v[j].x * ih + rr.v[j].x * ir;
            q.v[j].y = q.v[j].y * ih + rr.v[j].y * ir;
            q.v[j].z = q.v[j].z * ih + rr.v[j].z * ir;
            q.v[j].w = q.v[j].w * ih + rr.v[j].w * ir;
        }
    }
    const float s_pos = score_row<NV>(q, tt);
    if (lane == 0 && part == 0 && live) pos[b] = s_pos;

    float hinge = 0.f;
    if (first_block) {
#pragma unroll
        for (int u = 0; u < kFirst; ++u) {
            const float s = score_row<NV>(q, x0[u]);
            hinge += fmaxf(margin + s_pos - s, 0.f);
            if (sub == 0) out[4 * u + grp] = s;
        }
    }
    for (int k0 = 0; k0 < n_neg; k0 += 64) {
        if (k0 > 0) my = (k0 + lane < n_neg) ? (neg_t != nullptr ? (int)nrow[k0 + lane] : k_lo + k0 + lane) : 0;
        const int cnt = min(64, n_neg - k0);
        int it = (k0 == 0 && first_block) ? 4 * kFirst : 0;
        // full blocks: 8 rounds x 4 rows, all 16 gathers of the block issued before the first use
        for (; it + 32 <= cnt; it += 32) {
            RowV4<NV> x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = load_row_v4<NV>(ent, __shfl(my, it + 4 * u + grp), eld, dim, sub);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float s = score_row<NV>(q, x[u]);
                hinge += fmaxf(margin + s_pos - s, 0.f);
                if (sub == 0) out[k0 + it + 4 * u + grp] = s;
            }
        }
        // ragged tail: one round (up to 4 rows) at a time
        for (; it < cnt; it += 4) {
            const int k = it + grp;
            const bool valid = k < cnt;
            const int idx = __shfl(my, k);  // all lanes take part: a bpermute from an inactive lane reads 0
            const RowV4<NV> x = load_row_v4<NV>(ent, valid ? idx : 0, eld, dim, sub);
            const float s = score_row<NV>(q, x);
            if (valid) {
                hinge += fmaxf(margin + s_pos - s, 0.f);
                if (sub == 0) out[k0 + k] = s;
            }
        }
    }
    // every lane of a group holds the group's partial; combine the four groups in fixed order
    const float g1 = __shfl(hinge, 16), g2 = __shfl(hinge, 32), g3 = __shfl(hinge, 48);
    const float wsum = ((hinge + g1) + g2) + g3;
    if (WPT == 1) {
        if (loss_part != nullptr && lane == 0) loss_part[b] = wsum;
        return;
    }
    if (lane == 0) part_sum[wave] = wsum;
    __syncthreads();
    if (loss_part != nullptr && lane == 0 && part == 0 && live) {
        float tot = 0.f;
#pragma unroll
        for (int p = 0; p < WPT; ++p) tot += part_sum[wave + p];
        loss_part[b] = tot;
    }
}

template <typename IdxT, int NS, typename AccT, typename ElemT>
__global__ __launch_bounds__(256) void score_bwd_kernel(
    const ElemT* __restrict__ ent, int64_t eld, const ElemT* __restrict__ rel, int dim,
    const IdxT* __restrict__ h, const IdxT* __restrict__ r, const IdxT* __restrict__ t,
    const IdxT* __restrict__ neg_t, int64_t batch, int n_neg, float margin, float gscale,
    const float* __restrict__ dpos, const float* __restrict__ dneg,
    AccT* __restrict__ d_ent, int64_t dld, AccT* __restrict__ d_rel) {
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (b >= batch) return;
    const int lane = threadIdx.x & 63;
    const bool hinge_mode = (dpos == nullptr);

    const int64_t hi = (int64_t)h[b], ri = (int64_t)r[b], ti = (int64_t)t[b];
    RowS<NS> hh = load_row_s<NS>(ent, hi, eld, dim, lane);
    RowS<NS> rr = load_row_s<NS>(rel, ri, dim, dim, lane);
    RowS<NS> tt = load_row_s<NS>(ent, ti, eld, dim, lane);
    const float den_h = normalize_s<NS>(hh);
    const float den_r = normalize_s<NS>(rr);
    const float den_t = normalize_s<NS>(tt);
    RowS<NS> q, dq;
    float s_pos = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        q.v[i] = hh.v[i] + rr.v[i];
        dq.v[i] = 0.f;
        const float d = q.v[i] - tt.v[i];
        s_pos = fmaf(d, d, s_pos);
    }
    s_pos = sum64(s_pos);

    float c_pos = hinge_mode ? 0.f : gscale * dpos[b];
    const IdxT* nrow = neg_t + b * (int64_t)n_neg;
    // one negative: recompute its score (hinge mode), fold its contribution into dq, scatter its own gradient
    auto one_negative = [&](int k, int64_t xi, RowS<NS>& x) {
        const float den_x = normalize_s<NS>(x);
        float c;
        if (hinge_mode) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float d = q.v[i] - x.v[i];
                s = fmaf(d, d, s);
            }
            s = sum64(s);
            c = (margin + s_pos - s > 0.f) ? -gscale : 0.f;
            c_pos -= c;
        } else {
            c = gscale * dneg[b * (int64_t)n_neg + k];
        }
        if (c != 0.f) {  // wave-uniform
            RowS<NS> dx;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float e2 = 2.f * c * (q.v[i] - x.v[i]);
                dq.v[i] += e2;
                dx.v[i] = -e2;
            }
            normalize_bwd_atomic<NS, AccT>(x, den_x, dx, d_ent + xi * dld, dim, lane);
        }
    };
    int k = 0;
    constexpr int kAhead = (NS <= 2) ? 8 : 4;        // rows gathered before the first is consumed: hides the table latency
    for (; k + kAhead <= n_neg; k += kAhead) {
        int64_t xi[kAhead];
        RowS<NS> x[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) xi[u] = (int64_t)nrow[k + u];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) x[u] = load_row_s<NS>(ent, xi[u], eld, dim, lane);
#pragma unroll
        for (int u = 0; u < kAhead; ++u) one_negative(k + u, xi[u], x[u]);
    }
    for (; k < n_neg; ++k) {
        const int64_t xi = (int64_t)nrow[k];
        RowS<NS> x = load_row_s<NS>(ent, xi, eld, dim, lane);
        one_negative(k, xi, x);
    }
    if (c_pos != 0.f) {
        RowS<NS> dx;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float e2 = 2.f * c_pos * (q.v[i] - tt.v[i]);
            dq.v[i] += e2;
            dx.v[i] = -e2;
        }
        normalize_bwd_atomic<NS, AccT>(tt, den_t, dx, d_ent + ti * dld, dim, lane);
    }
    normalize_bwd_atomic<NS, AccT>(hh, den_h, dq, d_ent + hi * dld, dim, lane);
    normalize_bwd_atomic<NS, AccT>(rr, den_r, dq, d_rel + ri * (int64_t)dim, dim, lane);
}

// rank[b] = #{n : score[b,n] < score[b, t_b], n not filtered}: the filtered link-prediction rank of the true tail
template <typename IdxT>
__global__ __launch_bounds__(256) void rank_full_kernel(const float* __restrict__ scores, int64_t n_ent, const IdxT* __restrict__ t,
                                                        const uint8_t* __restrict__ filter, int32_t* __restrict__ rank) {
    __shared__ int red[4];
    const int64_t b = blockIdx.x;
    const float* row = scores + b * n_ent;
    const float ref = row[(int64_t)t[b]];
    int cnt = 0;
    // rows that are 16-byte addressable: one dwordx4 of scores (and one dword of filter bytes) per lane
    const bool vec = (n_ent & 3) == 0 && (((uintptr_t)scores) & 15) == 0 && (filter == nullptr || (((uintptr_t)filter) & 3) == 0);
    const int64_t n4 = vec ? n_ent >> 2 : 0;
    for (int64_t q = threadIdx.x; q < n4; q += 256) {
        const float4 v = reinterpret_cast<const float4*>(row)[q];
        const uchar4 f = filter != nullptr ? reinterpret_cast<const uchar4*>(filter + b * n_ent)[q] : make_uchar4(0, 0, 0, 0);
        cnt += ((!f.x && v.x < ref) ? 1 : 0) + ((!f.y && v.y < ref) ? 1 : 0) + ((!f.z && v.z < ref) ? 1 : 0) + ((!f.w && v.w < ref) ? 1 : 0);
    }
    for (int64_t n = 4 * n4 + threadIdx.x; n < n_ent; n += 256) {
        const bool skip = filter != nullptr && filter[b * n_ent + n] != 0;
        cnt += (!skip && row[n] < ref) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) rank[b] = red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float part[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = sum64(s);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) part[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 16; ++i) tot += part[i];
        out[0] = tot;
    }
}

template <typename IdxT>
static int launch_fwd(const float* ent, int64_t eld, const float* rel, int dim, const void* h, const void* r, const void* t,
                      const void* neg_t, int64_t batch, int n_neg, float margin, float* pos, float* neg,
                      float* loss_part, hipStream_t st) {
    // several waves per triple while the batch alone cannot fill the chip with independent gather chains (K = 128 at B = 1024,
    // and the all-entity ranking form of link prediction at a few thousand queries)
    const int wpt = (batch <= 4096 && n_neg >= 128) ? 4 : (batch <= 4096 && n_neg >= 96) ? 2 : 1;
    const dim3 grid((unsigned)ceil_div(batch, 4 / wpt)), block(256);
    const int nv = (dim + 63) / 64;
#define SNAG_FWD(NV)                                                                                                           \
    do {                                                                                                                       \
        if (wpt == 4) hipLaunchKernelGGL((score_fwd_kernel<IdxT, NV, 4>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h,  \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, pos, neg, loss_part);         \
        else if (wpt == 2) hipLaunchKernelGGL((score_fwd_kernel<IdxT, NV, 2>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h, \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, pos, neg, loss_part);         \
        else hipLaunchKernelGGL((score_fwd_kernel<IdxT, NV, 1>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h,       \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, pos, neg, loss_part);         \
    } while (0)
    switch (nv) {
        case 1: SNAG_FWD(1); break;
        case 2: SNAG_FWD(2); break;
        case 3: SNAG_FWD(3); break;
        case 4: SNAG_FWD(4); break;
        case 5: case 6: SNAG_FWD(6); break;
        default: SNAG_FWD(8); break;
    }
#undef SNAG_FWD
    return check_launch("snag_score_margin_fwd");
}

template <typename IdxT, typename AccT, typename ElemT = float>
static int launch_bwd(const ElemT* ent, int64_t eld, const ElemT* rel, int dim, const void* h, const void* r, const void* t,
                      const void* neg_t, int64_t batch, int n_neg, float margin, float gscale, const float* dpos,
                      const float* dneg, AccT* d_ent, int64_t dld, AccT* d_rel, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(batch, 4)), block(256);
    const int ns = (dim + 63) / 64;
#define SNAG_BWD(NS)                                                                                           \
    hipLaunchKernelGGL((score_bwd_kernel<IdxT, NS, AccT, ElemT>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h,  \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, gscale, dpos, \
                       dneg, d_ent, dld, d_rel)
    switch (ns) {
        case 1: SNAG_BWD(1); break;
        case 2: SNAG_BWD(2); break;
        case 3: SNAG_BWD(3); break;
        case 4: SNAG_BWD(4); break;
        case 5: case 6: SNAG_BWD(6); break;
        default: SNAG_BWD(8); break;
    }
#undef SNAG_BWD
    return check_launch("snag_score_margin_bwd");
}


// ---------------------------------------------------------------------------------------------
// bf16 tables, fp32 arithmetic (BASELINE config 5: |E| = 1M, dim 256, 128 negatives).  A lane owns 16-byte chunks of
// 8 bf16 (chunk sub, sub+16, ...), so a 16-lane row still covers 256 contiguous bytes per wave-instruction.
template <int NC>
struct RowB {
    float v[8 * NC];
};

template <int NC>
__device__ __forceinline__ RowB<NC> load_row_bf16(const uint16_t* __restrict__ base, int64_t row, int64_t ld, int dim, int sub) {
    RowB<NC> r;
    const uint4* p = reinterpret_cast<const uint4*>(base + row * ld);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = sub + 16 * j;
        const uint4 q = (c * 8 < dim) ? p[c] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r.v[8 * j + 2 * e] = __uint_as_float(w[e] << 16);             // low half = element 2e
            r.v[8 * j + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); // high half = element 2e+1
        }
    }
    return r;
}

template <int NC>
__device__ __forceinline__ float sumsq_b(const RowB<NC>& r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8 * NC; ++i) s = fmaf(r.v[i], r.v[i], s);
    return s;
}

template <int NC>
__device__ __forceinline__ float score_row_b(const RowB<NC>& q, const RowB<NC>& x) {
    const float inv = 1.0f / l2_denominator(sum16(sumsq_b<NC>(x)));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8 * NC; ++i) {
        const float d = q.v[i] - x.v[i] * inv;
        s = fmaf(d, d, s);
    }
    return sum16(s);
}

// WPT waves share one triple: each takes a contiguous 1/WPT of the negatives (a multiple of 64), so that at small batches the
// chain of dependent gather round trips per wave shrinks (K = 128: four passes -> one) and four times as many waves are in flight;
// the per-triple hinge partials meet in LDS and are added in a fixed order.
template <typename IdxT, int NC, int WPT>
__global__ __launch_bounds__(256) void score_fwd_bf16_kernel(
    const uint16_t* __restrict__ ent, int64_t eld, const uint16_t* __restrict__ rel, int dim,
    const IdxT* __restrict__ h, const IdxT* __restrict__ r, const IdxT* __restrict__ t,
    const IdxT* __restrict__ neg_t, int64_t batch, int n_neg, float margin,
    float* __restrict__ pos, float* __restrict__ neg, float* __restrict__ loss_part) {
    __shared__ float part_sum[4];
    const int wave = wave_id_uniform();
    const int64_t b = (int64_t)blockIdx.x * (4 / WPT) + wave / WPT;
    const int part = wave % WPT;
    const bool live = b < batch;
    if (WPT == 1 && !live) return;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    const int64_t bb = live ? b : 0;          // WPT > 1: dead waves still reach the barrier below, on row 0
    RowB<NC> q = load_row_bf16<NC>(ent, (int64_t)h[bb], eld, dim, sub);
    const RowB<NC> rr = load_row_bf16<NC>(rel, (int64_t)r[bb], dim, dim, sub);
    const RowB<NC> tt = load_row_bf16<NC>(ent, (int64_t)t[bb], eld, dim, sub);
    {
        const float ih = 1.0f / l2_denominator(sum16(sumsq_b<NC>(q)));
        const float ir = 1.0f / l2_denominator(sum16(sumsq_b<NC>(rr)));
#pragma unroll
        for (int i = 0; i < 8 * NC; ++i) q.v[i] = q.v[i] * ih + rr.v[i] * ir;
    }
    const float s_pos = score_row_b<NC>(q, tt);
    if (lane == 0 && part == 0 && live) pos[b] = s_pos;
    float hinge = 0.f;
    const IdxT* nrow = neg_t + bb * (int64_t)n_neg;
    float* out = neg + bb * (int64_t)n_neg;
    const int per_part = ((n_neg + WPT - 1) / WPT + 31) / 32 * 32;
    const int k_lo = part * per_part, k_hi = live ? min(n_neg, k_lo + per_part) : 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
        const int my = (k0 + lane < k_hi) ? (neg_t != nullptr ? (int)nrow[k0 + lane] : k0 + lane) : 0;
        const int cnt = min(64, k_hi - k0);
        int it = 0;
        // kRounds x 4 rows in flight per wave.  A bf16 row is half the bytes of an fp32 one, and at B = 1024 a CU holds one wave per
        // SIMD, so the gather is latency-bound: 8 rounds (32 rows, 16 KB per wave) where the registers allow it
        constexpr int kRounds = (NC <= 2) ? 8 : 4;
        for (; it + 4 * kRounds <= cnt; it += 4 * kRounds) {
            RowB<NC> x[kRounds];
#pragma unroll
            for (int u = 0; u < kRounds; ++u) x[u] = load_row_bf16<NC>(ent, __shfl(my, it + 4 * u + grp), eld, dim, sub);
#pragma unroll
            for (int u = 0; u < kRounds; ++u) {
                const float s = score_row_b<NC>(q, x[u]);
                hinge += fmaxf(margin + s_pos - s, 0.f);
                if (sub == 0) out[k0 + it + 4 * u + grp] = s;
            }
        }
        for (; it < cnt; it += 4) {
            const int k = it + grp;
            const bool valid = k < cnt;
            const int idx = __shfl(my, k);
            const RowB<NC> x = load_row_bf16<NC>(ent, valid ? idx : 0, eld, dim, sub);
            const float s = score_row_b<NC>(q, x);
            if (valid) {
                hinge += fmaxf(margin + s_pos - s, 0.f);
                if (sub == 0) out[k0 + k] = s;
            }
        }
    }
    const float g1 = __shfl(hinge, 16), g2 = __shfl(hinge, 32), g3 = __shfl(hinge, 48);
    const float wsum = ((hinge + g1) + g2) + g3;
    if (WPT == 1) {
        if (loss_part != nullptr && lane == 0) loss_part[b] = wsum;
        return;
    }
    if (lane == 0) part_sum[wave] = wsum;
    __syncthreads();
    if (loss_part != nullptr && lane == 0 && part == 0 && live) {
        float tot = 0.f;
#pragma unroll
        for (int p = 0; p < WPT; ++p) tot += part_sum[wave + p];
        loss_part[b] = tot;
    }
}

template <typename IdxT>
static int launch_fwd_bf16(const uint16_t* ent, int64_t eld, const uint16_t* rel, int dim, const void* h, const void* r, const void* t,
                           const void* neg_t, int64_t batch, int n_neg, float margin, float* pos, float* neg, float* loss_part,
                           hipStream_t st) {
    // several waves per triple while the batch alone cannot fill the chip with independent gather chains
    const int wpt = (batch <= 4096 && n_neg >= 128) ? 4 : (batch <= 4096 && n_neg >= 96) ? 2 : 1;
    const dim3 grid((unsigned)ceil_div(batch, 4 / wpt)), block(256);
    const int nc = (dim + 127) / 128;
#define SNAG_FWDB(NC)                                                                                                       \
    do {                                                                                                                    \
        if (wpt == 4) hipLaunchKernelGGL((score_fwd_bf16_kernel<IdxT, NC, 4>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h, \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, pos, neg, loss_part);      \
        else if (wpt == 2) hipLaunchKernelGGL((score_fwd_bf16_kernel<IdxT, NC, 2>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h, \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, pos, neg, loss_part);      \
        else hipLaunchKernelGGL((score_fwd_bf16_kernel<IdxT, NC, 1>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h, \
                       (const IdxT*)r, (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, pos, neg, loss_part);      \
    } while (0)
    switch (nc) {
        case 1: SNAG_FWDB(1); break;
        case 2: SNAG_FWDB(2); break;
        case 3: case 4: SNAG_FWDB(4); break;
        default: SNAG_FWDB(8); break;
    }
#undef SNAG_FWDB
    return check_launch("snag_score_margin_fwd_bf16");
}

int check_score_args(const void* ent, int64_t n_ent, const void* rel, int64_t n_rel, int dim, const void* h,
                     const void* r, const void* t, const void* neg_t, int idx_bytes, int64_t batch, int n_neg) {
    SNAG_REQUIRE(ent && rel && h && r && t, SNAG_ERR_ARG, "score: null table or index pointer");
    SNAG_REQUIRE(n_ent > 0 && n_rel > 0 && dim > 0, SNAG_ERR_ARG, "score: n_ent=%ld n_rel=%ld dim=%d must be positive",
                 (long)n_ent, (long)n_rel, dim);
    SNAG_REQUIRE(batch >= 0 && n_neg >= 0, SNAG_ERR_ARG, "score: batch=%ld n_neg=%d must be non-negative", (long)batch, n_neg);
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "score: idx_bytes=%d (want 4 or 8)", idx_bytes);
    SNAG_REQUIRE(dim <= 512, SNAG_ERR_UNSUPPORTED, "score: dim=%d > 512 not supported", dim);
    SNAG_REQUIRE(n_ent < (1LL << 31), SNAG_ERR_UNSUPPORTED, "score: n_ent=%ld >= 2^31", (long)n_ent);
    return SNAG_OK;
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_score_margin_fwd(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                     const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                     int64_t batch, int n_neg, float margin, float* pos, float* neg, float* loss_part,
                                     void* stream) {
    if (batch == 0) return SNAG_OK;
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(pos && (n_neg == 0 || neg), SNAG_ERR_ARG, "score_fwd: null output pointer");
    SNAG_REQUIRE(neg_t != nullptr || n_neg <= n_ent, SNAG_ERR_ARG, "score_fwd: implicit candidates need n_neg <= n_ent");
    SNAG_REQUIRE(dim % 4 == 0 && ent_ld % 4 == 0, SNAG_ERR_UNSUPPORTED, "score_fwd: dim=%d and ent_ld=%ld must be multiples of 4 (16-byte rows)", dim, (long)ent_ld);
    SNAG_REQUIRE(ent_ld >= dim, SNAG_ERR_ARG, "score_fwd: ent_ld=%ld < dim=%d", (long)ent_ld, dim);
    if (batch == 0) return SNAG_OK;
    hipStream_t st = (hipStream_t)stream;
    return idx_bytes == 4 ? launch_fwd<int32_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, pos, neg, loss_part, st)
                          : launch_fwd<int64_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, pos, neg, loss_part, st);
}

extern "C" SNAG_API int snag_score_margin_bwd(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                     const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                     int64_t batch, int n_neg, float margin, float grad_scale, const float* dpos,
                                     const float* dneg, float* d_ent, int64_t d_ent_ld, float* d_rel, void* stream) {
    if (batch == 0) return SNAG_OK;
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(d_ent && d_rel, SNAG_ERR_ARG, "score_bwd: null gradient pointer");
    SNAG_REQUIRE(n_neg == 0 || neg_t, SNAG_ERR_ARG, "score_bwd: neg_t is null but n_neg=%d", n_neg);
    SNAG_REQUIRE(ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_ARG, "score_bwd: row stride smaller than dim=%d", dim);
    SNAG_REQUIRE((dpos != nullptr) || (dneg == nullptr), SNAG_ERR_ARG, "score_bwd: dneg given without dpos");
    SNAG_REQUIRE((dpos == nullptr) || (dneg != nullptr) || n_neg == 0, SNAG_ERR_ARG,
                 "score_bwd: explicit mode needs both dpos and dneg");
    if (batch == 0) return SNAG_OK;
    hipStream_t st = (hipStream_t)stream;
    return idx_bytes == 4
               ? launch_bwd<int32_t, float>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, d_ent, d_ent_ld, d_rel, st)
               : launch_bwd<int64_t, float>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, d_ent, d_ent_ld, d_rel, st);
}

extern "C" SNAG_API int snag_score_margin_bwd_fixed(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                    const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes, int64_t batch,
                                                    int n_neg, float margin, float grad_scale, const float* dpos, const float* dneg,
                                                    long long* acc_ent, long long* acc_rel, void* stream) {
    if (batch == 0) return SNAG_OK;
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(acc_ent && acc_rel, SNAG_ERR_ARG, "score_bwd_fixed: null accumulator");
    SNAG_REQUIRE(n_neg == 0 || neg_t, SNAG_ERR_ARG, "score_bwd_fixed: neg_t is null but n_neg=%d", n_neg);
    SNAG_REQUIRE(ent_ld >= dim, SNAG_ERR_ARG, "score_bwd_fixed: row stride smaller than dim=%d", dim);
    SNAG_REQUIRE((dpos != nullptr) || (dneg == nullptr), SNAG_ERR_ARG, "score_bwd_fixed: dneg given without dpos");
    SNAG_REQUIRE((dpos == nullptr) || (dneg != nullptr) || n_neg == 0, SNAG_ERR_ARG, "score_bwd_fixed: explicit mode needs both dpos and dneg");
    hipStream_t st = (hipStream_t)stream;
    return idx_bytes == 4
               ? launch_bwd<int32_t, long long>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, acc_ent, dim, acc_rel, st)
               : launch_bwd<int64_t, long long>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, acc_ent, dim, acc_rel, st);
}

namespace snag {
__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ acc, int64_t rows, int dim, float* __restrict__ dst,
                                                             int64_t ld, int accumulate) {
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim;
        const int c = (int)(i % dim);
        const float v = (float)((double)acc[i] * (1.0 / 4294967296.0));
        float* o = dst + r * ld + c;
        *o = accumulate ? *o + v : v;
    }
}
}  // namespace snag

extern "C" SNAG_API int snag_fixed_to_float(const long long* acc, int64_t rows, int dim, float* dst, int64_t dst_ld, int accumulate, void* stream) {
    SNAG_REQUIRE(rows >= 0 && dim >= 1 && dst_ld >= dim, SNAG_ERR_ARG, "fixed_to_float: bad sizes");
    if (rows == 0) return SNAG_OK;
    SNAG_REQUIRE(acc && dst, SNAG_ERR_ARG, "fixed_to_float: null pointer");
    const int64_t total = rows * dim;
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc, rows,
                       dim, dst, dst_ld, accumulate);
    return check_launch("snag_fixed_to_float");
}

namespace snag {
// Self-adversarial negative-sampling loss over precomputed scores (SURVEY.md par. 8 a7 "pluggable epilogue"; the in-tree hints
// are the unused flags --adv_temp / --emb_dim, config.py:53-54, and MARGIN=12; PARITY UNPINNED, oracle = its own restatement):
//   p_k = softmax_k(-adv_temp * neg[b,k])  (treated as a constant),  lower score = more plausible
//   loss_b = -log sigmoid(margin - pos[b]) - sum_k p_k log sigmoid(neg[b,k] - margin)
//   dpos[b] = sigmoid(pos - margin),  dneg[b,k] = -p_k sigmoid(margin - neg[b,k])
// One wave per triple.
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void selfadv_loss_kernel(const float* __restrict__ pos, const float* __restrict__ neg, int64_t B, int K,
                                                           float margin, float adv_temp, float* __restrict__ loss_part,
                                                           float* __restrict__ dpos, float* __restrict__ dneg) {
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float* nb = neg + b * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, -adv_temp * nb[k]);
    mx = max64(mx);
    float den = 0.f;
    for (int k = lane; k < K; k += 64) den += expf(-adv_temp * nb[k] - mx);
    den = sum64(den);
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float n = nb[k];
        const float p = expf(-adv_temp * n - mx) / den;
        acc += p * log_sigmoid(n - margin);
        if (dneg) dneg[b * K + k] = -p * sigmoid(margin - n);
    }
    acc = sum64(acc);
    if (lane == 0) {
        const float sp = pos[b];
        loss_part[b] = -log_sigmoid(margin - sp) - acc;
        if (dpos) dpos[b] = sigmoid(sp - margin);
    }
}
}  // namespace snag

extern "C" SNAG_API int snag_selfadv_loss(const float* pos, const float* neg, int64_t batch, int n_neg, float margin, float adv_temp,
                                          float* loss_part, float* dpos, float* dneg, void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_neg >= 0, SNAG_ERR_ARG, "selfadv_loss: negative size");
    if (batch == 0) return SNAG_OK;
    SNAG_REQUIRE(pos && loss_part && (neg || n_neg == 0), SNAG_ERR_ARG, "selfadv_loss: null pointer");
    hipLaunchKernelGGL(selfadv_loss_kernel, dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, (hipStream_t)stream, pos, neg, batch, n_neg, margin,
                       adv_temp, loss_part, dpos, dneg);
    return check_launch("snag_selfadv_loss");
}

extern "C" SNAG_API int snag_reduce_sum(const float* x, int64_t n, float* out, void* stream) {
    SNAG_REQUIRE(out && (x || n == 0) && n >= 0, SNAG_ERR_ARG, "reduce_sum: bad arguments (n=%ld)", (long)n);
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    return check_launch("snag_reduce_sum");
}

extern "C" SNAG_API int snag_rank_full_table(const float* scores, int64_t batch, int64_t n_ent, const void* t, int idx_bytes,
                                             const uint8_t* filter, int32_t* rank, void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_ent >= 1, SNAG_ERR_ARG, "rank_full_table: bad sizes");
    if (batch == 0) return SNAG_OK;
    SNAG_REQUIRE(scores && t && rank, SNAG_ERR_ARG, "rank_full_table: null pointer");
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "rank_full_table: idx_bytes=%d", idx_bytes);
    hipStream_t st = (hipStream_t)stream;
    if (idx_bytes == 4) hipLaunchKernelGGL((rank_full_kernel<int32_t>), dim3((unsigned)batch), dim3(256), 0, st, scores, n_ent, (const int32_t*)t, filter, rank);
    else hipLaunchKernelGGL((rank_full_kernel<int64_t>), dim3((unsigned)batch), dim3(256), 0, st, scores, n_ent, (const int64_t*)t, filter, rank);
    return check_launch("snag_rank_full_table");
}

extern "C" SNAG_API int snag_score_margin_fwd_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel,
                                                   int dim, const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                   int64_t batch, int n_neg, float margin, float* pos, float* neg, float* loss_part,
                                                   void* stream) {
    if (batch == 0) return SNAG_OK;
    SNAG_REQUIRE(ent && rel && h && r && t && pos && (n_neg == 0 || neg), SNAG_ERR_ARG, "score_fwd_bf16: null pointer");
    SNAG_REQUIRE(n_ent > 0 && n_rel > 0 && batch >= 0 && n_neg >= 0, SNAG_ERR_ARG, "score_fwd_bf16: bad sizes");
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "score_fwd_bf16: idx_bytes=%d", idx_bytes);
    SNAG_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 1024 && ent_ld % 8 == 0 && ent_ld >= dim, SNAG_ERR_UNSUPPORTED,
                 "score_fwd_bf16: dim=%d ent_ld=%ld must be multiples of 8 (16-byte chunks), dim <= 1024", dim, (long)ent_ld);
    SNAG_REQUIRE(neg_t != nullptr || n_neg <= n_ent, SNAG_ERR_ARG, "score_fwd_bf16: implicit candidates need n_neg <= n_ent");
    SNAG_REQUIRE(n_ent < (1LL << 31), SNAG_ERR_UNSUPPORTED, "score_fwd_bf16: n_ent >= 2^31");
    hipStream_t st = (hipStream_t)stream;
    return idx_bytes == 4 ? launch_fwd_bf16<int32_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, pos, neg, loss_part, st)
                          : launch_fwd_bf16<int64_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, pos, neg, loss_part, st);
}

// ---- bf16 tables (BASELINE config 5), backward in its two-launch forms: explicit upstream gradients, and the bit-reproducible
// fixed-point scatter.  fp32 arithmetic and accumulation, exactly as the fp32-table entries above.
extern "C" SNAG_API int snag_score_margin_bwd_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                                                   const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                   int64_t batch, int n_neg, float margin, float grad_scale, const float* dpos,
                                                   const float* dneg, float* d_ent, int64_t d_ent_ld, float* d_rel, void* stream) {
    if (batch == 0) return SNAG_OK;
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(d_ent && d_rel, SNAG_ERR_ARG, "score_bwd_bf16: null gradient pointer");
    SNAG_REQUIRE(n_neg == 0 || neg_t, SNAG_ERR_ARG, "score_bwd_bf16: neg_t is null but n_neg=%d", n_neg);
    SNAG_REQUIRE(ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_ARG, "score_bwd_bf16: row stride smaller than dim=%d", dim);
    SNAG_REQUIRE((dpos != nullptr) || (dneg == nullptr), SNAG_ERR_ARG, "score_bwd_bf16: dneg given without dpos");
    SNAG_REQUIRE((dpos == nullptr) || (dneg != nullptr) || n_neg == 0, SNAG_ERR_ARG, "score_bwd_bf16: explicit mode needs both dpos and dneg");
    hipStream_t st = (hipStream_t)stream;
    return idx_bytes == 4
               ? launch_bwd<int32_t, float, uint16_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, d_ent, d_ent_ld, d_rel, st)
               : launch_bwd<int64_t, float, uint16_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, d_ent, d_ent_ld, d_rel, st);
}

extern "C" SNAG_API int snag_score_margin_bwd_fixed_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                                                         const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes, int64_t batch,
                                                         int n_neg, float margin, float grad_scale, const float* dpos, const float* dneg,
                                                         long long* acc_ent, long long* acc_rel, void* stream) {
    if (batch == 0) return SNAG_OK;
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(acc_ent && acc_rel, SNAG_ERR_ARG, "score_bwd_fixed_bf16: null accumulator");
    SNAG_REQUIRE(n_neg == 0 || neg_t, SNAG_ERR_ARG, "score_bwd_fixed_bf16: neg_t is null but n_neg=%d", n_neg);
    SNAG_REQUIRE(ent_ld >= dim, SNAG_ERR_ARG, "score_bwd_fixed_bf16: row stride smaller than dim=%d", dim);
    SNAG_REQUIRE((dpos != nullptr) || (dneg == nullptr), SNAG_ERR_ARG, "score_bwd_fixed_bf16: dneg given without dpos");
    SNAG_REQUIRE((dpos == nullptr) || (dneg != nullptr) || n_neg == 0, SNAG_ERR_ARG, "score_bwd_fixed_bf16: explicit mode needs both dpos and dneg");
    hipStream_t st = (hipStream_t)stream;
    return idx_bytes == 4
               ? launch_bwd<int32_t, long long, uint16_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, acc_ent, dim, acc_rel, st)
               : launch_bwd<int64_t, long long, uint16_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, dpos, dneg, acc_ent, dim, acc_rel, st);
}
