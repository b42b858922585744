// a7, training form WITHOUT floating-point atomics: the hinge scorer's forward and backward as a per-triple pass followed by a
// per-destination pass (reference: model/MSNEA.py:110-127 + its autograd).
//
// Why: snag_score_margin_fwd_bwd scatter-adds (K+3) gradient rows per triple with fp32 atomics; those execute at the memory side
// at ~1.3 TB/s chip-wide whatever the schedule (MI355X_MICROARCH.md "Global float atomics"): 14 us of the 18 us kernel at the
// DB15K shape, and their sums depend on the arrival order.  Plain stores run 4-5x that rate, so (guide, "Scatter / gather /
// embedding"): store every contribution once, then sum per destination through an inverted index.  Here the contributions
// collapse further, because the backward of F.normalize annihilates the component along the row itself:
//
//   s(x) = |q - x^|^2,  q = h^ + r^          d s / d x^ = -2 (q - x^)  ==  -2 q   (mod x^)
//   grad x_e = nbwd(x^_e, G_e),   G_e = sum_{(b,k): neg[b,k]=e} 2 a_bk q_b  -  sum_{b: t_b=e} 2 c_b q_b  +  sum_{b: h_b=e} dq_b
//   grad r_j = nbwd(r^_j, sum_{b: r_b=j} dq_b)
//   a_bk = grad_scale * [margin + s_pos,b - s_bk > 0],  c_b = sum_k a_bk,  dq_b = 2 c_b (q_b - t^_b) - 2 sum_k a_bk (q_b - x^_bk)
//
// so every contribution is  weight * (one of the 2B rows q_b, dq_b):
//   pass A (one wave per triple, the forward kernel's 16-lane-row x float4 gather): scores, losses, q_b and dq_b (2 x 512 B
//          coalesced stores per triple) and one 8-byte (row, weight) entry per touched destination, slotted by a returning integer
//          atomic on the destination's counter (fixed-capacity bucket, overflow list behind it);
//   pass B (one 16-lane group per destination entity / relation): gathers its entries' rows (1 MB table of q / dq rows:
//          L2-resident), sums them in 64-bit FIXED POINT -- order-free, so the result is bit-reproducible although the slot order
//          is not --, applies the normalisation backward once and WRITES the gradient row (no zero-fill of the target, no
//          read-modify-write), then clears its counter for the next launch.
#include <algorithm>

#include "seg_rows.hip.h"

namespace snag {

constexpr int kSegFast = 16;                // destinations with at most this many entries take the sorted fp32 path of pass B

struct SegHeader {                          // first 64 bytes of the workspace (zeroed once by the caller)
    unsigned ovf_cnt[2];                    // overflow entries of the even / odd launch
    unsigned phase;                         // which of the two the NEXT pass A uses (flipped by pass B)
    unsigned cur;                           // which one the CURRENT launch uses (written by pass A, read by pass B)
    unsigned pad[12];
};
struct SegOverflow {
    int dest, src;
    float w;
    int pad;
};

struct SegWork {
    SegHeader* hdr;
    unsigned* cnt;          // [n_dest]
    SegEntry* list;         // [n_dest, kSegCap]
    SegOverflow* ovf;       // [ovf_cap]
    float* rows;            // [2 * batch, dim]: q_b then dq_b
    unsigned ovf_cap;
    int cap;                // list slots per destination (power of two >= 16, sized from the batch: seg_capacity)
};

static size_t seg_align(size_t x) { return (x + 255) / 256 * 256; }

// slots per destination: four times the mean number of entries a destination receives, as a power of two in [16, 1024]
// (uniformly drawn negatives are Poisson around the mean; hub heads / tails of skewed data spill into the overflow list)
static int seg_capacity(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg) {
    const int64_t mean4 = 4 * (batch * (n_neg + 3)) / std::max<int64_t>(1, n_ent + n_rel);
    int cap = 16;
    while (cap < mean4 && cap < 1024) cap *= 2;
    return cap;
}

static size_t seg_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    const size_t n_dest = (size_t)(n_ent + n_rel);
    const size_t kSegCap = (size_t)seg_capacity(n_ent, n_rel, batch, n_neg);
    return seg_align(sizeof(SegHeader)) + seg_align(n_dest * 4) + seg_align(n_dest * kSegCap * sizeof(SegEntry)) +
           seg_align((size_t)batch * (n_neg + 3) * sizeof(SegOverflow)) + seg_align((size_t)2 * batch * dim * 4);
}

static SegWork seg_carve(void* base, int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    const size_t n_dest = (size_t)(n_ent + n_rel);
    const size_t kSegCap = (size_t)seg_capacity(n_ent, n_rel, batch, n_neg);
    char* p = (char*)base;
    SegWork w;
    w.cap = (int)kSegCap;
    w.hdr = (SegHeader*)p; p += seg_align(sizeof(SegHeader));
    w.cnt = (unsigned*)p; p += seg_align(n_dest * 4);
    w.list = (SegEntry*)p; p += seg_align(n_dest * kSegCap * sizeof(SegEntry));
    w.ovf = (SegOverflow*)p; p += seg_align((size_t)batch * (n_neg + 3) * sizeof(SegOverflow));
    w.rows = (float*)p;
    w.ovf_cap = (unsigned)(batch * (n_neg + 3));
    return w;
}

// A slot in dest's bucket is RESERVED early (returning integer atomic, issued before the row gathers so that its round trip hides
// behind them) and FILLED once the weight is known; a reservation past the bucket's capacity goes to the overflow list.
__device__ __forceinline__ unsigned seg_reserve(const SegWork& w, int dest) { return atomicAdd(w.cnt + dest, 1u); }
__device__ __forceinline__ void seg_fill(const SegWork& w, unsigned which, int dest, unsigned pos, int src, float weight) {
    if (pos < (unsigned)w.cap) {
        w.list[(int64_t)dest * w.cap + pos] = SegEntry{src, weight};
    } else {
        const unsigned o = atomicAdd(&w.hdr->ovf_cnt[which], 1u);
        if (o < w.ovf_cap) w.ovf[o] = SegOverflow{dest, src, weight, 0};
    }
}

// ---- pass A --------------------------------------------------------------------------------------------------------------------
template <typename IdxT, int NV>
__global__ __launch_bounds__(256) void score_seg_a_kernel(
    const float* __restrict__ ent, int64_t eld, const float* __restrict__ rel, int dim, int n_ent,
    const IdxT* __restrict__ h, const IdxT* __restrict__ r, const IdxT* __restrict__ t, const IdxT* __restrict__ neg_t,
    int64_t batch, int n_neg, float margin, float gscale,
    float* __restrict__ pos, float* __restrict__ neg, float* __restrict__ loss_part, SegWork w) {
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    const bool live = b < batch;
    const unsigned which = w.hdr->phase & 1u;           // stable during pass A (only pass B flips it)
    float hinge = 0.f;
    if (live) {
        const int64_t hi = (int64_t)h[b], ri = (int64_t)r[b], ti = (int64_t)t[b];
        const IdxT* nrow = neg_t + b * (int64_t)n_neg;
        // slots for the positive tail, the head and the relation (lanes 0, 1, 2); the negatives' follow per id block below
        const int dest3 = lane == 0 ? (int)ti : (lane == 1 ? (int)hi : n_ent + (int)ri);
        unsigned pos3 = 0u;
        if (lane < 3) pos3 = seg_reserve(w, dest3);
        // (the first id block of negatives travels with h / r / t, and its first 32 rows with the positive's three rows: two round trips in
        // front of the arithmetic instead of four)
        const int cnt0 = min(64, n_neg);
        const int my0 = (lane < cnt0) ? (int)nrow[lane] : 0;
        Row4<NV> q = ld_row4<NV>(ent, hi, eld, dim, sub);
        const Row4<NV> rr = ld_row4<NV>(rel, ri, dim, dim, sub);
        Row4<NV> tt = ld_row4<NV>(ent, ti, eld, dim, sub);
        Row4<NV> x0[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 4 * u + grp;
            const int idx = __shfl(my0, k & 63);
            x0[u] = ld_row4<NV>(ent, (k < cnt0) ? idx : 0, eld, dim, sub);
        }
        {
            const float ih = 1.0f / l2_denominator(sum16(sumsq4<NV>(q)));
            const float ir = 1.0f / l2_denominator(sum16(sumsq4<NV>(rr)));
            const float it = 1.0f / l2_denominator(sum16(sumsq4<NV>(tt)));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                q.v[j].x = q.v[j].x * ih + rr.v[j].x * ir; q.v[j].y = q.v[j].y * ih + rr.v[j].y * ir;
                q.v[j].z = q.v[j].z * ih + rr.v[j].z * ir; q.v[j].w = q.v[j].w * ih + rr.v[j].w * ir;
                tt.v[j].x *= it; tt.v[j].y *= it; tt.v[j].z *= it; tt.v[j].w *= it;
            }
        }
        // e = q - t^ ; s_pos = |e|^2
        Row4<NV> acc;                    // this 16-lane group's share of  sum_k a_k (q - x^_k)
        float s_pos = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            acc.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            float d;
            d = q.v[j].x - tt.v[j].x; s_pos = fmaf(d, d, s_pos); d = q.v[j].y - tt.v[j].y; s_pos = fmaf(d, d, s_pos);
            d = q.v[j].z - tt.v[j].z; s_pos = fmaf(d, d, s_pos); d = q.v[j].w - tt.v[j].w; s_pos = fmaf(d, d, s_pos);
        }
        s_pos = sum16(s_pos);
        if (lane == 0) pos[b] = s_pos;
        float c_pos = 0.f;
        for (int k0 = 0; k0 < n_neg; k0 += 64) {
            const int cnt = min(64, n_neg - k0);
            const int my = k0 == 0 ? my0 : ((lane < cnt) ? (int)nrow[k0 + lane] : 0);
            unsigned my_pos = 0u;
            if (lane < cnt) my_pos = seg_reserve(w, my);    // in flight while the rows are gathered
            float my_s = 0.f;                               // lane k: score of negative k0 + k
            for (int it0 = 0; it0 < cnt; it0 += 32) {       // 8 rounds x 4 rows gathered before the first is used
                Row4<NV> x[8];
                if (k0 == 0 && it0 == 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = x0[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = it0 + 4 * u + grp;
                        const int idx = __shfl(my, k & 63);
                        x[u] = ld_row4<NV>(ent, (k < cnt) ? idx : 0, eld, dim, sub);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = it0 + 4 * u + grp;
                    const bool valid = k < cnt;
                    const float inv = 1.0f / l2_denominator(sum16(sumsq4<NV>(x[u])));
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {          // x[u] <- q - x^
                        x[u].v[j].x = q.v[j].x - x[u].v[j].x * inv; s = fmaf(x[u].v[j].x, x[u].v[j].x, s);
                        x[u].v[j].y = q.v[j].y - x[u].v[j].y * inv; s = fmaf(x[u].v[j].y, x[u].v[j].y, s);
                        x[u].v[j].z = q.v[j].z - x[u].v[j].z * inv; s = fmaf(x[u].v[j].z, x[u].v[j].z, s);
                        x[u].v[j].w = q.v[j].w - x[u].v[j].w * inv; s = fmaf(x[u].v[j].w, x[u].v[j].w, s);
                    }
                    s = sum16(s);
                    const float viol = margin + s_pos - s;
                    const float a = (valid && viol > 0.f) ? gscale : 0.f;      // uniform inside the 16-lane group
                    hinge += valid ? fmaxf(viol, 0.f) : 0.f;
                    c_pos += a;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        acc.v[j].x = fmaf(a, x[u].v[j].x, acc.v[j].x); acc.v[j].y = fmaf(a, x[u].v[j].y, acc.v[j].y);
                        acc.v[j].z = fmaf(a, x[u].v[j].z, acc.v[j].z); acc.v[j].w = fmaf(a, x[u].v[j].w, acc.v[j].w);
                    }
                    // hand the score to the lane that owns negative k (lane k of this id block): group g of round u holds k = it0+4u+g
                    const float sk = __shfl(s, (lane & 3) * 16);
                    if (((lane - it0) >> 2) == u && lane >= it0 && lane < it0 + 32) my_s = sk;
                }
            }
            // lane k: score out, its negative's list entry filled in (weight 2 a_k; zero for an inactive hinge)
            if (lane < cnt) {
                neg[b * (int64_t)n_neg + k0 + lane] = my_s;
                seg_fill(w, which, my, my_pos, (int)b, (margin + s_pos - my_s > 0.f) ? 2.f * gscale : 0.f);
            }
        }
        // the four groups' partial sums, combined in fixed order: every group then holds the whole triple's values
        {
            const float h1 = __shfl_xor(hinge, 16), c1 = __shfl_xor(c_pos, 16);
            hinge += h1; c_pos += c1;
            const float h2 = __shfl_xor(hinge, 32), c2 = __shfl_xor(c_pos, 32);
            hinge += h2; c_pos += c2;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float4& v = acc.v[j];
                v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
                v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
            }
        }
        // rows[b] = q_b ; rows[batch + b] = dq_b = 2 c (q - t^) - 2 sum_k a_k (q - x^_k)   (group 0 / group 1 store one each)
        if (grp == 0) st_row4<NV>(w.rows, b, dim, dim, sub, q);
        if (grp == 1) {
            Row4<NV> dq;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                dq.v[j].x = 2.f * (c_pos * (q.v[j].x - tt.v[j].x) - acc.v[j].x); dq.v[j].y = 2.f * (c_pos * (q.v[j].y - tt.v[j].y) - acc.v[j].y);
                dq.v[j].z = 2.f * (c_pos * (q.v[j].z - tt.v[j].z) - acc.v[j].z); dq.v[j].w = 2.f * (c_pos * (q.v[j].w - tt.v[j].w) - acc.v[j].w);
            }
            st_row4<NV>(w.rows, batch + b, dim, dim, sub, dq);
        }
        if (lane < 3) seg_fill(w, which, dest3, pos3, lane == 0 ? (int)b : (int)(batch + b), lane == 0 ? -2.f * c_pos : 1.f);
    }
    // per-triple loss: summed in fixed order by one extra workgroup of pass B (a launch boundary already separates the two passes: no
    // last-block election here -- that cost this pass a drain of its stores, a workgroup barrier and two dependent atomic round trips)
    if (live && lane == 0) loss_part[b] = hinge;
    if (blockIdx.x == 0 && threadIdx.x == 0) w.hdr->cur = which;          // pass B of this launch reads the overflow list `which`
}

// ---- pass B --------------------------------------------------------------------------------------------------------------------
// Two dependent memory round trips per destination: (1) its counter, its first 16 list entries (one coalesced 128-byte read: lane
// i of the group takes entry i) and its own table row; (2) the rows its entries name.  Up to kSegFast entries are put in a
// canonical order (by (row, weight) key, found with 16 in-group shuffles) and summed in fp32 in that order -- the slot order the
// atomics of pass A produced varies from run to run, the sorted order does not.  Larger destinations (hubs, big batches) sum in
// 64-bit fixed point instead, which is order-free.
// LN: the entity table IS the output of a LayerNorm (the Mformer's read-out rows, model/SNAG_tools.py:218-222) and the caller wants the
// gradient carried through it here: the group that has just formed its entity's gradient row g applies the LayerNorm backward to it in
// registers -- dT = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), dxhat = g gamma, the arithmetic of ln_bwd_kernel -- writes dT (and
// dT * keep * scale), and the workgroup's 16 rows leave their sums of g xhat / g / dT(m) in part[block][3][dim] for the slab reduction the
// fusion backward runs anyway.  d_ent is then not written at all.
template <int NV, bool LN>
__global__ __launch_bounds__(256) void score_seg_b_kernel(const float* __restrict__ ent, int64_t eld, const float* __restrict__ rel, int dim,
                                                          int n_ent, int n_dest, float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel,
                                                          SegWork w, int write_untouched, const float* __restrict__ loss_part, int64_t batch,
                                                          float* __restrict__ loss, snag_ln_bwd_tail tail) {
    if (blockIdx.x == gridDim.x - 1) {
        // the extra workgroup: loss = sum of the per-triple losses pass A left, in one fixed order (thread-strided partial sums, a wave
        // reduction each, the four wave sums added in wave order)
        __shared__ float part[4];
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < batch; i += 256) s += loss_part[i];
        s = sum64(s);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0 && loss != nullptr) loss[0] = ((part[0] + part[1]) + part[2]) + part[3];
        return;
    }
    const int lane = threadIdx.x & 63, sub = lane & 15, gbase = lane & 48;
    const int dest_raw = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4);
    const unsigned which = w.hdr->cur & 1u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {               // the other overflow list belongs to the previous launch: clear it, hand it to the next
        w.hdr->ovf_cnt[which ^ 1u] = 0u;
        w.hdr->phase = which ^ 1u;
    }
    const bool have = dest_raw < n_dest;
    const int dest = have ? dest_raw : 0;                      // idle groups shadow destination 0 without storing (shuffles need all lanes)
    const bool is_rel = dest >= n_ent;
    // round trip 1
    const unsigned n = have ? w.cnt[dest] : 0u;
    const SegEntry mine = w.list[(int64_t)dest * w.cap + sub];
    const float* xrow = is_rel ? rel + (int64_t)(dest - n_ent) * dim : ent + (int64_t)dest * eld;      // (one load sequence, not two branches)
    Row4<NV> x = ld_row4<NV>(xrow, 0, 0, dim, sub);
    // (LN: everything the LayerNorm backward of this row needs is requested in the same round trip)
    [[maybe_unused]] const bool row_ok = have && !is_rel;
    [[maybe_unused]] Row4<NV> tr, gam;
    [[maybe_unused]] float ln_mean = 0.f, ln_rstd = 0.f;
    [[maybe_unused]] uchar4 kk[NV];
    if constexpr (LN) {
        const int64_t lr = row_ok ? dest : 0;
        tr = ld_row4<NV>(tail.t, lr, dim, dim, sub);
        gam = ld_row4<NV>(tail.gamma, 0, 0, dim, sub);
        ln_mean = tail.stats[2 * lr];
        ln_rstd = tail.stats[2 * lr + 1];
        if (tail.dtm != nullptr) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = (sub + 16 * j) * 4;
                kk[j] = *reinterpret_cast<const uchar4*>(tail.keep + lr * tail.keep_ld + (c < dim ? c : 0));
            }
        }
    }
    float* out = is_rel ? d_rel + (int64_t)(dest - n_ent) * dim : d_ent + (int64_t)dest * dld;
    Row4<NV> g;
#pragma unroll
    for (int j = 0; j < NV; ++j) g.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have && n > 0u && sub == 0) w.cnt[dest] = 0u;          // ready for the next launch
    const int nf = (int)min(n, (unsigned)kSegFast);
    // canonical order of the first nf entries: rank = number of entries with a smaller (src, weight bits, slot) key
    const long long key = ((long long)mine.src << 32) | (unsigned)__float_as_int(mine.w);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kSegFast; ++j) {
        const long long kj = __shfl(key, gbase + j);
        rank += (j < nf && (kj < key || (kj == key && j < sub))) ? 1 : 0;
    }
    const bool fast = n <= (unsigned)kSegFast;
    if (__ballot(fast && n > 0u) != 0ull) {
        // sorted fp32 sums, four rows gathered per step (the whole wave walks to the largest count among its four groups)
        int wave_max = nf;
        wave_max = max(wave_max, __shfl_xor(wave_max, 16));
        wave_max = max(wave_max, __shfl_xor(wave_max, 32));
        for (int r0 = 0; r0 < wave_max; r0 += 4) {
            Row4<NV> y[4];
            float wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // the lane of this group holding rank r0 + u
                const unsigned long long m = __ballot(sub < nf && rank == r0 + u) >> gbase;
                const int src_lane = gbase + (int)__builtin_ctzll(m | 0x10000ull);
                const int src = __shfl(mine.src, src_lane);
                wt[u] = (fast && r0 + u < nf) ? __shfl(mine.w, src_lane) : 0.f;
                y[u] = ld_row4<NV>(w.rows, (fast && r0 + u < nf) ? src : 0, dim, dim, sub);
                if (!(fast && r0 + u < nf)) wt[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    g.v[j].x = fmaf(wt[u], y[u].v[j].x, g.v[j].x); g.v[j].y = fmaf(wt[u], y[u].v[j].y, g.v[j].y);
                    g.v[j].z = fmaf(wt[u], y[u].v[j].z, g.v[j].z); g.v[j].w = fmaf(wt[u], y[u].v[j].w, g.v[j].w);
                }
        }
    }
    if (!fast) {
        // big destination: order-free 64-bit fixed-point sums over its whole bucket and its share of the overflow list
        long long acc[4 * NV];
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) acc[i] = 0;
        fix_add_run<NV>(acc, w.list + (int64_t)dest * w.cap, min(n, (unsigned)w.cap), w.rows, dim, sub, gbase);
        if (n > (unsigned)w.cap) {                            // hub: its surplus sits somewhere in the overflow list
            const unsigned n_ovf = min(w.hdr->ovf_cnt[which], w.ovf_cap);
            for (unsigned i = 0; i < n_ovf; ++i) {
                const SegOverflow e = w.ovf[i];
                if (e.dest == dest) fix_add_row<NV>(acc, ld_row4<NV>(w.rows, e.src, dim, dim, sub), e.w);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            g.v[j].x = (float)((double)acc[4 * j + 0] * (1.0 / kFix)); g.v[j].y = (float)((double)acc[4 * j + 1] * (1.0 / kFix));
            g.v[j].z = (float)((double)acc[4 * j + 2] * (1.0 / kFix)); g.v[j].w = (float)((double)acc[4 * j + 3] * (1.0 / kFix));
        }
    }
    normalize_bwd4<NV>(x, g);
    if constexpr (!LN) {
        if (have && (n > 0u || write_untouched)) st_row4<NV>(out, 0, 0, dim, sub, g);      // n == 0: g is exactly zero
    } else {
        if (have && is_rel && (n > 0u || write_untouched)) st_row4<NV>(out, 0, 0, dim, sub, g);
        // ---- LayerNorm backward of the entity rows (every row: an untouched entity has g = 0 and contributes zeros) ----
        __shared__ float red[4][3][64 * NV];
        const float mean = ln_mean, rstd = ln_rstd;
        Row4<NV> xh, dxh, o;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool in = (sub + 16 * j) * 4 < dim;
            const float tv[4] = {tr.v[j].x, tr.v[j].y, tr.v[j].z, tr.v[j].w}, gv[4] = {g.v[j].x, g.v[j].y, g.v[j].z, g.v[j].w};
            const float ga[4] = {gam.v[j].x, gam.v[j].y, gam.v[j].z, gam.v[j].w};
            float xv[4], dv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[e] = in ? (tv[e] - mean) * rstd : 0.f;
                dv[e] = gv[e] * ga[e];
                s1 += dv[e];
                s2 = fmaf(dv[e], xv[e], s2);
            }
            xh.v[j] = make_float4(xv[0], xv[1], xv[2], xv[3]);
            dxh.v[j] = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
        s1 = sum16(s1) / (float)dim;
        s2 = sum16(s2) / (float)dim;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            o.v[j].x = rstd * (dxh.v[j].x - s1 - xh.v[j].x * s2); o.v[j].y = rstd * (dxh.v[j].y - s1 - xh.v[j].y * s2);
            o.v[j].z = rstd * (dxh.v[j].z - s1 - xh.v[j].z * s2); o.v[j].w = rstd * (dxh.v[j].w - s1 - xh.v[j].w * s2);
        }
        if (row_ok) st_row4<NV>(tail.dt, dest, dim, dim, sub, o);
        Row4<NV> z = o;
        if (tail.dtm != nullptr) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const uchar4 k = kk[j];
                z.v[j].x = k.x ? o.v[j].x * tail.keep_scale : 0.f; z.v[j].y = k.y ? o.v[j].y * tail.keep_scale : 0.f;
                z.v[j].z = k.z ? o.v[j].z * tail.keep_scale : 0.f; z.v[j].w = k.w ? o.v[j].w * tail.keep_scale : 0.f;
            }
            if (row_ok) st_row4<NV>(tail.dtm, dest, dim, dim, sub, z);
        }
        // the workgroup's 16 rows: column sums of g xhat (dgamma), g (dbeta) and z (the bias gradient of the dense in front).  A wave's four
        // rows meet in registers (two butterfly steps across its 16-lane groups), the four waves in LDS: a fixed order
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (sub + 16 * j) * 4;
            float v[3][4] = {{g.v[j].x * xh.v[j].x, g.v[j].y * xh.v[j].y, g.v[j].z * xh.v[j].z, g.v[j].w * xh.v[j].w},
                             {g.v[j].x, g.v[j].y, g.v[j].z, g.v[j].w},
                             {z.v[j].x, z.v[j].y, z.v[j].z, z.v[j].w}};
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = row_ok ? v[q][e] : 0.f;
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                    v[q][e] = t;
                }
            if (lane < 16 && c < dim) {
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(&red[wv][q][c]) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
            }
        }
        __syncthreads();
        const int first_row = (int)blockIdx.x * 16;
        if (first_row < n_ent) {
            for (int i = threadIdx.x; i < 3 * dim; i += 256) {
                const int which = i / dim, c = i % dim;
                tail.part[((int64_t)blockIdx.x * 3 + which) * dim + c] = ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
            }
        }
    }
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int64_t snag_score_seg_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    if (n_ent < 1 || n_rel < 1 || batch < 0 || n_neg < 0 || dim < 1) return 0;
    return (int64_t)seg_work_bytes(n_ent, n_rel, batch, n_neg, dim);
}

static int score_seg_launch(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                            const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                            int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                            float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld,
                            float* d_rel, int write_untouched, void* work, int64_t work_bytes, const snag_ln_bwd_tail* tail, void* stream) {
    SNAG_REQUIRE(ent && rel && (d_ent || tail) && d_rel && work && (batch == 0 || loss_part), SNAG_ERR_ARG,
                 "score_fwd_bwd_seg: null table / gradient / workspace / loss_part");
    if (tail != nullptr) {
        SNAG_REQUIRE(tail->t && tail->stats && tail->gamma && tail->dt && tail->part, SNAG_ERR_ARG, "score_fwd_bwd_seg_ln: null t / stats / gamma / dt / part");
        SNAG_REQUIRE((tail->dtm == nullptr) == (tail->keep == nullptr) && (tail->keep == nullptr || (tail->keep_ld >= dim && tail->keep_ld % 4 == 0 && ((uintptr_t)tail->keep & 3) == 0)),
                     SNAG_ERR_ARG, "score_fwd_bwd_seg_ln: dtm and keep come together; the mask rows must be 4-byte addressable");
        SNAG_REQUIRE(ent_ld == dim && write_untouched, SNAG_ERR_UNSUPPORTED, "score_fwd_bwd_seg_ln: the table must be the dense [n_ent, dim] LayerNorm output, every row written");
        SNAG_REQUIRE((((uintptr_t)tail->t | (uintptr_t)tail->gamma | (uintptr_t)tail->dt | (uintptr_t)tail->dtm) & 15) == 0, SNAG_ERR_UNSUPPORTED,
                     "score_fwd_bwd_seg_ln: t / gamma / dt / dtm must be 16-byte aligned");
        if (d_ent == nullptr) { d_ent = tail->dt; d_ent_ld = dim; }           // (never written in this form; keeps the checks below uniform)
    }
    SNAG_REQUIRE(n_ent > 0 && n_rel > 0 && dim > 0 && batch >= 0 && n_neg >= 0, SNAG_ERR_ARG, "score_fwd_bwd_seg: bad sizes");
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "score_fwd_bwd_seg: idx_bytes=%d (want 4 or 8)", idx_bytes);
    SNAG_REQUIRE(dim <= 512 && dim % 4 == 0 && ent_ld % 4 == 0 && d_ent_ld % 4 == 0 && ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_UNSUPPORTED,
                 "score_fwd_bwd_seg: dim=%d (<= 512) and the row strides must be multiples of 4 (16-byte rows)", dim);
    SNAG_REQUIRE(n_ent + n_rel < (1LL << 27), SNAG_ERR_UNSUPPORTED, "score_fwd_bwd_seg: n_ent + n_rel >= 2^27");
    SNAG_REQUIRE(work_bytes >= (int64_t)seg_work_bytes(n_ent, n_rel, batch, n_neg, dim), SNAG_ERR_ARG, "score_fwd_bwd_seg: workspace holds %ld bytes, need %ld",
                 (long)work_bytes, (long)seg_work_bytes(n_ent, n_rel, batch, n_neg, dim));
    SNAG_REQUIRE(((uintptr_t)work & 255) == 0 && ((uintptr_t)ent & 15) == 0 && ((uintptr_t)rel & 15) == 0 && ((uintptr_t)d_ent & 15) == 0 && ((uintptr_t)d_rel & 15) == 0,
                 SNAG_ERR_UNSUPPORTED, "score_fwd_bwd_seg: tables and gradients must be 16-byte aligned, the workspace 256-byte aligned");
    SNAG_REQUIRE(batch == 0 || (h && r && t && pos && (n_neg == 0 || (neg && neg_t))), SNAG_ERR_ARG, "score_fwd_bwd_seg: null index or score pointer");
    hipStream_t st = (hipStream_t)stream;
    const SegWork w = seg_carve(work, n_ent, n_rel, batch, n_neg, dim);
    const int nv = (dim + 63) / 64;
    const int n_dest = (int)(n_ent + n_rel);
    if (batch > 0) {
        const dim3 grid((unsigned)ceil_div(batch, 4)), block(256);
#define SNAG_SEG_A(IDX, NV)                                                                                                         \
        hipLaunchKernelGGL((score_seg_a_kernel<IDX, NV>), grid, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, (const IDX*)h, (const IDX*)r, \
                           (const IDX*)t, (const IDX*)neg_t, batch, n_neg, margin, grad_scale, pos, neg, loss_part, w)
#define SNAG_SEG_AI(NV) do { if (idx_bytes == 4) SNAG_SEG_A(int32_t, NV); else SNAG_SEG_A(int64_t, NV); } while (0)
        switch (nv) {
            case 1: SNAG_SEG_AI(1); break;
            case 2: SNAG_SEG_AI(2); break;
            case 3: SNAG_SEG_AI(3); break;
            case 4: SNAG_SEG_AI(4); break;
            case 5: case 6: SNAG_SEG_AI(6); break;
            default: SNAG_SEG_AI(8); break;
        }
#undef SNAG_SEG_AI
#undef SNAG_SEG_A
        int rc = check_launch("snag_score_margin_fwd_bwd_seg (pass A)");
        if (rc) return rc;
    }
    // (+ 1: the workgroup that sums the per-triple losses; with an empty batch it writes loss = 0)
    const dim3 gridb((unsigned)ceil_div((int64_t)n_dest * 16, 256) + 1), block(256);
    const snag_ln_bwd_tail tl = tail ? *tail : snag_ln_bwd_tail{};
#define SNAG_SEG_B(NV) do { if (tail) hipLaunchKernelGGL((score_seg_b_kernel<NV, true>), gridb, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, n_dest, d_ent, d_ent_ld, d_rel, w, \
                                                          write_untouched, (const float*)loss_part, batch, loss, tl); \
                            else hipLaunchKernelGGL((score_seg_b_kernel<NV, false>), gridb, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, n_dest, d_ent, d_ent_ld, d_rel, w, \
                                                    write_untouched, (const float*)loss_part, batch, loss, tl); } while (0)
    switch (nv) {
        case 1: SNAG_SEG_B(1); break;
        case 2: SNAG_SEG_B(2); break;
        case 3: SNAG_SEG_B(3); break;
        case 4: SNAG_SEG_B(4); break;
        case 5: case 6: SNAG_SEG_B(6); break;
        default: SNAG_SEG_B(8); break;
    }
#undef SNAG_SEG_B
    return check_launch("snag_score_margin_fwd_bwd_seg (pass B)");
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_seg(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                      const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                      int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                      float* loss_part, float* loss, uint32_t* ticket, float* d_ent, int64_t d_ent_ld,
                                                      float* d_rel, int write_untouched, void* work, int64_t work_bytes, void* stream) {
    (void)ticket;          // (the loss is summed by pass B since round 3: no last-block election, the ticket words are not touched)
    return score_seg_launch(ent, ent_ld, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg, margin, grad_scale, pos, neg, loss_part, loss, d_ent, d_ent_ld,
                            d_rel, write_untouched, work, work_bytes, nullptr, stream);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_seg_ln(const float* ent, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                         const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                         int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                         float* loss_part, float* loss, float* d_rel, void* work, int64_t work_bytes,
                                                         const snag_ln_bwd_tail* tail, void* stream) {
    SNAG_REQUIRE(tail != nullptr, SNAG_ERR_ARG, "score_fwd_bwd_seg_ln: null tail");
    return score_seg_launch(ent, dim, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg, margin, grad_scale, pos, neg, loss_part, loss, nullptr, dim,
                            d_rel, 1, work, work_bytes, tail, stream);
}

