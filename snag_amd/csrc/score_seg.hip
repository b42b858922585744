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

#include "common.hip.h"

namespace snag {

constexpr int kSegFast = 16;                // destinations with at most this many entries take the sorted fp32 path of pass B
constexpr double kFix = 4294967296.0;       // 32 fractional bits: resolution 2.3e-10, |sum| < 2^31

struct SegHeader {                          // first 64 bytes of the workspace (zeroed once by the caller)
    unsigned ovf_cnt[2];                    // overflow entries of the even / odd launch
    unsigned phase;                         // which of the two the NEXT pass A uses (flipped by pass B)
    unsigned cur;                           // which one the CURRENT launch uses (written by pass A, read by pass B)
    unsigned n_touched[2];                  // touched-destination form: length of the touched list of the even / odd launch
    unsigned pad[10];
};
struct SegEntry {                           // one contribution: weight * rows[src]
    int src;
    float w;
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
    // touched-destination form (tables far larger than a batch's footprint: BASELINE configs[4], 1M entities)
    unsigned* touched;                  // [batch * (n_neg + 3)]: every destination that received an entry, appended by its FIRST entry
    unsigned long long* hkeys[2];       // [hmask + 1] each: keys (dest, chunk) + 1 of the overflow hash of the even / odd launch, 0 = free
    SegEntry* hbuckets;                 // [hmask + 1, kHashChunk]: entries cap + 64 j .. of destination dest live in the slot of key (dest, j)
    unsigned hmask;
};
constexpr int kHashChunk = 64;

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

// ---- touched-destination form: workspace ------------------------------------------------------------------------------------
// Entries beyond a destination's inline bucket (hot rows of skewed data) go to an open-addressed hash of 64-entry chunks keyed
// (dest, chunk): insertion is one compare-and-swap per probe, no thread ever waits for another, and entry pos of a destination has
// exactly one home, (chunk (pos - cap) / 64, slot (pos - cap) % 64) -- a plain store.  Chunks <= E / 64 + E / cap for E entries.
static unsigned touched_hash_slots(int64_t batch, int n_neg) {
    const int64_t e = std::max<int64_t>(1, batch * (n_neg + 3));
    const int64_t need = 2 * (e / kHashChunk + e / 16 + 1);
    unsigned h = 1024;
    while ((int64_t)h < need) h *= 2;
    return h;
}
static size_t touched_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    const size_t n_dest = (size_t)(n_ent + n_rel), e = (size_t)batch * (n_neg + 3), hs = touched_hash_slots(batch, n_neg);
    return seg_align(sizeof(SegHeader)) + seg_align(n_dest * 4) + seg_align(n_dest * 16 * sizeof(SegEntry)) + seg_align(e * 4) + 2 * seg_align(hs * 8) +
           seg_align(hs * kHashChunk * sizeof(SegEntry)) + seg_align((size_t)2 * batch * dim * 4);
}
static SegWork touched_carve(void* base, int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    const size_t n_dest = (size_t)(n_ent + n_rel), e = (size_t)batch * (n_neg + 3), hs = touched_hash_slots(batch, n_neg);
    char* p = (char*)base;
    SegWork w{};
    w.cap = 16;
    w.hdr = (SegHeader*)p; p += seg_align(sizeof(SegHeader));
    w.cnt = (unsigned*)p; p += seg_align(n_dest * 4);
    w.list = (SegEntry*)p; p += seg_align(n_dest * 16 * sizeof(SegEntry));
    w.touched = (unsigned*)p; p += seg_align(e * 4);
    w.hkeys[0] = (unsigned long long*)p; p += seg_align(hs * 8);
    w.hkeys[1] = (unsigned long long*)p; p += seg_align(hs * 8);
    w.hbuckets = (SegEntry*)p; p += seg_align(hs * kHashChunk * sizeof(SegEntry));
    w.rows = (float*)p;
    w.hmask = (unsigned)hs - 1u;
    w.ovf = nullptr; w.ovf_cap = 0;
    return w;
}
__device__ __forceinline__ unsigned long long touched_key(int dest, unsigned chunk) { return (((unsigned long long)(unsigned)dest << 24) | chunk) + 1ull; }
__device__ __forceinline__ unsigned touched_hash(unsigned long long key) { return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40); }

template <int NV>
struct Row4 {
    float4 v[NV];
};

template <int NV>
__device__ __forceinline__ Row4<NV> ld_row4(const float* __restrict__ base, int64_t row, int64_t ld, int dim, int sub) {
    Row4<NV> r;
    const float4* p = reinterpret_cast<const float4*>(base + row * ld);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        r.v[j] = (c * 4 < dim) ? p[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return r;
}

// bf16 table (raw bit patterns): the same column mapping, 8 bytes (4 values) per lane and j -- a 16-lane group covers 128 contiguous bytes
template <int NV>
__device__ __forceinline__ Row4<NV> ld_row4(const uint16_t* __restrict__ base, int64_t row, int64_t ld, int dim, int sub) {
    Row4<NV> r;
    const uint2* p = reinterpret_cast<const uint2*>(base + row * ld);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        if (c * 4 < dim) {
            const uint2 u = p[c];
            r.v[j] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        } else {
            r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    return r;
}

template <int NV>
__device__ __forceinline__ void st_row4(float* __restrict__ base, int64_t row, int64_t ld, int dim, int sub, const Row4<NV>& r) {
    float4* p = reinterpret_cast<float4*>(base + row * ld);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = sub + 16 * j;
        if (c * 4 < dim) p[c] = r.v[j];
    }
}

template <int NV>
__device__ __forceinline__ float sumsq4(const Row4<NV>& r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        s = fmaf(r.v[j].x, r.v[j].x, s); s = fmaf(r.v[j].y, r.v[j].y, s);
        s = fmaf(r.v[j].z, r.v[j].z, s); s = fmaf(r.v[j].w, r.v[j].w, s);
    }
    return s;
}

// A slot in dest's bucket is RESERVED early (returning integer atomic, issued before the row gathers so that its round trip hides
// behind them) and FILLED once the weight is known; a reservation past the bucket's capacity goes to the overflow list.
__device__ __forceinline__ unsigned seg_reserve(const SegWork& w, int dest) { return atomicAdd(w.cnt + dest, 1u); }
template <bool TOUCHED>
__device__ __forceinline__ void seg_fill(const SegWork& w, unsigned which, int dest, unsigned pos, int src, float weight) {
    if (pos < (unsigned)w.cap) {
        w.list[(int64_t)dest * w.cap + pos] = SegEntry{src, weight};
    } else if constexpr (TOUCHED) {
        const unsigned over = pos - (unsigned)w.cap;
        const unsigned long long key = touched_key(dest, over / kHashChunk);
        unsigned long long* keys = w.hkeys[which];
        unsigned hslot = touched_hash(key) & w.hmask;
        for (unsigned probe = 0; probe <= w.hmask; ++probe) {            // claim the chunk's slot or find it claimed (no waiting: CAS per probe)
            const unsigned long long old = atomicCAS(keys + hslot, 0ull, key);
            if (old == 0ull || old == key) break;
            hslot = (hslot + 1u) & w.hmask;
        }
        w.hbuckets[(int64_t)hslot * kHashChunk + over % kHashChunk] = SegEntry{src, weight};
    } else {
        const unsigned o = atomicAdd(&w.hdr->ovf_cnt[which], 1u);
        if (o < w.ovf_cap) w.ovf[o] = SegOverflow{dest, src, weight, 0};
    }
    if constexpr (TOUCHED) {
        if (pos == 0u) w.touched[atomicAdd(&w.hdr->n_touched[which], 1u)] = (unsigned)dest;      // the destination's first entry announces it
    }
}

// ---- pass A --------------------------------------------------------------------------------------------------------------------
template <typename IdxT, int NV, typename TabT = float, bool TOUCHED = false>
__global__ __launch_bounds__(256) void score_seg_a_kernel(
    const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim, int n_ent,
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
        Row4<NV> q = ld_row4<NV>(ent, hi, eld, dim, sub);
        const Row4<NV> rr = ld_row4<NV>(rel, ri, dim, dim, sub);
        Row4<NV> tt = ld_row4<NV>(ent, ti, eld, dim, sub);
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
            const int my = (lane < cnt) ? (int)nrow[k0 + lane] : 0;
            unsigned my_pos = 0u;
            if (lane < cnt) my_pos = seg_reserve(w, my);    // in flight while the rows are gathered
            float my_s = 0.f;                               // lane k: score of negative k0 + k
            for (int it0 = 0; it0 < cnt; it0 += 32) {       // 8 rounds x 4 rows gathered before the first is used
                Row4<NV> x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = it0 + 4 * u + grp;
                    const int idx = __shfl(my, k & 63);
                    x[u] = ld_row4<NV>(ent, (k < cnt) ? idx : 0, eld, dim, sub);
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
                seg_fill<TOUCHED>(w, which, my, my_pos, (int)b, (margin + s_pos - my_s > 0.f) ? 2.f * gscale : 0.f);
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
        if (lane < 3) seg_fill<TOUCHED>(w, which, dest3, pos3, lane == 0 ? (int)b : (int)(batch + b), lane == 0 ? -2.f * c_pos : 1.f);
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
template <int NV, typename TabT = float, bool TOUCHED = false>
__global__ __launch_bounds__(256) void score_seg_b_kernel(const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim,
                                                          int n_ent, int n_dest, float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel,
                                                          SegWork w, int write_untouched, const float* __restrict__ loss_part, int64_t batch,
                                                          float* __restrict__ loss, int group_blocks) {
    if ((int)blockIdx.x == group_blocks) {
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
    const unsigned which = w.hdr->cur & 1u;
    if constexpr (TOUCHED) {
        if ((int)blockIdx.x > group_blocks) {
            // the remaining workgroups wipe the keys of the PREVIOUS launch's overflow hash (nobody reads that table in this launch)
            unsigned long long* keys = w.hkeys[which ^ 1u];
            for (unsigned i = ((unsigned)blockIdx.x - group_blocks - 1u) * 256u + threadIdx.x; i <= w.hmask; i += (gridDim.x - group_blocks - 1u) * 256u) keys[i] = 0ull;
            return;
        }
    }
    const int lane = threadIdx.x & 63, sub = lane & 15, gbase = lane & 48;
    const int group = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4);
    if (blockIdx.x == 0 && threadIdx.x == 0) {               // the other overflow list belongs to the previous launch: clear it, hand it to the next
        w.hdr->ovf_cnt[which ^ 1u] = 0u;
        if constexpr (TOUCHED) w.hdr->n_touched[which ^ 1u] = 0u;
        w.hdr->phase = which ^ 1u;
    }
    int dest_raw = group;
    bool have = group < n_dest;
    if constexpr (TOUCHED) {
        // walk the touched list (at most batch * (n_neg + 3) destinations, however large the tables are)
        const unsigned nt = w.hdr->n_touched[which];
        if ((unsigned)blockIdx.x * 16u >= nt) return;           // (whole workgroup: no barrier follows)
        have = (unsigned)group < nt;
        dest_raw = have ? (int)w.touched[group] : 0;
    }
    const int dest = have ? dest_raw : 0;                      // idle groups shadow destination 0 without storing (shuffles need all lanes)
    const bool is_rel = dest >= n_ent;
    // round trip 1
    const unsigned n = have ? w.cnt[dest] : 0u;
    const SegEntry mine = w.list[(int64_t)dest * w.cap + sub];
    Row4<NV> x = is_rel ? ld_row4<NV>(rel, dest - n_ent, dim, dim, sub) : ld_row4<NV>(ent, dest, eld, dim, sub);
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
        // big destination: order-free 64-bit fixed-point sums over its whole bucket and its share of the overflow
        long long acc[4 * NV];
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) acc[i] = 0;
        auto add_row = [&](const Row4<NV>& y, float wt) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                acc[4 * j + 0] += __double2ll_rn((double)(wt * y.v[j].x) * kFix);
                acc[4 * j + 1] += __double2ll_rn((double)(wt * y.v[j].y) * kFix);
                acc[4 * j + 2] += __double2ll_rn((double)(wt * y.v[j].z) * kFix);
                acc[4 * j + 3] += __double2ll_rn((double)(wt * y.v[j].w) * kFix);
            }
        };
        auto add = [&](int src, float wt) { add_row(ld_row4<NV>(w.rows, src, dim, dim, sub), wt); };
        // a contiguous run of entries, 16 at a time: lane i of the group fetches entry i (one coalesced 128-byte read), then the rows
        // they name are gathered four at a time -- two dependent round trips per 16 entries instead of two per entry
        auto add_run = [&](const SegEntry* run, unsigned count) {
            for (unsigned i0 = 0; i0 < count; i0 += 16) {
                const SegEntry e = (i0 + sub < count) ? run[i0 + sub] : SegEntry{0, 0.f};
                const unsigned m = min(16u, count - i0);
                for (unsigned u0 = 0; u0 < m; u0 += 4) {
                    Row4<NV> y[4];
                    float wt[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = __shfl(e.src, gbase + (int)((u0 + u) & 15u));
                        wt[u] = (u0 + u < m) ? __shfl(e.w, gbase + (int)((u0 + u) & 15u)) : 0.f;
                        y[u] = ld_row4<NV>(w.rows, (u0 + u < m) ? src : 0, dim, dim, sub);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) add_row(y[u], wt[u]);
                }
            }
        };
        const unsigned in_list = min(n, (unsigned)w.cap);
        add_run(w.list + (int64_t)dest * w.cap, in_list);
        if (n > (unsigned)w.cap) {
            if constexpr (TOUCHED) {
                // hot row: its surplus sits in 64-entry chunks of the overflow hash, chunk j under the key (dest, j)
                const unsigned over = n - (unsigned)w.cap;
                const unsigned long long* keys = w.hkeys[which];
                for (unsigned j = 0; j * kHashChunk < over; ++j) {
                    const unsigned long long key = touched_key(dest, j);
                    unsigned hslot = touched_hash(key) & w.hmask;
                    for (unsigned probe = 0; probe <= w.hmask && keys[hslot] != key; ++probe) hslot = (hslot + 1u) & w.hmask;
                    add_run(w.hbuckets + (int64_t)hslot * kHashChunk, min((unsigned)kHashChunk, over - j * kHashChunk));
                }
            } else {                                          // hub: its surplus sits somewhere in the overflow list
                const unsigned n_ovf = min(w.hdr->ovf_cnt[which], w.ovf_cap);
                for (unsigned i = 0; i < n_ovf; ++i) {
                    const SegOverflow e = w.ovf[i];
                    if (e.dest == dest) add(e.src, e.w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            g.v[j].x = (float)((double)acc[4 * j + 0] * (1.0 / kFix)); g.v[j].y = (float)((double)acc[4 * j + 1] * (1.0 / kFix));
            g.v[j].z = (float)((double)acc[4 * j + 2] * (1.0 / kFix)); g.v[j].w = (float)((double)acc[4 * j + 3] * (1.0 / kFix));
        }
    }
    // the backward of F.normalize on this row:  (G - x^ (x^ . G)) / |x|      (clamped norm: G / eps)
    const float den = l2_denominator(sum16(sumsq4<NV>(x)));
    const float inv = 1.0f / den;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        x.v[j].x *= inv; x.v[j].y *= inv; x.v[j].z *= inv; x.v[j].w *= inv;
        dot = fmaf(x.v[j].x, g.v[j].x, dot); dot = fmaf(x.v[j].y, g.v[j].y, dot);
        dot = fmaf(x.v[j].z, g.v[j].z, dot); dot = fmaf(x.v[j].w, g.v[j].w, dot);
    }
    dot = sum16(dot);
    if (!(den > 1e-12f)) dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g.v[j].x = (g.v[j].x - x.v[j].x * dot) * inv; g.v[j].y = (g.v[j].y - x.v[j].y * dot) * inv;
        g.v[j].z = (g.v[j].z - x.v[j].z * dot) * inv; g.v[j].w = (g.v[j].w - x.v[j].w * dot) * inv;
    }
    if (have && (n > 0u || write_untouched)) st_row4<NV>(out, 0, 0, dim, sub, g);      // n == 0: g is exactly zero
}

}  // namespace snag

using namespace snag;

namespace snag {
// both forms: argument checks, pass A, pass B
template <typename TabT, bool TOUCHED>
static int launch_seg(const char* who, const TabT* ent, int64_t ent_ld, int64_t n_ent, const TabT* rel, int64_t n_rel, int dim, const void* h, const void* r,
                      const void* t, const void* neg_t, int idx_bytes, int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                      float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld, float* d_rel, int write_untouched, void* work, int64_t work_bytes,
                      hipStream_t st) {
    SNAG_REQUIRE(ent && rel && d_ent && d_rel && work && (batch == 0 || loss_part), SNAG_ERR_ARG, "%s: null table / gradient / workspace / loss_part", who);
    SNAG_REQUIRE(n_ent > 0 && n_rel > 0 && dim > 0 && batch >= 0 && n_neg >= 0, SNAG_ERR_ARG, "%s: bad sizes", who);
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "%s: idx_bytes=%d (want 4 or 8)", who, idx_bytes);
    SNAG_REQUIRE(dim <= 512 && dim % 4 == 0 && ent_ld % 4 == 0 && d_ent_ld % 4 == 0 && ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_UNSUPPORTED,
                 "%s: dim=%d (<= 512) and the row strides must be multiples of 4 (16-byte fp32 / 8-byte bf16 rows)", who, dim);
    SNAG_REQUIRE(n_ent + n_rel < (1LL << 27), SNAG_ERR_UNSUPPORTED, "%s: n_ent + n_rel >= 2^27", who);
    const size_t need = TOUCHED ? touched_work_bytes(n_ent, n_rel, batch, n_neg, dim) : seg_work_bytes(n_ent, n_rel, batch, n_neg, dim);
    SNAG_REQUIRE(work_bytes >= (int64_t)need, SNAG_ERR_ARG, "%s: workspace holds %ld bytes, need %ld", who, (long)work_bytes, (long)need);
    constexpr uintptr_t tab_align = sizeof(TabT) == 4 ? 15 : 7;
    SNAG_REQUIRE(((uintptr_t)work & 255) == 0 && ((uintptr_t)ent & tab_align) == 0 && ((uintptr_t)rel & tab_align) == 0 && ((uintptr_t)d_ent & 15) == 0 &&
                 ((uintptr_t)d_rel & 15) == 0, SNAG_ERR_UNSUPPORTED, "%s: tables and gradients must be 16-byte aligned (bf16 tables: 8), the workspace 256-byte aligned", who);
    SNAG_REQUIRE(batch == 0 || (h && r && t && pos && (n_neg == 0 || (neg && neg_t))), SNAG_ERR_ARG, "%s: null index or score pointer", who);
    const SegWork w = TOUCHED ? touched_carve(work, n_ent, n_rel, batch, n_neg, dim) : seg_carve(work, n_ent, n_rel, batch, n_neg, dim);
    const int nv = (dim + 63) / 64;
    const int n_dest = (int)(n_ent + n_rel);
    if (batch > 0) {
        const dim3 grid((unsigned)ceil_div(batch, 4)), block(256);
#define SNAG_SEG_A(IDX, NV)                                                                                                         \
        hipLaunchKernelGGL((score_seg_a_kernel<IDX, NV, TabT, TOUCHED>), grid, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, (const IDX*)h, (const IDX*)r, \
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
        int rc = check_launch(who);
        if (rc) return rc;
    }
    // pass B: one 16-lane group per destination (touched form: per POSSIBLY touched destination -- the list cannot be longer than the
    // batch's entries), + 1 workgroup that sums the per-triple losses (an empty batch: loss = 0), + (touched form) 8 that wipe the
    // previous launch's hash keys
    const int64_t groups = TOUCHED ? std::min<int64_t>(n_dest, std::max<int64_t>(1, batch * (n_neg + 3))) : (int64_t)n_dest;
    const int group_blocks = ceil_div(groups * 16, 256);
    const dim3 gridb((unsigned)group_blocks + 1u + (TOUCHED ? 8u : 0u)), block(256);
#define SNAG_SEG_B(NV) hipLaunchKernelGGL((score_seg_b_kernel<NV, TabT, TOUCHED>), gridb, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, n_dest, d_ent, d_ent_ld, d_rel, w, \
                                          write_untouched, (const float*)loss_part, batch, loss, group_blocks)
    switch (nv) {
        case 1: SNAG_SEG_B(1); break;
        case 2: SNAG_SEG_B(2); break;
        case 3: SNAG_SEG_B(3); break;
        case 4: SNAG_SEG_B(4); break;
        case 5: case 6: SNAG_SEG_B(6); break;
        default: SNAG_SEG_B(8); break;
    }
#undef SNAG_SEG_B
    return check_launch(who);
}
}  // namespace snag

extern "C" SNAG_API int64_t snag_score_seg_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    if (n_ent < 1 || n_rel < 1 || batch < 0 || n_neg < 0 || dim < 1) return 0;
    return (int64_t)seg_work_bytes(n_ent, n_rel, batch, n_neg, dim);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_seg(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                      const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                      int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                      float* loss_part, float* loss, uint32_t* ticket, float* d_ent, int64_t d_ent_ld,
                                                      float* d_rel, int write_untouched, void* work, int64_t work_bytes, void* stream) {
    (void)ticket;          // (the loss is summed by pass B since round 3: no last-block election, the ticket words are not touched)
    return launch_seg<float, false>("snag_score_margin_fwd_bwd_seg", ent, ent_ld, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg, margin, grad_scale,
                                    pos, neg, loss_part, loss, d_ent, d_ent_ld, d_rel, write_untouched, work, work_bytes, (hipStream_t)stream);
}

extern "C" SNAG_API int64_t snag_score_touched_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    if (n_ent < 1 || n_rel < 1 || batch < 0 || n_neg < 0 || dim < 1) return 0;
    return (int64_t)touched_work_bytes(n_ent, n_rel, batch, n_neg, dim);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_touched(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                          int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                          float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld, float* d_rel,
                                                          void* work, int64_t work_bytes, void* stream) {
    return launch_seg<float, true>("snag_score_margin_fwd_bwd_touched", ent, ent_ld, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg, margin,
                                   grad_scale, pos, neg, loss_part, loss, d_ent, d_ent_ld, d_rel, 0, work, work_bytes, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_touched_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                                                               const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                               int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                               float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld, float* d_rel,
                                                               void* work, int64_t work_bytes, void* stream) {
    return launch_seg<uint16_t, true>("snag_score_margin_fwd_bwd_touched_bf16", ent, ent_ld, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg,
                                      margin, grad_scale, pos, neg, loss_part, loss, d_ent, d_ent_ld, d_rel, 0, work, work_bytes, (hipStream_t)stream);
}
