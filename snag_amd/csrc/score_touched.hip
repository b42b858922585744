// a7, training form for tables FAR LARGER than a batch's footprint (BASELINE configs[4]: |E| = 1M, dim 256, 128 negative tails, bf16
// tables): the segmented scorer of score_seg.hip (no float atomics, bit-reproducible; reference model/MSNEA.py:110-127 + its autograd)
// with a second pass that walks the batch's ENTRIES -- at most batch * (n_neg + 3) -- instead of every destination's counter.
//
//   pass A   one wave per (triple, part): a triple's negatives are dealt to `parts` waves (4 at 128 negatives: at one wave per triple a
//            1,024-triple batch is 4 waves per CU, and the gathers run at the latency of a single wave's chain).  Each part scores its
//            negatives against q_b = h^ + r^ and leaves its share dq_{b,p} of the query gradient as a row of its own -- the gradient is
//            linear in these rows, so head and relation simply receive one entry per part.  Every entry (destination, row, weight)
//            reserves a slot in its destination with a returning integer atomic and is written, IN TRIPLE ORDER (coalesced), as a
//            16-byte record {dest, row, weight, slot}; slots 1..15 also go to the destination's inline bucket, slots >= 16 (hot rows
//            of skewed index distributions) to an open-addressed hash of 64-entry chunks keyed (dest, chunk): wait-free insertion (one
//            compare-and-swap per probe), every entry has exactly one home.
//   pass B'  one 16-lane group per hash slot: the chunk's <= 64 rows summed in 64-bit fixed point into the slot's partial row.
//   pass B   one 16-lane group per RECORD.  The record whose slot is 0 owns its destination: with one entry (nearly every destination of
//            a uniformly drawn batch) it has all it needs in hand -- two dependent round trips: the record, then {counter, own table
//            row, the entry's row} --, with up to 16 it adds the bucket in a canonical order (sorted fp32); hot rows are handed to
//            pass C.  Records with other slots leave at once.  The per-part losses are summed here too (fixed order).
//   pass C   one 16-lane group per hot row: bucket + chunk partials in fixed point (order-free), normalisation backward, store.
// Only touched rows of d_ent / d_rel are written; the caller keeps the rest zero.
#include <algorithm>

#include "seg_rows.hip.h"

namespace snag {

#ifndef SNAG_TA_WAVES
#define SNAG_TA_WAVES 2                  // (waves per SIMD pass A is compiled for: 2 = up to 256 registers, no spills: 46 -> 38 us against 3)
#endif
#ifndef SNAG_TB_WAVES
#define SNAG_TB_WAVES 4
#endif
#ifndef SNAG_HUB_GRID
#define SNAG_HUB_GRID 512
#endif
constexpr int kTCap = 16;                   // inline slots per destination (slot 0 is the owner's own record)
constexpr int kTChunk = 64;                 // entries per hash chunk

struct TouchedHeader {                      // first 64 bytes of the workspace (zeroed once by the caller)
    unsigned phase;                         // which hash-key table / hub counter the NEXT pass A uses (flipped by pass B)
    unsigned cur;                           // which one the CURRENT launch uses (written by pass A, read by the later passes)
    unsigned n_hubs[2];
    unsigned n_chunks[2];                   // claimed hash chunks of the even / odd launch (their slots: TouchedWork::chunks)
    unsigned pad[10];
};
struct TouchedRec {                         // one entry in triple order: weight * rows[src] goes to destination dest; slot = arrival number there
    int dest, src;
    float w;
    unsigned slot;
};
struct TouchedHub {
    int dest;
    unsigned rec;                           // index of the owner's record (the entry in slot 0)
};

struct TouchedWork {
    TouchedHeader* hdr;
    unsigned* cnt;                          // [n_dest] entries received (reset by whoever finishes the destination)
    SegEntry* list;                         // [n_dest, kTCap]: slots 1..15
    TouchedRec* rec;                        // [n_rec]
    unsigned long long* hkeys[2];           // [hmask + 1] each: (dest, chunk) + 1, 0 = free; even / odd launch
    SegEntry* hbuckets;                     // [hmask + 1, kTChunk]
    long long* hpart;                       // [hmask + 1, dim]: fixed-point sum of a chunk
    unsigned* chunks;                       // [hmask + 1]: hash slots of the claimed chunks, in claiming order (pass B' walks this list, not the table)
    TouchedHub* hubs;                       // [max_hubs]
    float* rows;                            // [(1 + parts) * batch, dim]: q_b, then dq_{b,p} at batch + b * parts + p
    float* lpart;                           // [batch * parts] hinge sums per (triple, part)
    unsigned hmask, max_hubs;
    int parts, per;                         // waves per triple, negatives per part
    int64_t n_rec;                          // batch * parts * (per + 3)
};

static size_t t_align(size_t x) { return (x + 255) / 256 * 256; }

// waves per triple: enough of them to hide the gathers' latency (pass A is 4 waves per CU at batch 1,024 otherwise)
static int touched_parts(int64_t batch, int n_neg) {
    int p = n_neg >= 96 ? 4 : (n_neg >= 48 ? 2 : 1);
    while (p > 1 && batch * p > 16384) p /= 2;
    while ((n_neg + p - 1) / p > 250) p *= 2;           // pass A keeps a part's negatives in four 64-lane id blocks and a 1,024-slot LDS table per workgroup
    return p;
}
static unsigned touched_hash_slots(int64_t n_rec) {
    const int64_t need = 2 * (n_rec / kTChunk + n_rec / kTCap + 1);
    unsigned h = 1024;
    while ((int64_t)h < need) h *= 2;
    return h;
}
struct TouchedSizes {
    size_t n_dest, n_rec, hs, max_hubs;
    int parts, per;
};
static TouchedSizes touched_sizes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg) {
    TouchedSizes z;
    z.parts = touched_parts(batch, n_neg);
    z.per = (n_neg + z.parts - 1) / z.parts;
    z.n_dest = (size_t)(n_ent + n_rel);
    z.n_rec = (size_t)batch * z.parts * (z.per + 3);
    z.hs = touched_hash_slots((int64_t)z.n_rec);
    z.max_hubs = z.n_rec / (kTCap + 1) + 1;
    return z;
}
static size_t touched_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    const TouchedSizes z = touched_sizes(n_ent, n_rel, batch, n_neg);
    return t_align(sizeof(TouchedHeader)) + t_align(z.n_dest * 4) + t_align(z.n_dest * kTCap * sizeof(SegEntry)) + t_align(z.n_rec * sizeof(TouchedRec)) +
           2 * t_align(z.hs * 8) + t_align(z.hs * kTChunk * sizeof(SegEntry)) + t_align(z.hs * (size_t)dim * 8) + t_align(z.hs * 4) + t_align(z.max_hubs * sizeof(TouchedHub)) +
           t_align((size_t)(1 + z.parts) * batch * dim * 4) + t_align((size_t)batch * z.parts * 4);
}
static TouchedWork touched_carve(void* base, int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    const TouchedSizes z = touched_sizes(n_ent, n_rel, batch, n_neg);
    char* p = (char*)base;
    TouchedWork w{};
    w.hdr = (TouchedHeader*)p; p += t_align(sizeof(TouchedHeader));
    w.cnt = (unsigned*)p; p += t_align(z.n_dest * 4);
    w.list = (SegEntry*)p; p += t_align(z.n_dest * kTCap * sizeof(SegEntry));
    w.rec = (TouchedRec*)p; p += t_align(z.n_rec * sizeof(TouchedRec));
    w.hkeys[0] = (unsigned long long*)p; p += t_align(z.hs * 8);
    w.hkeys[1] = (unsigned long long*)p; p += t_align(z.hs * 8);
    w.hbuckets = (SegEntry*)p; p += t_align(z.hs * kTChunk * sizeof(SegEntry));
    w.hpart = (long long*)p; p += t_align(z.hs * (size_t)dim * 8);
    w.chunks = (unsigned*)p; p += t_align(z.hs * 4);
    w.hubs = (TouchedHub*)p; p += t_align(z.max_hubs * sizeof(TouchedHub));
    w.rows = (float*)p; p += t_align((size_t)(1 + z.parts) * batch * dim * 4);
    w.lpart = (float*)p;
    w.hmask = (unsigned)z.hs - 1u;
    w.max_hubs = (unsigned)z.max_hubs;
    w.parts = z.parts;
    w.per = z.per;
    w.n_rec = (int64_t)z.n_rec;
    return w;
}

__device__ __forceinline__ unsigned long long touched_key(int dest, unsigned chunk) { return (((unsigned long long)(unsigned)dest << 24) | chunk) + 1ull; }
// (the top 32 bits of the product: masked down to the table size by the callers -- tables above 2^24 slots keep spreading)
__device__ __forceinline__ unsigned touched_hash(unsigned long long key) { return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 32); }
// slot of the chunk (dest, j) in the table `keys` (present: pass A inserted it)
__device__ __forceinline__ unsigned touched_find(const unsigned long long* __restrict__ keys, unsigned hmask, int dest, unsigned j) {
    const unsigned long long key = touched_key(dest, j);
    unsigned hslot = touched_hash(key) & hmask;
    for (unsigned probe = 0; probe <= hmask && keys[hslot] != key; ++probe) hslot = (hslot + 1u) & hmask;
    return hslot;
}

// the entry's record, and its home: bucket slot 1..15 or a hash chunk (slot 0 lives in the record alone)
__device__ __forceinline__ void touched_emit(const TouchedWork& w, unsigned which, int64_t rec_idx, int dest, unsigned slot, int src, float weight) {
    w.rec[rec_idx] = TouchedRec{dest, src, weight, slot};
    if (slot == 0u) return;
    if (slot < (unsigned)kTCap) {
        w.list[(int64_t)dest * kTCap + slot] = SegEntry{src, weight};
        return;
    }
    const unsigned over = slot - (unsigned)kTCap;
    const unsigned long long key = touched_key(dest, over / kTChunk);
    unsigned long long* keys = w.hkeys[which];
    unsigned hslot = touched_hash(key) & w.hmask;
    for (unsigned probe = 0; probe <= w.hmask; ++probe) {            // claim the chunk's slot or find it claimed (no waiting: one CAS per probe)
        const unsigned long long old = atomicCAS(keys + hslot, 0ull, key);
        if (old == 0ull) w.chunks[atomicAdd(&w.hdr->n_chunks[which], 1u)] = hslot;      // the claimer lists the chunk for pass B'
        if (old == 0ull || old == key) break;
        hslot = (hslot + 1u) & w.hmask;
    }
    w.hbuckets[(int64_t)hslot * kTChunk + over % kTChunk] = SegEntry{src, weight};
}

// ---- pass A --------------------------------------------------------------------------------------------------------------------
template <typename IdxT, int NV, typename TabT, int kAgg>
__global__ __launch_bounds__(256, SNAG_TA_WAVES) void touched_a_kernel(
    const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim, int n_ent,
    const IdxT* __restrict__ h, const IdxT* __restrict__ r, const IdxT* __restrict__ t, const IdxT* __restrict__ neg_t,
    int64_t batch, int n_neg_all, float margin, float gscale, float* __restrict__ pos, float* __restrict__ neg, TouchedWork w) {
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const int64_t vb_raw = (int64_t)blockIdx.x * 4 + wave;         // (triple, part)
    const unsigned which = w.hdr->phase & 1u;                      // stable during pass A (only pass B flips it)
    if (blockIdx.x == 0 && threadIdx.x == 0) w.hdr->cur = which;
    // SLOT RESERVATION, one returning atomic per workgroup and distinct destination.  Every entry needs its arrival number at its
    // destination; taken one by one (a returning atomicAdd per entry) the ~9,000 entries a Zipf(1.0) batch sends to its hottest row queue
    // up on one address at ~12 ns each -- pass A took 127 us against 42 on uniform indices.  The workgroup's entries (four (triple, part)
    // waves: <= 4 (per + 3)) first meet in an LDS hash keyed by destination (local arrival number = an LDS atomic), then ONE lane per
    // distinct destination reserves the whole run at the global counter; an entry's slot is the run's base + its local number.  The global
    // round trip is in flight while the rows are gathered and scored; it is waited for just before the records are written.
    // kAgg LDS hash slots: 256 when the workgroup's <= 4 (per + 3) entries fill at most 5/8 of them, 1,024 otherwise (per <= 250: touched_parts)
    __shared__ unsigned agg_key[kAgg], agg_cnt[kAgg], agg_base[kAgg];
    for (int i = threadIdx.x; i < kAgg; i += 256) { agg_key[i] = 0u; agg_cnt[i] = 0u; }
    __syncthreads();
    const bool live = vb_raw < batch * w.parts;
    const int64_t vb = live ? vb_raw : 0;
    const int64_t b = vb / w.parts;
    const int part = (int)(vb % w.parts);
    const int k_lo = min(n_neg_all, part * w.per);
    const int n_neg = live ? min(n_neg_all - k_lo, w.per) : 0;
    const int64_t rec0 = vb * (w.per + 3);                          // this part's records: t, h, r, then its negatives (slots past n_neg: void)
    float hinge = 0.f;
    const int64_t hi = (int64_t)h[b], ri = (int64_t)r[b], ti = (int64_t)t[b];
    const IdxT* nrow = neg_t + b * (int64_t)n_neg_all + k_lo;
    // local arrival number of an entry at `dest` inside this workgroup, and the LDS slot that stands for the destination
    auto agg_insert = [&](int dest, unsigned& local, int& aslot) {
        unsigned a = (((unsigned)dest * 2654435761u) >> 22) & (kAgg - 1);
        for (;;) {
            const unsigned old = atomicCAS(&agg_key[a], 0u, (unsigned)dest + 1u);
            if (old == 0u || old == (unsigned)dest + 1u) break;
            a = (a + 1u) & (kAgg - 1);
        }
        local = atomicAdd(&agg_cnt[a], 1u);
        aslot = (int)a;
    };
    // the positive tail, the head and the relation (lanes 0, 1, 2); the negatives' destinations follow
    const int dest3 = lane == 0 ? (int)ti : (lane == 1 ? (int)hi : n_ent + (int)ri);
    unsigned local3 = 0u;
    int aslot3 = 0;
    if (live && lane < 3) agg_insert(dest3, local3, aslot3);
    constexpr int kMaxBlocks = 4;                                   // id blocks of 64 negatives per part held in registers (per <= 256)
    unsigned nlocal[kMaxBlocks];
    int naslot[kMaxBlocks];
#pragma unroll
    for (int q = 0; q < kMaxBlocks; ++q) {
        nlocal[q] = 0u; naslot[q] = 0;
        const int k0 = 64 * q;
        if (k0 < w.per && lane < max(0, min(64, n_neg - k0))) agg_insert((int)nrow[k0 + lane], nlocal[q], naslot[q]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kAgg; i += 256)
        if (agg_cnt[i] != 0u) agg_base[i] = atomicAdd(w.cnt + (agg_key[i] - 1u), agg_cnt[i]);       // in flight while the rows are gathered
    // RB rounds x 4 rows gathered before the first is used (wide rows: 4 rounds -- 64 fewer registers, three waves per SIMD instead of two;
    // a group meets its negatives in the same order either way, so the sums are the same bits).  (Issuing the first batch next to the
    // head / relation / tail rows, before any arithmetic, was tried: 241 registers, or 168 with spills -- 78 us instead of 43.)
    constexpr int RB = NV >= 3 ? 4 : 8;
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
    if (live && lane == 0 && part == 0) pos[b] = s_pos;
    float c_pos = 0.f;
    for (int k0 = 0; k0 < w.per; k0 += 64) {
        const int cnt = max(0, min(64, n_neg - k0));
        const int my = (lane < cnt) ? (int)nrow[k0 + lane] : 0;
        float my_s = 0.f;                               // lane k: score of negative k0 + k
        for (int it0 = 0; it0 < cnt; it0 += 4 * RB) {
            Row4<NV> x[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int k = it0 + 4 * u + grp;
                const int idx = __shfl(my, k & 63);
                x[u] = ld_row4<NV>(ent, (k < cnt) ? idx : 0, eld, dim, sub);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
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
                const float sk = __shfl(s, (lane & 3) * 16);
                if (((lane - it0) >> 2) == u && lane >= it0 && lane < it0 + 4 * RB) my_s = sk;
            }
        }
        // the reserved runs have arrived by now (first id block: the workgroup meets here; same trip count in every wave)
        if (k0 == 0) __syncthreads();
        // lane k: score out, its negative's record (slot = its destination's base + its local number; weight 2 a_k, zero for an inactive
        // hinge); slots of this id block past the part's negatives are void records
        const int q = k0 >> 6;
        const unsigned nl = q == 0 ? nlocal[0] : (q == 1 ? nlocal[1] : (q == 2 ? nlocal[2] : nlocal[3]));
        const int na = q == 0 ? naslot[0] : (q == 1 ? naslot[1] : (q == 2 ? naslot[2] : naslot[3]));
        if (lane < cnt) {
            neg[b * (int64_t)n_neg_all + k_lo + k0 + lane] = my_s;
            touched_emit(w, which, rec0 + 3 + k0 + lane, my, agg_base[na] + nl, (int)b, (margin + s_pos - my_s > 0.f) ? 2.f * gscale : 0.f);
        } else if (live && k0 + lane < w.per) {
            w.rec[rec0 + 3 + k0 + lane] = TouchedRec{-1, 0, 0.f, 1u};
        }
    }
    if (w.per <= 0) __syncthreads();
    // the four groups' partial sums, combined in fixed order: every group then holds the whole part's values
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
    // rows[b] = q_b (part 0) ; rows[batch + vb] = dq_{b,p} = 2 c_p (q - t^) - 2 sum_{k in p} a_k (q - x^_k)
    if (live && grp == 0 && part == 0) st_row4<NV>(w.rows, b, dim, dim, sub, q);
    if (live && grp == 1) {
        Row4<NV> dq;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            dq.v[j].x = 2.f * (c_pos * (q.v[j].x - tt.v[j].x) - acc.v[j].x); dq.v[j].y = 2.f * (c_pos * (q.v[j].y - tt.v[j].y) - acc.v[j].y);
            dq.v[j].z = 2.f * (c_pos * (q.v[j].z - tt.v[j].z) - acc.v[j].z); dq.v[j].w = 2.f * (c_pos * (q.v[j].w - tt.v[j].w) - acc.v[j].w);
        }
        st_row4<NV>(w.rows, batch + vb, dim, dim, sub, dq);
    }
    if (!live) return;
    if (lane < 3) touched_emit(w, which, rec0 + lane, dest3, agg_base[aslot3] + local3, lane == 0 ? (int)b : (int)(batch + vb), lane == 0 ? -2.f * c_pos : 1.f);
    if (lane == 0) w.lpart[vb] = hinge;
}

// ---- pass B': chunk sums (a role of pass B's launch: its first `chunk_blocks` workgroups; only pass C reads the result) ------------------
template <int NV>
__device__ __forceinline__ void touched_chunk_body(const TouchedWork& w, int dim, int vblock, int nblocks) {
    // ONE WAVE per listed chunk (a fixed grid walks the list): its four 16-lane groups take 16 entries each -- four dependent round trips
    // instead of sixteen for a full chunk -- and meet in registers; integer sums, so the result does not depend on who added what
    const int lane = threadIdx.x & 63, sub = lane & 15, gbase = lane & 48, grp = lane >> 4;
    const unsigned which = w.hdr->cur & 1u;
    const unsigned n_list = min(w.hdr->n_chunks[which], w.hmask + 1u);
    for (unsigned ci = (unsigned)vblock * 4u + (threadIdx.x >> 6); ci < n_list; ci += (unsigned)nblocks * 4u) {
        const unsigned hslot = w.chunks[ci];
        const unsigned long long key = w.hkeys[which][hslot];
        const int dest = (int)((key - 1ull) >> 24);
        const unsigned j = (unsigned)((key - 1ull) & 0xffffffull);
        const unsigned n = w.cnt[dest];                         // (hot rows' counters are reset by pass C, not by pass B)
        const unsigned over = n > (unsigned)kTCap ? n - (unsigned)kTCap : 0u;
        const unsigned count = over > j * kTChunk ? min((unsigned)kTChunk, over - j * kTChunk) : 0u;
        const unsigned mine = count > 16u * grp ? min(16u, count - 16u * grp) : 0u;
        long long acc[4 * NV];
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) acc[i] = 0;
        fix_add_run<NV>(acc, w.hbuckets + (int64_t)hslot * kTChunk + 16 * grp, mine, w.rows, dim, sub, gbase);
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) {
            acc[i] += __shfl_xor(acc[i], 16);
            acc[i] += __shfl_xor(acc[i], 32);
        }
        if (grp == 0) {
            long long* out = w.hpart + (int64_t)hslot * dim;
#pragma unroll
            for (int jv = 0; jv < NV; ++jv) {
                const int c = (sub + 16 * jv) * 4;
                if (c < dim) {
                    reinterpret_cast<longlong2*>(out + c)[0] = make_longlong2(acc[4 * jv + 0], acc[4 * jv + 1]);
                    reinterpret_cast<longlong2*>(out + c)[1] = make_longlong2(acc[4 * jv + 2], acc[4 * jv + 3]);
                }
            }
        }
    }
}

// ---- pass B: one group per record -----------------------------------------------------------------------------------------------------
template <int NV, typename TabT>
__global__ __launch_bounds__(256, SNAG_TB_WAVES) void touched_b_kernel(const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim, int n_ent,
                                                        float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel, TouchedWork w,
                                                        float* __restrict__ loss_part, int64_t batch, float* __restrict__ loss, int group_blocks,
                                                        int chunk_blocks) {
    // workgroup roles: [0, chunk_blocks) chunk sums of the overflow hash (pass B'), then one group per record, then the loss sum, then
    // the wipe of the previous launch's hash keys
    if ((int)blockIdx.x < chunk_blocks) {
        touched_chunk_body<NV>(w, dim, (int)blockIdx.x, chunk_blocks);
        return;
    }
    const int bid = (int)blockIdx.x - chunk_blocks;
    if (bid == group_blocks) {
        // per-triple losses = the parts' sums in part order; the total in one fixed order (thread-strided, wave reductions, wave order)
        __shared__ float part[4];
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < batch; i += 256) {
            float l = 0.f;
            for (int p = 0; p < w.parts; ++p) l += w.lpart[i * w.parts + p];
            loss_part[i] = l;
            s += l;
        }
        s = sum64(s);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0 && loss != nullptr) loss[0] = ((part[0] + part[1]) + part[2]) + part[3];
        return;
    }
    const unsigned which = w.hdr->cur & 1u;
    if (bid > group_blocks) {
        // the remaining workgroups wipe the keys of the PREVIOUS launch's overflow hash (nobody reads that table in this launch)
        unsigned long long* keys = w.hkeys[which ^ 1u];
        const unsigned nb = gridDim.x - chunk_blocks - group_blocks - 1u;
        for (unsigned i = ((unsigned)bid - group_blocks - 1u) * 256u + threadIdx.x; i <= w.hmask; i += nb * 256u) keys[i] = 0ull;
        return;
    }
    if (bid == 0 && threadIdx.x == 0) {               // hand the other table / hub counter to the next launch
        w.hdr->n_hubs[which ^ 1u] = 0u;
        w.hdr->n_chunks[which ^ 1u] = 0u;
        w.hdr->phase = which ^ 1u;
    }
    const int lane = threadIdx.x & 63, sub = lane & 15, gbase = lane & 48;
    const int64_t g = ((int64_t)bid * 256 + threadIdx.x) >> 4;
    // round trip 1: the record (the 16 lanes of a group read the same 16 bytes)
    const TouchedRec rc = g < w.n_rec ? w.rec[g] : TouchedRec{-1, 0, 0.f, 1u};
    const bool owner = rc.dest >= 0 && rc.slot == 0u;
    if (__ballot(owner) == 0ull) return;
    const int dest = owner ? rc.dest : 0;
    const bool is_rel = dest >= n_ent;
    // round trip 2: how many entries arrived, the destination's own row, the owner entry's row, the rest of the bucket
    const unsigned n = owner ? w.cnt[dest] : 0u;
    // (one load sequence through a selected pointer: as two branches the compiler waits for the row inside its branch, before the
    // other round-trip-2 loads are even issued)
    const TabT* xrow = is_rel ? rel + (int64_t)(dest - n_ent) * dim : ent + (int64_t)dest * eld;
    Row4<NV> x = ld_row4<NV>(xrow, 0, 0, dim, sub);
    const Row4<NV> y0 = ld_row4<NV>(w.rows, owner ? rc.src : 0, dim, dim, sub);
    const bool hub = n > (unsigned)kTCap;
    if (owner && !hub && sub == 0) w.cnt[dest] = 0u;          // ready for the next launch (hot rows: pass C)
    // hot rows go to pass C: one list slot per wave-instruction
    {
        const unsigned long long hm = __ballot(owner && hub && sub == 0);
        if (hm != 0ull) {
            const int leader = __builtin_ctzll(hm);
            unsigned base = 0u;
            if (lane == leader) base = atomicAdd(&w.hdr->n_hubs[which], (unsigned)__builtin_popcountll(hm));
            base = __shfl(base, leader);
            if (owner && hub && sub == 0) {
                const unsigned at = base + (unsigned)__builtin_popcountll(hm & ((1ull << lane) - 1ull));
                if (at < w.max_hubs) w.hubs[at] = TouchedHub{dest, (unsigned)g};
            }
        }
    }
    Row4<NV> gsum;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float w0 = (owner && n == 1u) ? rc.w : 0.f;
        gsum.v[j] = make_float4(w0 * y0.v[j].x, w0 * y0.v[j].y, w0 * y0.v[j].z, w0 * y0.v[j].w);
    }
    const bool multi = owner && !hub && n > 1u;
    if (__ballot(multi) != 0ull) {
        // (a third round trip, for the ~6 % of a uniform batch's destinations that received more than one entry: the bucket's slots 1..15)
        SegEntry mine = multi ? w.list[(int64_t)dest * kTCap + sub] : SegEntry{0, 0.f};
        if (sub == 0) mine = SegEntry{rc.src, rc.w};              // slot 0 is the record itself
        // 2..16 entries: canonical order by (row, weight bits, slot) key -- the slot order the atomics produced varies from run to run, the
        // sorted order does not --, fp32 sums in that order, four rows gathered per step
        const int nf = multi ? (int)n : 0;
        const long long key = ((long long)mine.src << 32) | (unsigned)__float_as_int(mine.w);
        int rank = 0;
#pragma unroll
        for (int j = 0; j < kTCap; ++j) {
            const long long kj = __shfl(key, gbase + j);
            rank += (j < nf && (kj < key || (kj == key && j < sub))) ? 1 : 0;
        }
        int wave_max = nf;
        wave_max = max(wave_max, __shfl_xor(wave_max, 16));
        wave_max = max(wave_max, __shfl_xor(wave_max, 32));
        for (int r0 = 0; r0 < wave_max; r0 += 2) {                 // (two rows per step: this path is rare, the kernel's register count is every group's)
            Row4<NV> y[2];
            float wt[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned long long m = __ballot(sub < nf && rank == r0 + u) >> gbase;
                const int src_lane = gbase + (int)__builtin_ctzll(m | 0x10000ull);
                const int src = __shfl(mine.src, src_lane);
                const bool on = r0 + u < nf;
                wt[u] = on ? __shfl(mine.w, src_lane) : 0.f;
                y[u] = ld_row4<NV>(w.rows, on ? src : 0, dim, dim, sub);
                if (!on) wt[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    gsum.v[j].x = fmaf(wt[u], y[u].v[j].x, gsum.v[j].x); gsum.v[j].y = fmaf(wt[u], y[u].v[j].y, gsum.v[j].y);
                    gsum.v[j].z = fmaf(wt[u], y[u].v[j].z, gsum.v[j].z); gsum.v[j].w = fmaf(wt[u], y[u].v[j].w, gsum.v[j].w);
                }
        }
    }
    normalize_bwd4<NV>(x, gsum);
    if (owner && !hub) {
        float* out = is_rel ? d_rel + (int64_t)(dest - n_ent) * dim : d_ent + (int64_t)dest * dld;
        st_row4<NV>(out, 0, 0, dim, sub, gsum);
    }
}

// ---- pass C: hot rows -----------------------------------------------------------------------------------------------------------------
// One WORKGROUP per hot row (a persistent grid walks the list): its sixteen 16-lane groups share the row's chunk partials -- group i takes
// chunks i, i + 16, ... -- and meet in LDS; 64-bit integer sums are order-free, so the result does not depend on who added what.  (One group
// per hot row walked a Zipf(1.0) batch's hottest row -- 9,200 entries, 144 chunks -- alone: 215 us.)
template <int NV, typename TabT>
__global__ __launch_bounds__(256) void touched_hub_kernel(const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim, int n_ent,
                                                          float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel, TouchedWork w) {
    __shared__ long long red[16][16][4 * NV];
    __shared__ long long tot[16][4 * NV];
    const int lane = threadIdx.x & 63, sub = lane & 15, gbase = lane & 48;
    const int gi = threadIdx.x >> 4;                       // group inside the workgroup
    const unsigned which = w.hdr->cur & 1u;
    const unsigned nh = min(w.hdr->n_hubs[which], w.max_hubs);
    const unsigned long long* keys = w.hkeys[which];
    for (unsigned hb_i = blockIdx.x; hb_i < nh; hb_i += gridDim.x) {
        const TouchedHub hb = w.hubs[hb_i];
        const int dest = hb.dest;
        const bool is_rel = dest >= n_ent;
        const unsigned n = w.cnt[dest];
        const unsigned over = n > (unsigned)kTCap ? n - (unsigned)kTCap : 0u;
        long long acc[4 * NV];
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) acc[i] = 0;
        if (gi == 0) {                                     // the owner's own entry and the bucket's slots 1..15
            const TouchedRec rc = w.rec[hb.rec];
            fix_add_row<NV>(acc, ld_row4<NV>(w.rows, rc.src, dim, dim, sub), rc.w);
            fix_add_run<NV, 2>(acc, w.list + (int64_t)dest * kTCap + 1, (unsigned)kTCap - 1u, w.rows, dim, sub, gbase);
        }
        for (unsigned j = (unsigned)gi; j * kTChunk < over; j += 16u) {          // the chunk partials pass B' left
            const unsigned hslot = touched_find(keys, w.hmask, dest, j);
            const long long* pr = w.hpart + (int64_t)hslot * dim;
#pragma unroll
            for (int jv = 0; jv < NV; ++jv) {
                const int c = (sub + 16 * jv) * 4;
                if (c < dim) {
                    const longlong2 a = reinterpret_cast<const longlong2*>(pr + c)[0], b2 = reinterpret_cast<const longlong2*>(pr + c)[1];
                    acc[4 * jv + 0] += a.x; acc[4 * jv + 1] += a.y; acc[4 * jv + 2] += b2.x; acc[4 * jv + 3] += b2.y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) red[gi][sub][i] = acc[i];
        __syncthreads();
        // the sixteen groups' partial rows summed by ALL threads: thread (gi, sub) takes components gi, gi + 16, ... of lane sub's share
        // (integer sums: any order gives the same bits); group 0 collects the totals
        for (int i = gi; i < 4 * NV; i += 16) {
            long long t = 0;
#pragma unroll 4
            for (int g2 = 0; g2 < 16; ++g2) t += red[g2][sub][i];
            tot[sub][i] = t;
        }
        __syncthreads();
        if (gi == 0) {
            if (sub == 0) w.cnt[dest] = 0u;                // ready for the next launch
#pragma unroll
            for (int i = 0; i < 4 * NV; ++i) acc[i] = tot[sub][i];
            const TabT* xrow = is_rel ? rel + (int64_t)(dest - n_ent) * dim : ent + (int64_t)dest * eld;
            Row4<NV> x = ld_row4<NV>(xrow, 0, 0, dim, sub);
            Row4<NV> gsum;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                gsum.v[j].x = (float)((double)acc[4 * j + 0] * (1.0 / kFix)); gsum.v[j].y = (float)((double)acc[4 * j + 1] * (1.0 / kFix));
                gsum.v[j].z = (float)((double)acc[4 * j + 2] * (1.0 / kFix)); gsum.v[j].w = (float)((double)acc[4 * j + 3] * (1.0 / kFix));
            }
            normalize_bwd4<NV>(x, gsum);
            float* out = is_rel ? d_rel + (int64_t)(dest - n_ent) * dim : d_ent + (int64_t)dest * dld;
            st_row4<NV>(out, 0, 0, dim, sub, gsum);
        }
        __syncthreads();                                   // the LDS partials are the next hot row's
    }
}

template <typename TabT>
static int launch_touched(const char* who, const TabT* ent, int64_t ent_ld, int64_t n_ent, const TabT* rel, int64_t n_rel, int dim, const void* h, const void* r,
                          const void* t, const void* neg_t, int idx_bytes, int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                          float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld, float* d_rel, void* work, int64_t work_bytes, hipStream_t st) {
    SNAG_REQUIRE(ent && rel && d_ent && d_rel && work && (batch == 0 || loss_part), SNAG_ERR_ARG, "%s: null table / gradient / workspace / loss_part", who);
    SNAG_REQUIRE(n_ent > 0 && n_rel > 0 && dim > 0 && batch >= 0 && n_neg >= 0, SNAG_ERR_ARG, "%s: bad sizes", who);
    SNAG_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SNAG_ERR_ARG, "%s: idx_bytes=%d (want 4 or 8)", who, idx_bytes);
    SNAG_REQUIRE(dim <= 512 && dim % 4 == 0 && ent_ld % 4 == 0 && d_ent_ld % 4 == 0 && ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_UNSUPPORTED,
                 "%s: dim=%d (<= 512) and the row strides must be multiples of 4 (16-byte fp32 / 8-byte bf16 rows)", who, dim);
    SNAG_REQUIRE(n_ent + n_rel < (1LL << 27) && batch * (int64_t)(n_neg + 12) < (1LL << 31), SNAG_ERR_UNSUPPORTED, "%s: n_ent + n_rel >= 2^27 or batch * n_neg >= 2^31", who);
    const size_t need = touched_work_bytes(n_ent, n_rel, batch, n_neg, dim);
    SNAG_REQUIRE(work_bytes >= (int64_t)need, SNAG_ERR_ARG, "%s: workspace holds %ld bytes, need %ld", who, (long)work_bytes, (long)need);
    constexpr uintptr_t tab_align = sizeof(TabT) == 4 ? 15 : 7;
    SNAG_REQUIRE(((uintptr_t)work & 255) == 0 && ((uintptr_t)ent & tab_align) == 0 && ((uintptr_t)rel & tab_align) == 0 && ((uintptr_t)d_ent & 15) == 0 &&
                 ((uintptr_t)d_rel & 15) == 0, SNAG_ERR_UNSUPPORTED, "%s: tables and gradients must be 16-byte aligned (bf16 tables: 8), the workspace 256-byte aligned", who);
    SNAG_REQUIRE(batch == 0 || (h && r && t && pos && (n_neg == 0 || (neg && neg_t))), SNAG_ERR_ARG, "%s: null index or score pointer", who);
    const TouchedWork w = touched_carve(work, n_ent, n_rel, batch, n_neg, dim);
    const int nv = (dim + 63) / 64;
    const dim3 block(256);
#define SNAG_NV_SWITCH(CALL)          \
    switch (nv) {                     \
        case 1: CALL(1); break;       \
        case 2: CALL(2); break;       \
        case 3: CALL(3); break;       \
        case 4: CALL(4); break;       \
        case 5: case 6: CALL(6); break; \
        default: CALL(8); break;      \
    }
    if (batch > 0) {
        const dim3 grid((unsigned)ceil_div(batch * w.parts, 4));
        const bool small_agg = 4 * (w.per + 3) <= 160;
#define SNAG_TA_(IDX, NV, AGG) hipLaunchKernelGGL((touched_a_kernel<IDX, NV, TabT, AGG>), grid, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, \
                             (const IDX*)h, (const IDX*)r, (const IDX*)t, (const IDX*)neg_t, batch, n_neg, margin, grad_scale, pos, neg, w)
#define SNAG_TA(NV) do { if (idx_bytes == 4) { if (small_agg) SNAG_TA_(int32_t, NV, 256); else SNAG_TA_(int32_t, NV, 1024); } \
                         else { if (small_agg) SNAG_TA_(int64_t, NV, 256); else SNAG_TA_(int64_t, NV, 1024); } } while (0)
        SNAG_NV_SWITCH(SNAG_TA)
#undef SNAG_TA
#undef SNAG_TA_
        int rc = check_launch(who);
        if (rc) return rc;
    }
    // pass B: one group per record, + 1 workgroup that sums the losses (an empty batch: loss = 0), + 8 that wipe the previous launch's hash keys
    const int group_blocks = ceil_div(w.n_rec * 16, 256);
    const int chunk_blocks = batch > 0 ? 512 : 0;                   // pass B': a fixed grid of 2,048 waves walks the list of claimed chunks
    const dim3 gridb((unsigned)(chunk_blocks + group_blocks) + 9u);
#define SNAG_TB(NV) hipLaunchKernelGGL((touched_b_kernel<NV, TabT>), gridb, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, d_ent, d_ent_ld, d_rel, w, loss_part, batch, loss, \
                                       group_blocks, chunk_blocks)
    SNAG_NV_SWITCH(SNAG_TB)
#undef SNAG_TB
    int rc = check_launch(who);
    if (rc || batch == 0) return rc;
    const dim3 gridh((unsigned)std::min<int64_t>(SNAG_HUB_GRID, (int64_t)w.max_hubs));
#define SNAG_TH(NV) hipLaunchKernelGGL((touched_hub_kernel<NV, TabT>), gridh, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, d_ent, d_ent_ld, d_rel, w)
    SNAG_NV_SWITCH(SNAG_TH)
#undef SNAG_TH
#undef SNAG_NV_SWITCH
    return check_launch(who);
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int64_t snag_score_touched_work_bytes(int64_t n_ent, int64_t n_rel, int64_t batch, int n_neg, int dim) {
    if (n_ent < 1 || n_rel < 1 || batch < 0 || n_neg < 0 || dim < 1) return 0;
    return (int64_t)touched_work_bytes(n_ent, n_rel, batch, n_neg, dim);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_touched(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                          const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                          int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                          float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld, float* d_rel,
                                                          void* work, int64_t work_bytes, void* stream) {
    return launch_touched<float>("snag_score_margin_fwd_bwd_touched", ent, ent_ld, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg, margin,
                                 grad_scale, pos, neg, loss_part, loss, d_ent, d_ent_ld, d_rel, work, work_bytes, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_touched_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                                                               const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                               int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                               float* loss_part, float* loss, float* d_ent, int64_t d_ent_ld, float* d_rel,
                                                               void* work, int64_t work_bytes, void* stream) {
    return launch_touched<uint16_t>("snag_score_margin_fwd_bwd_touched_bf16", ent, ent_ld, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg,
                                    margin, grad_scale, pos, neg, loss_part, loss, d_ent, d_ent_ld, d_rel, work, work_bytes, (hipStream_t)stream);
}
