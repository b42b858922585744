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
//            A negative tail's 16-lane group has the tail's row AND q_b in registers when it scores it, i.e. everything the gradient row of a
//            destination with ONE entry is made of: it writes that row -- normalize_bwd(x, w q_b), statement for statement pass B's
//            single-entry arithmetic -- straight into d_ent (round 5).  Nearly every destination of a uniformly drawn batch has one entry,
//            so pass B no longer re-reads the 69 MB of table rows pass A just gathered, nor the q rows, nor writes 137 MB; where more
//            entries arrive the row written here is provisional -- ONE writer per row, the entry that reserved slot 0 (nothing races) --
//            and pass B overwrites it.
//   pass B   one 16-lane group per RECORD.  The record whose slot is 0 owns its destination: with one entry that pass A has written
//            (flag in the record) it only resets the counter -- two dependent round trips: the record, then the counter --; otherwise a third
//            one fetches {own table row, the entry's row, the bucket} and with up to 16 entries it adds them in a canonical order (sorted
//            fp32); hot rows are handed to pass C.  Records with other slots leave at once.  The per-part losses are summed here too.
//   relations (round 5): a 1,024-triple batch sends 4 x 1,024 entries to ~1,000 relation rows -- every one of them a multi-entry destination and
//            a few of them "hot" (> 16 entries): on uniform entity indices the hot-row launch existed for four relations (16 us of dependent
//            round trips).  While n_rel x batch is small the relation rows take no part in the records at all: one 16-lane group per relation
//            (a role of pass B's launch) scans the batch's relation ids, finds its triples in batch order and sums their dq rows -- the order
//            and the arithmetic (sorted fp32 up to 16 entries, fixed point beyond) of the generic path, so the same bits -- and writes EVERY
//            relation's row (zeros where no triple names it: the caller no longer clears d_rel).
//   pass C   one 16-lane group per hot row: bucket + chunk partials in fixed point (order-free), normalisation backward, store.
// Only touched rows of d_ent are written (the caller keeps the rest zero); every row of d_rel is.
#include <algorithm>

#include "seg_rows.hip.h"

namespace snag {

#ifndef SNAG_TA_WAVES
#define SNAG_TA_WAVES 2                  // (waves per SIMD pass A is compiled for: 2 = up to 256 registers, no spills: 46 -> 38 us against 3)
#endif
#ifndef SNAG_TB_WAVES
#define SNAG_TB_WAVES 3                  // (162 registers with four rows per gather step: the launch is a handful of dependent round trips per wave, not occupancy-bound)
#endif
#ifndef SNAG_HUB_GRID
#define SNAG_HUB_GRID 1024               // (hot-row launch of a Zipf(1.0) batch: 512 workgroups 34.5 us, 1,024: 31.8, 2,048: 31.7)
#endif
#ifndef SNAG_T_COMBINE
#define SNAG_T_COMBINE 1                 // a triple's parts meet in LDS: one head / tail / relation entry per triple (0: one per part; A/B)
#endif
#ifndef SNAG_T_ABLATE
#define SNAG_T_ABLATE 0                  // timing experiments (wrong results): 1 no hot-row launch, 2 no pass B launch, 4 pass A stores no gradient rows, 8 pass B resets no counters,
                                         // 16 pass B does no arithmetic, 32 no relation role, 64 no chunk-sum role
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
    unsigned slot;                          // | kRecWritten: pass A has written the destination's gradient row as if this were its only entry
};
constexpr unsigned kRecWritten = 0x80000000u;
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
    int rel_role;                           // the relation rows are summed by pass B's relation role: pass A emits no relation entries
    int combine;                            // the parts of a triple sit in ONE workgroup (2 or 4 parts): they meet in LDS and the triple sends ONE head, ONE
                                            // tail (and one relation) entry instead of one per part
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
__device__ __forceinline__ void touched_emit(const TouchedWork& w, unsigned which, int64_t rec_idx, int dest, unsigned slot, int src, float weight,
                                             unsigned flags = 0u) {
    w.rec[rec_idx] = TouchedRec{dest, src, weight, slot | flags};
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
    int64_t batch, int n_neg_all, float margin, float gscale, float* __restrict__ pos, float* __restrict__ neg, TouchedWork w,
    float* __restrict__ d_ent, int64_t dld) {
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
    const int n3 = (w.combine && part != 0) ? 0 : (w.rel_role ? 2 : 3);      // (relation role: the relation's entry is a void record; combined parts: part 0 speaks for the triple)
    if (live && lane < n3) agg_insert(dest3, local3, aslot3);
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
    constexpr int RB = NV >= 3 ? 4 : 8;       // (round 5, with the row stores in the loop: 2 or 8 rounds, or the next batch requested before the current one
                                              //  is scored -- register double-buffered --, all measured slower: 54.9 / 60.9 / 52.4-85.2 us against 46.2)
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
    // the reserved runs have arrived by now -- their round trip went out in front of the head / relation / tail gather the arithmetic above
    // waited for: the workgroup meets here, and every negative then knows its slot BEFORE its row is scored (slot 0 writes the row)
    __syncthreads();
    for (int k0 = 0; k0 < w.per; k0 += 64) {
        const int cnt = max(0, min(64, n_neg - k0));
        const int my = (lane < cnt) ? (int)nrow[k0 + lane] : 0;
        float my_s = 0.f;                               // lane k: score of negative k0 + k
        const int qb = k0 >> 6;
        const unsigned nl = qb == 0 ? nlocal[0] : (qb == 1 ? nlocal[1] : (qb == 2 ? nlocal[2] : nlocal[3]));
        const int na = qb == 0 ? naslot[0] : (qb == 1 ? naslot[1] : (qb == 2 ? naslot[2] : naslot[3]));
        const unsigned my_slot = (lane < cnt) ? agg_base[na] + nl : 1u;      // lane k: arrival number of negative k0 + k at its destination
        auto issue = [&](int it0, Row4<NV> (&x)[RB]) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                    const int k = it0 + 4 * u + grp;
                    const int idx = __shfl(my, k & 63);
                    x[u] = ld_row4<NV>(ent, (k < cnt) ? idx : 0, eld, dim, sub);
                }
        };
        auto score = [&](int it0, Row4<NV> (&x)[RB]) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int k = it0 + 4 * u + grp;
                const bool valid = k < cnt;
                const float den = l2_denominator(sum16(sumsq4<NV>(x[u])));
                const float inv = 1.0f / den;
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) {          // x[u] <- x^ ; s = |q - x^|^2
                    float d;
                    x[u].v[j].x *= inv; d = q.v[j].x - x[u].v[j].x; s = fmaf(d, d, s);
                    x[u].v[j].y *= inv; d = q.v[j].y - x[u].v[j].y; s = fmaf(d, d, s);
                    x[u].v[j].z *= inv; d = q.v[j].z - x[u].v[j].z; s = fmaf(d, d, s);
                    x[u].v[j].w *= inv; d = q.v[j].w - x[u].v[j].w; s = fmaf(d, d, s);
                }
                s = sum16(s);
                const float viol = margin + s_pos - s;
                const float a = (valid && viol > 0.f) ? gscale : 0.f;      // uniform inside the 16-lane group
                hinge += valid ? fmaxf(viol, 0.f) : 0.f;
                c_pos += a;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    acc.v[j].x = fmaf(a, q.v[j].x - x[u].v[j].x, acc.v[j].x); acc.v[j].y = fmaf(a, q.v[j].y - x[u].v[j].y, acc.v[j].y);
                    acc.v[j].z = fmaf(a, q.v[j].z - x[u].v[j].z, acc.v[j].z); acc.v[j].w = fmaf(a, q.v[j].w - x[u].v[j].w, acc.v[j].w);
                }
                // the tail's gradient row as if this entry were its destination's only one: normalize_bwd4(x, w q_b) with the entry's weight
                // (2 gscale, or 0 for an inactive hinge) -- pass B's single-entry arithmetic on the same values, so the same bits
                {
                    const float w0 = (viol > 0.f) ? 2.f * gscale : 0.f;
                    Row4<NV> gr;
                    float dot = 0.f;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        gr.v[j] = make_float4(w0 * q.v[j].x, w0 * q.v[j].y, w0 * q.v[j].z, w0 * q.v[j].w);
                        dot = fmaf(x[u].v[j].x, gr.v[j].x, dot); dot = fmaf(x[u].v[j].y, gr.v[j].y, dot);
                        dot = fmaf(x[u].v[j].z, gr.v[j].z, dot); dot = fmaf(x[u].v[j].w, gr.v[j].w, dot);
                    }
                    dot = sum16(dot);
                    if (!(den > 1e-12f)) dot = 0.f;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        gr.v[j].x = (gr.v[j].x - x[u].v[j].x * dot) * inv; gr.v[j].y = (gr.v[j].y - x[u].v[j].y * dot) * inv;
                        gr.v[j].z = (gr.v[j].z - x[u].v[j].z * dot) * inv; gr.v[j].w = (gr.v[j].w - x[u].v[j].w * dot) * inv;
                    }
                    const int idx = __shfl(my, k & 63);
                    const unsigned slot_k = __shfl(my_slot, k & 63);
                    // the FIRST entry to arrive at a destination writes: one writer per row (a Zipf batch's hot rows would otherwise take
                    // thousands of write-through stores each: pass A 153 us instead of 53)
                    // (write-through -- sc1 -- stores for these rows, so that the launch does not end on an L2 full of dirty lines: measured
                    // 2-3 % slower on the same box, 72.3-73.3 against 70.5-70.8 us for the entry)
                    if (valid && slot_k == 0u && !(SNAG_T_ABLATE & 4)) st_row4<NV>(d_ent, idx, dld, dim, sub, gr);
                }
                const float sk = __shfl(s, (lane & 3) * 16);
                if (((lane - it0) >> 2) == u && lane >= it0 && lane < it0 + 4 * RB) my_s = sk;
            }
                };
        for (int it0 = 0; it0 < cnt; it0 += 4 * RB) {
            Row4<NV> x[RB];
            issue(it0, x);
            score(it0, x);
        }
        // lane k: score out, its negative's record (slot = its destination's base + its local number; weight 2 a_k, zero for an inactive
        // hinge); slots of this id block past the part's negatives are void records
        if (lane < cnt) {
            neg[b * (int64_t)n_neg_all + k_lo + k0 + lane] = my_s;
            touched_emit(w, which, rec0 + 3 + k0 + lane, my, my_slot, (int)b, (margin + s_pos - my_s > 0.f) ? 2.f * gscale : 0.f, my_slot == 0u ? kRecWritten : 0u);
        } else if (live && k0 + lane < w.per) {
            w.rec[rec0 + 3 + k0 + lane] = TouchedRec{-1, 0, 0.f, 1u};
        }
    }
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
    // COMBINED (2 or 4 parts = the waves of this workgroup): the parts' sums meet in LDS in part order and part 0 writes ONE row
    // rows[batch + b] = dq_b for the triple -- head, tail and relation then receive one entry per triple, not one per part (a head dealt to four
    // parts was a four-entry destination for pass B: list fetch, sort, four rows)
    __shared__ float comb[4][64 * NV];
    __shared__ float ccomb[4];
    if (w.combine) {
        if (grp == 1) {
#pragma unroll
            for (int j = 0; j < NV; ++j) *reinterpret_cast<float4*>(&comb[wave][(sub + 16 * j) * 4]) = acc.v[j];
            if (sub == 0) ccomb[wave] = c_pos;
        }
        __syncthreads();
        if (live && part == 0) {          // (a wave past the batch's end has part == 0 too; wave + pp would leave the four rows)
            c_pos = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) acc.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int pp = 0; pp < w.parts; ++pp) {
                c_pos += ccomb[wave + pp];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(&comb[wave + pp][(sub + 16 * j) * 4]);
                    acc.v[j].x += v.x; acc.v[j].y += v.y; acc.v[j].z += v.z; acc.v[j].w += v.w;
                }
            }
        }
    }
    const bool speaks = !w.combine || part == 0;                   // this wave emits the triple's (or its part's) head / tail / relation entries
    const int64_t dq_row = w.combine ? batch + b : batch + vb;
    if (live && grp == 0 && part == 0) st_row4<NV>(w.rows, b, dim, dim, sub, q);
    if (live && grp == 1 && speaks) {
        Row4<NV> dq;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            dq.v[j].x = 2.f * (c_pos * (q.v[j].x - tt.v[j].x) - acc.v[j].x); dq.v[j].y = 2.f * (c_pos * (q.v[j].y - tt.v[j].y) - acc.v[j].y);
            dq.v[j].z = 2.f * (c_pos * (q.v[j].z - tt.v[j].z) - acc.v[j].z); dq.v[j].w = 2.f * (c_pos * (q.v[j].w - tt.v[j].w) - acc.v[j].w);
        }
        st_row4<NV>(w.rows, dq_row, dim, dim, sub, dq);
    }
    if (!live) return;
    if (lane < n3) touched_emit(w, which, rec0 + lane, dest3, agg_base[aslot3] + local3, lane == 0 ? (int)b : (int)dq_row, lane == 0 ? -2.f * c_pos : 1.f);
    else if (lane < 3) w.rec[rec0 + lane] = TouchedRec{-1, 0, 0.f, 1u};
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

// ---- relation role of pass B's launch: one 16-lane group per relation ------------------------------------------------------------------
template <int NV, typename TabT>
__device__ __forceinline__ void touched_rel_body(const TabT* __restrict__ rel, int dim, int n_rel, const void* __restrict__ r_idx, int idx_bytes, int64_t batch,
                                                 float* __restrict__ d_rel, const TouchedWork& w, int vblock) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4, gbase = lane & 48;
    const int rho = (vblock * 4 + (int)(threadIdx.x >> 6)) * 4 + grp;
    const bool live = rho < n_rel;
    const int parts = w.combine ? 1 : w.parts;                     // dq rows per triple
    // scan 1: the triples that name this relation, in batch order; lane `sub` keeps the sub-th one (16 are all the sorted fp32 path can take).
    // Eight 64-id blocks are requested together: one round trip per 512 triples, not eight.
    int cnt = 0, mine_b = 0;
    auto rid = [&](int64_t i) { return i < batch ? (idx_bytes == 4 ? reinterpret_cast<const int32_t*>(r_idx)[i] : (int)reinterpret_cast<const int64_t*>(r_idx)[i]) : -1; };
    for (int64_t b0 = 0; b0 < batch; b0 += 512) {
        int rv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rv[q] = rid(b0 + 64 * q + lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int rho_g = __shfl(rho, 16 * g4);
                unsigned long long m = __ballot(rv[q] == rho_g);
                if (grp == g4) {
                    while (m != 0ull) {
                        if (cnt == sub) mine_b = (int)b0 + 64 * q + __builtin_ctzll(m);
                        ++cnt;
                        m &= m - 1ull;
                    }
                }
            }
        }
    }
    const int n = live ? cnt * parts : 0;                          // entries: one dq row per (triple, part), weight 1
    Row4<NV> x = ld_row4<NV>(rel, live ? rho : 0, dim, dim, sub);
    Row4<NV> gsum;
#pragma unroll
    for (int j = 0; j < NV; ++j) gsum.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool small = n <= kTCap;
    if (__ballot(small && n > 0) != 0ull) {
        // up to 16 entries: fp32 sums in ascending (triple, part) order = the generic path's sorted-key order (key = entry row, weight 1)
        int wave_max = small ? n : 0;
        wave_max = max(wave_max, __shfl_xor(wave_max, 16));
        wave_max = max(wave_max, __shfl_xor(wave_max, 32));
        for (int e0 = 0; e0 < wave_max; e0 += 4) {
            Row4<NV> y[4];
            float wt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u;
                const bool on = small && e < n;
                const int bb = __shfl(mine_b, gbase + ((e / parts) & 15));
                wt[u] = on ? 1.f : 0.f;
                y[u] = ld_row4<NV>(w.rows, on ? batch + (int64_t)bb * parts + e % parts : 0, dim, dim, sub);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    gsum.v[j].x = fmaf(wt[u], y[u].v[j].x, gsum.v[j].x); gsum.v[j].y = fmaf(wt[u], y[u].v[j].y, gsum.v[j].y);
                    gsum.v[j].z = fmaf(wt[u], y[u].v[j].z, gsum.v[j].z); gsum.v[j].w = fmaf(wt[u], y[u].v[j].w, gsum.v[j].w);
                }
        }
    }
    if (__ballot(!small) != 0ull) {
        // more than 16 entries: 64-bit fixed point (order-free), as the hot-row pass sums them
        long long acc[4 * NV];
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) acc[i] = 0;
        const bool mid = !small && cnt <= 16;                     // its triples are all in the group's lanes: four rows per step
        if (__ballot(mid) != 0ull) {
            int wave_max = mid ? n : 0;
            wave_max = max(wave_max, __shfl_xor(wave_max, 16));
            wave_max = max(wave_max, __shfl_xor(wave_max, 32));
            for (int e0 = 0; e0 < wave_max; e0 += 4) {
                Row4<NV> y[4];
                float wt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u;
                    const bool on = mid && e < n;
                    const int bb = __shfl(mine_b, gbase + ((e / parts) & 15));
                    wt[u] = on ? 1.f : 0.f;
                    y[u] = ld_row4<NV>(w.rows, on ? batch + (int64_t)bb * parts + e % parts : 0, dim, dim, sub);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) fix_add_row<NV>(acc, y[u], wt[u]);
            }
        }
        const bool big = live && cnt > 16;                        // (a relation named by more than 16 triples of the batch: scan 2 walks the batch again)
        if (__ballot(big) != 0ull) {
            for (int64_t b0 = 0; b0 < batch; b0 += 64) {
                const int rv = rid(b0 + lane);
                const int want = big ? rho : -2;
                unsigned long long mg[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) mg[g4] = __ballot(rv == __shfl(want, 16 * g4));
                unsigned long long m = grp == 0 ? mg[0] : (grp == 1 ? mg[1] : (grp == 2 ? mg[2] : mg[3]));
                unsigned long long any = mg[0] | mg[1] | mg[2] | mg[3];
                while (any != 0ull) {                                     // (uniform trip count: the longest group's)
                    const bool on = m != 0ull;
                    const int64_t bb = b0 + (on ? __builtin_ctzll(m) : 0);
                    for (int p0 = 0; p0 < parts; p0 += 4) {
                        Row4<NV> y[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) y[u] = ld_row4<NV>(w.rows, (on && p0 + u < parts) ? batch + bb * parts + p0 + u : 0, dim, dim, sub);
#pragma unroll
                        for (int u = 0; u < 4; ++u) fix_add_row<NV>(acc, y[u], (on && p0 + u < parts) ? 1.f : 0.f);
                    }
                    m &= m - 1ull;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) mg[g4] &= mg[g4] - (mg[g4] != 0ull ? 1ull : 0ull);
                    any = mg[0] | mg[1] | mg[2] | mg[3];
                }
            }
        }
        if (!small) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                gsum.v[j].x = (float)((double)acc[4 * j + 0] * (1.0 / kFix)); gsum.v[j].y = (float)((double)acc[4 * j + 1] * (1.0 / kFix));
                gsum.v[j].z = (float)((double)acc[4 * j + 2] * (1.0 / kFix)); gsum.v[j].w = (float)((double)acc[4 * j + 3] * (1.0 / kFix));
            }
        }
    }
    normalize_bwd4<NV>(x, gsum);
    if (live) {
        if (n == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) gsum.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        st_row4<NV>(d_rel, rho, dim, dim, sub, gsum);
    }
}

// ---- pass B: one group per record -----------------------------------------------------------------------------------------------------
template <int NV, typename TabT>
__global__ __launch_bounds__(256, SNAG_TB_WAVES) void touched_b_kernel(const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim, int n_ent,
                                                        float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel, TouchedWork w,
                                                        float* __restrict__ loss_part, int64_t batch, float* __restrict__ loss, int group_blocks,
                                                        int chunk_blocks, int rel_blocks, int n_rel, const void* __restrict__ r_idx, int idx_bytes) {
    // workgroup roles: [0, chunk_blocks) chunk sums of the overflow hash (pass B'), then one group per record, then the loss sum, then
    // the wipe of the previous launch's hash keys
    if ((int)blockIdx.x < rel_blocks) {
        if (!(SNAG_T_ABLATE & 32)) touched_rel_body<NV, TabT>(rel, dim, n_rel, r_idx, idx_bytes, batch, d_rel, w, (int)blockIdx.x);
        return;
    }
    if ((int)blockIdx.x - rel_blocks < chunk_blocks) {
        if (!(SNAG_T_ABLATE & 64)) touched_chunk_body<NV>(w, dim, (int)blockIdx.x - rel_blocks, chunk_blocks);
        return;
    }
    const int bid = (int)blockIdx.x - rel_blocks - chunk_blocks;
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
        const unsigned nb = gridDim.x - rel_blocks - chunk_blocks - group_blocks - 1u;
        for (unsigned i = ((unsigned)bid - group_blocks - 1u) * 256u + threadIdx.x; i <= w.hmask; i += nb * 256u) keys[i] = 0ull;
        return;
    }
    if (bid == 0 && threadIdx.x == 0) {               // hand the other table / hub counter to the next launch
        w.hdr->n_hubs[which ^ 1u] = 0u;
        w.hdr->n_chunks[which ^ 1u] = 0u;
        w.hdr->phase = which ^ 1u;
    }
    const int lane = threadIdx.x & 63, sub = lane & 15, gbase = lane & 48, grp = lane >> 4;
    // CLASSIFICATION, one LANE per record (round 5; a 16-lane group per record made this launch 8,400 workgroups -- eight rounds of
    // two dependent round trips -- for the one record in twelve that has arithmetic to do).  Round trip 1: the record; round trip 2, for
    // the record that owns its destination (slot 0): how many entries arrived.  One entry whose row pass A has written: reset the counter, done.
    const int64_t g = (int64_t)bid * 256 + threadIdx.x;
    const TouchedRec rc = g < w.n_rec ? w.rec[g] : TouchedRec{-1, 0, 0.f, 1u};
    const bool owner = rc.dest >= 0 && (rc.slot & ~kRecWritten) == 0u;
    if (__ballot(owner) == 0ull) return;
    const unsigned n1 = owner ? w.cnt[rc.dest] : 0u;
    const bool hub1 = n1 > (unsigned)kTCap;
    if (owner && !hub1 && !(SNAG_T_ABLATE & 8)) w.cnt[rc.dest] = 0u;                  // ready for the next launch (hot rows: pass C)
    // hot rows go to pass C: one list slot per wave-instruction
    {
        const unsigned long long hm = __ballot(owner && hub1);
        if (hm != 0ull) {
            const int leader = __builtin_ctzll(hm);
            unsigned base = 0u;
            if (lane == leader) base = atomicAdd(&w.hdr->n_hubs[which], (unsigned)__builtin_popcountll(hm));
            base = __shfl(base, leader);
            if (owner && hub1) {
                const unsigned at = base + (unsigned)__builtin_popcountll(hm & ((1ull << lane) - 1ull));
                if (at < w.max_hubs) w.hubs[at] = TouchedHub{rc.dest, (unsigned)g};
            }
        }
    }
    // ARITHMETIC, one 16-lane group per record that has some: its destination received 2..16 entries, or one that pass A did not write
    unsigned long long todo_mask = (SNAG_T_ABLATE & 16) ? 0ull : __ballot(owner && !hub1 && !(n1 == 1u && (rc.slot & kRecWritten) != 0u));
    while (todo_mask != 0ull) {
        // the next four such records, one per group (a group past the last one idles along: its loads go to row 0, it stores nothing)
        unsigned long long mm = todo_mask;
        int src_lane = 0;
        bool todo = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool have = mm != 0ull;
            const int l = have ? __builtin_ctzll(mm) : 0;
            if (i == grp) { src_lane = l; todo = have; }
            mm &= mm - 1ull;
        }
        todo_mask = mm;
        // (every lane takes part in the exchanges -- the record's lane may sit in another group -- and selects afterwards)
        const int dest_x = __shfl(rc.dest, src_lane);
        const int rsrc = __shfl(rc.src, src_lane);
        const float rw = __shfl(rc.w, src_lane);
        const unsigned n_x = __shfl(n1, src_lane);
        const int dest = todo ? dest_x : 0;
        const unsigned n = todo ? n_x : 0u;
        const bool is_rel = dest >= n_ent;
        // round trip 3: the destination's own row, the owner entry's row, the rest of the bucket
        // (one load sequence through a selected pointer: as two branches the compiler waits for the row inside its branch, before the
        // other loads are even issued)
        const TabT* xrow = !todo ? ent : (is_rel ? rel + (int64_t)(dest - n_ent) * dim : ent + (int64_t)dest * eld);
        Row4<NV> x = ld_row4<NV>(xrow, 0, 0, dim, sub);
        const Row4<NV> y0 = ld_row4<NV>(w.rows, todo ? rsrc : 0, dim, dim, sub);
        Row4<NV> gsum;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float w0 = (todo && n == 1u) ? rw : 0.f;
            gsum.v[j] = make_float4(w0 * y0.v[j].x, w0 * y0.v[j].y, w0 * y0.v[j].z, w0 * y0.v[j].w);
        }
        const bool multi = todo && n > 1u;
        if (__ballot(multi) != 0ull) {
            // (the bucket's slots 1..15: destinations that received more than one entry -- ~6 % of a uniform batch's, and every head / relation /
            // positive tail of a triple dealt to several parts)
            SegEntry mine = multi ? w.list[(int64_t)dest * kTCap + sub] : SegEntry{0, 0.f};
            if (sub == 0) mine = SegEntry{rsrc, rw};                  // slot 0 is the record itself
            // 2..16 entries: canonical order by (row, weight bits, slot) key -- the slot order the atomics produced varies from run to run, the
            // sorted order does not --, fp32 sums in that order, two rows gathered per step
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
            for (int r0 = 0; r0 < wave_max; r0 += 4) {                 // four rows per step: a head or tail dealt to four parts is ONE round trip
                Row4<NV> y[4];
                float wt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long m = __ballot(sub < nf && rank == r0 + u) >> gbase;
                    const int pick = gbase + (int)__builtin_ctzll(m | 0x10000ull);
                    const int src = __shfl(mine.src, pick);
                    const bool on = r0 + u < nf;
                    wt[u] = on ? __shfl(mine.w, pick) : 0.f;
                    y[u] = ld_row4<NV>(w.rows, on ? src : 0, dim, dim, sub);
                    if (!on) wt[u] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        gsum.v[j].x = fmaf(wt[u], y[u].v[j].x, gsum.v[j].x); gsum.v[j].y = fmaf(wt[u], y[u].v[j].y, gsum.v[j].y);
                        gsum.v[j].z = fmaf(wt[u], y[u].v[j].z, gsum.v[j].z); gsum.v[j].w = fmaf(wt[u], y[u].v[j].w, gsum.v[j].w);
                    }
            }
        }
        normalize_bwd4<NV>(x, gsum);
        if (todo) {
            float* out = is_rel ? d_rel + (int64_t)(dest - n_ent) * dim : d_ent + (int64_t)dest * dld;
            st_row4<NV>(out, 0, 0, dim, sub, gsum);
        }
    }
}

// ---- pass C: hot rows -----------------------------------------------------------------------------------------------------------------
// One WORKGROUP per hot row (a persistent grid walks the list): its sixteen 16-lane groups share the row's inline entries (one each) and
// chunk partials -- group i takes chunks i, i + 16, ... -- and meet in LDS; 64-bit integer sums are order-free, so the result does not depend on who added what.  (One group
// per hot row walked a Zipf(1.0) batch's hottest row -- 9,200 entries, 144 chunks -- alone: 215 us.)
template <int NV, typename TabT>
__global__ __launch_bounds__(256) void touched_hub_kernel(const TabT* __restrict__ ent, int64_t eld, const TabT* __restrict__ rel, int dim, int n_ent,
                                                          float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel, TouchedWork w) {
    __shared__ long long red[16][16][4 * NV];
    __shared__ long long tot[16][4 * NV];
    const int lane = threadIdx.x & 63, sub = lane & 15;
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
        {   // the sixteen inline entries, ONE per group (group 0: the owner's own record; group i: bucket slot i): two dependent round trips for
            // all of them (group 0 used to walk them alone: one entry fetch + eight gather steps behind each other)
            SegEntry e0;
            if (gi == 0) {
                const TouchedRec rc = w.rec[hb.rec];
                e0 = SegEntry{rc.src, rc.w};
            } else {
                e0 = w.list[(int64_t)dest * kTCap + gi];
            }
            const bool have = (unsigned)gi < min(n, (unsigned)kTCap);
            fix_add_row<NV>(acc, ld_row4<NV>(w.rows, have ? e0.src : 0, dim, dim, sub), have ? e0.w : 0.f);
        }
        // the chunk partials pass B' left, FOUR chunks of the group per step: their hash slots are probed side by side (first probes together; a
        // collision falls back to the sequential search) and their partial rows requested together -- two dependent round trips per four chunks
        // instead of eight (a Zipf(1.0) batch's hottest row has ~144 chunks, nine per group: the launch was that row's 18 round trips)
        for (unsigned j0 = (unsigned)gi; j0 * kTChunk < over; j0 += 64u) {
            unsigned hs[4];
            bool on[4];
            unsigned long long k0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned j = j0 + 16u * u;
                on[u] = j * kTChunk < over;
                hs[u] = touched_hash(touched_key(dest, on[u] ? j : 0u)) & w.hmask;
                k0[u] = keys[hs[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (on[u] && k0[u] != touched_key(dest, j0 + 16u * u)) hs[u] = touched_find(keys, w.hmask, dest, j0 + 16u * u);
            longlong2 pa[4][NV], pb[4][NV];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long* pr = w.hpart + (int64_t)hs[u] * dim;
#pragma unroll
                for (int jv = 0; jv < NV; ++jv) {
                    const int c = (sub + 16 * jv) * 4;
                    const int cc = c < dim ? c : 0;
                    pa[u][jv] = reinterpret_cast<const longlong2*>(pr + cc)[0];
                    pb[u][jv] = reinterpret_cast<const longlong2*>(pr + cc)[1];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int jv = 0; jv < NV; ++jv) {
                    const int c = (sub + 16 * jv) * 4;
                    if (on[u] && c < dim) {
                        acc[4 * jv + 0] += pa[u][jv].x; acc[4 * jv + 1] += pa[u][jv].y; acc[4 * jv + 2] += pb[u][jv].x; acc[4 * jv + 3] += pb[u][jv].y;
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
    TouchedWork w = touched_carve(work, n_ent, n_rel, batch, n_neg, dim);
    // the relation role scans the whole batch once per relation: while that product is small (configs[4]: 1,000 x 1,024)
    w.rel_role = (n_rel * std::max<int64_t>(batch, 1) <= (1LL << 24)) ? 1 : 0;
    w.combine = (SNAG_T_COMBINE && (w.parts == 2 || w.parts == 4)) ? 1 : 0;
    if (!w.rel_role && hipMemsetAsync(d_rel, 0, sizeof(float) * (size_t)n_rel * dim, st) != hipSuccess) return fail(SNAG_ERR_LAUNCH, "%s: memset failed", who);
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
                             (const IDX*)h, (const IDX*)r, (const IDX*)t, (const IDX*)neg_t, batch, n_neg, margin, grad_scale, pos, neg, w, d_ent, d_ent_ld)
#define SNAG_TA(NV) do { if (idx_bytes == 4) { if (small_agg) SNAG_TA_(int32_t, NV, 256); else SNAG_TA_(int32_t, NV, 1024); } \
                         else { if (small_agg) SNAG_TA_(int64_t, NV, 256); else SNAG_TA_(int64_t, NV, 1024); } } while (0)
        SNAG_NV_SWITCH(SNAG_TA)
#undef SNAG_TA
#undef SNAG_TA_
        int rc = check_launch(who);
        if (rc) return rc;
    }
    // pass B: one lane per record, + 1 workgroup that sums the losses (an empty batch: loss = 0), + 8 that wipe the previous launch's hash keys
    const int group_blocks = ceil_div(w.n_rec, 256);          // one lane per record
    const int chunk_blocks = batch > 0 ? 512 : 0;                   // pass B': a fixed grid of 2,048 waves walks the list of claimed chunks
    const int rel_blocks = w.rel_role ? ceil_div(n_rel, 16) : 0;   // one 16-lane group per relation
    const dim3 gridb((unsigned)(rel_blocks + chunk_blocks + group_blocks) + 9u);
#define SNAG_TB(NV) hipLaunchKernelGGL((touched_b_kernel<NV, TabT>), gridb, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, d_ent, d_ent_ld, d_rel, w, loss_part, batch, loss, \
                                       group_blocks, chunk_blocks, rel_blocks, (int)n_rel, r, idx_bytes)
    if (!(SNAG_T_ABLATE & 2)) { SNAG_NV_SWITCH(SNAG_TB) }
#undef SNAG_TB
    int rc = check_launch(who);
    if (rc || batch == 0) return rc;
    const dim3 gridh((unsigned)std::min<int64_t>(SNAG_HUB_GRID, (int64_t)w.max_hubs));
#define SNAG_TH(NV) hipLaunchKernelGGL((touched_hub_kernel<NV, TabT>), gridh, block, 0, st, ent, ent_ld, rel, dim, (int)n_ent, d_ent, d_ent_ld, d_rel, w)
    if (!(SNAG_T_ABLATE & 1)) { SNAG_NV_SWITCH(SNAG_TH) }
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
