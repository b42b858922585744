// a7, training form: the hinge scorer's forward AND backward in ONE launch (reference: model/MSNEA.py:110-127 + its autograd).
//
// The two-launch form gathers the same K+3 table rows twice (snag_score_margin_fwd, then snag_score_margin_bwd, which in hinge
// mode recomputes every score anyway) and needs a third launch to sum the per-triple losses.  Here each row is gathered ONCE; the
// launch writes pos / neg / loss_part, scatter-adds the gradients, and the block that finishes last sums the per-triple losses
// in a fixed order.
//
// Layout: a row spread over the whole wave (col = lane + 64*i), so every fp32-atomic wave-instruction adds 256 contiguous bytes
// of one gradient row: the full-rate shape of the memory-side atomic units, which bound this kernel (K+3 rows of dim floats
// added per triple at ~1.3 TB/s chip-wide).
//
// Parallelism: WPT waves share one triple, each taking a contiguous 1/WPT of its negatives.  A wave's life is "gather a chunk of
// rows in one round trip, then compute and issue that chunk's atomics"; hipcc drains the vector-memory counter to zero between
// chunks (loads and atomics share it, in order, and the atomics sit behind wave-uniform-but-not-provably-so branches), so the
// overlap of one wave's gather latency with another's atomics comes from having many short waves per CU (16 at B = 1024, K = 32
// with WPT = 4) rather than from software pipelining inside a wave (measured: 41 us for the pipelined one-wave-per-triple form
// with 4 waves per CU, every wait a vmcnt(0)).  The partial dq = d loss / d (h^ + r^) of the WPT waves and their hinge sums meet
// in LDS and are added in wave order by the triple's first wave, which owns the head, relation and positive-tail gradients.
#include "common.hip.h"
#include "score_rows.hip.h"

namespace snag {

constexpr int kTicketShards = 32;
static_assert(32 * (kTicketShards + 1) == SNAG_SCORE_TICKET_WORDS, "ticket layout: one 128-byte line per shard word + the top word");

template <typename IdxT, int NS, int WPT, typename ElemT>
__global__ __launch_bounds__(256) void score_fused_kernel(
    const ElemT* __restrict__ ent, int64_t eld, const ElemT* __restrict__ rel, int dim,
    const IdxT* __restrict__ h, const IdxT* __restrict__ r, const IdxT* __restrict__ t,
    const IdxT* __restrict__ neg_t, int64_t batch, int n_neg_all, float margin, float gscale,
    float* __restrict__ pos, float* __restrict__ neg, float* __restrict__ loss_part, float* __restrict__ loss,
    unsigned* __restrict__ ticket, float* __restrict__ d_ent, int64_t dld, float* __restrict__ d_rel) {
    constexpr int CH = (NS <= 2) ? 8 : 4;                        // rows gathered per round trip (CH*NS registers)
    __shared__ float dq_part[(WPT > 1) ? 4 * NS * 64 : 1];
    __shared__ float hinge_part[4], cpos_part[4];
    __shared__ unsigned is_last;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int part = wave % WPT;
    const int64_t b = (int64_t)blockIdx.x * (4 / WPT) + wave / WPT;
    const bool live = b < batch;
    float hinge = 0.f, c_pos = 0.f;
    RowS<NS> hh, rr, tt, q, dq;
    float den_h = 1.f, den_r = 1.f, den_t = 1.f, s_pos = 0.f;
    int64_t hi = 0, ri = 0, ti = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) dq.v[i] = 0.f;
    if (live) {
        hi = (int64_t)h[b]; ri = (int64_t)r[b]; ti = (int64_t)t[b];
        // this wave's share of the negatives: [k_lo, k_lo + n_neg)
        const int per_part = (n_neg_all + WPT - 1) / WPT;
        const int k_lo = min(n_neg_all, part * per_part);
        const int n_neg = min(n_neg_all - k_lo, per_part);
        const IdxT* nrow = neg_t + b * (int64_t)n_neg_all + k_lo;
        float* out = neg + b * (int64_t)n_neg_all + k_lo;
        int my = (lane < n_neg) ? (int)nrow[lane] : 0;
        hh = load_row_s<NS>(ent, hi, eld, dim, lane);
        rr = load_row_s<NS>(rel, ri, dim, dim, lane);
        tt = load_row_s<NS>(ent, ti, eld, dim, lane);
        RowS<NS> x[CH];
        auto fetch = [&](int c, int cnt) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (c + u < cnt) x[u] = load_row_s<NS>(ent, (int64_t)__builtin_amdgcn_readlane(my, c + u), eld, dim, lane);
        };
        fetch(0, min(64, n_neg));                 // first chunk rides the same round trip as head / relation / tail
        den_h = normalize_s<NS>(hh);
        den_r = normalize_s<NS>(rr);
        den_t = normalize_s<NS>(tt);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            q.v[i] = hh.v[i] + rr.v[i];
            const float d = q.v[i] - tt.v[i];
            s_pos = fmaf(d, d, s_pos);
        }
        s_pos = sum64(s_pos);
        if (lane == 0 && part == 0) pos[b] = s_pos;
        for (int k0 = 0; k0 < n_neg; k0 += 64) {
            const int cnt = min(64, n_neg - k0);
            if (k0 > 0) my = (k0 + lane < n_neg) ? (int)nrow[k0 + lane] : 0;
            float sc = 0.f;                       // lane k keeps the score of negative k0 + k: one coalesced store per id block
            for (int c = 0; c < cnt; c += CH) {
                if (c > 0 || k0 > 0) fetch(c, cnt);
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (c + u < cnt) {            // wave-uniform
                        const float den_x = normalize_s<NS>(x[u]);
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const float d = q.v[i] - x[u].v[i];
                            s = fmaf(d, d, s);
                        }
                        s = sum64(s);
                        if (lane == c + u) sc = s;
                        const float viol = margin + s_pos - s;
                        hinge += fmaxf(viol, 0.f);
                        if (viol > 0.f) {         // wave-uniform: d loss / d s_neg = -gscale, d loss / d s_pos += gscale
                            c_pos += gscale;
                            RowS<NS> dx;
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                const float e2 = -2.f * gscale * (q.v[i] - x[u].v[i]);
                                dq.v[i] += e2;
                                dx.v[i] = -e2;
                            }
                            normalize_bwd_atomic<NS, float>(x[u], den_x, dx, d_ent + (int64_t)__builtin_amdgcn_readlane(my, c + u) * dld, dim, lane);
                        }
                    }
                }
            }
            if (lane < cnt) out[k0 + lane] = sc;
        }
    }
    // ---- the triple's waves meet: partial dq, hinge and positive coefficient are added in wave order by its first wave -----------
    if (WPT > 1) {
#pragma unroll
        for (int i = 0; i < NS; ++i) dq_part[(wave * NS + i) * 64 + lane] = dq.v[i];
        if (lane == 0) { hinge_part[wave] = hinge; cpos_part[wave] = c_pos; }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int p = 1; p < WPT; ++p) {
#pragma unroll
                for (int i = 0; i < NS; ++i) dq.v[i] += dq_part[((wave + p) * NS + i) * 64 + lane];
                hinge += hinge_part[wave + p];
                c_pos += cpos_part[wave + p];
            }
        }
    }
    if (live && part == 0 && c_pos != 0.f) {      // some hinge of this triple is active
        RowS<NS> dx;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float e2 = 2.f * c_pos * (q.v[i] - tt.v[i]);
            dq.v[i] += e2;
            dx.v[i] = -e2;
        }
        normalize_bwd_atomic<NS, float>(tt, den_t, dx, d_ent + ti * dld, dim, lane);
        normalize_bwd_atomic<NS, float>(hh, den_h, dq, d_ent + hi * dld, dim, lane);
        normalize_bwd_atomic<NS, float>(rr, den_r, dq, d_rel + ri * (int64_t)dim, dim, lane);
    }
    const bool owner = live && part == 0 && lane == 0;
    if (loss == nullptr) {
        if (owner && loss_part != nullptr) loss_part[b] = hinge;
        return;
    }
    // ---- sum of the per-triple losses by the block that finishes last (fixed order: bit-reproducible) --------------------------
    // Hand-off without fences (MI355X_MICROARCH.md, "Valid forms", first row of the measured table): every byte handed off is
    // stored write-through (agent-scope relaxed store = global_store sc1) and drained (vmcnt(0)) before the workgroup barrier; ONE
    // lane then takes a ticket with a returning agent-scope atomic; the block whose ticket is the last one reads every loss_part
    // with agent-scope relaxed loads (global_load sc1: they bypass the L1, and this kernel never read those lines before).
    // One ticket word serialises its arrivals at ~12 ns each (1,024 blocks: +9 us measured), so the tickets are a two-level tree:
    // block -> shard word (blockIdx % kTicketShards, each on its own 128-B line), last arrival of a shard -> top word.
    if (owner) __hip_atomic_store(loss_part + b, hinge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned shard = blockIdx.x % kTicketShards;
        const unsigned in_shard = (gridDim.x - shard + kTicketShards - 1) / kTicketShards;       // blocks with this remainder
        const unsigned shards = min(gridDim.x, (unsigned)kTicketShards);
        unsigned last = 0u;
        unsigned* sw = ticket + 32u * shard;
        if (__hip_atomic_fetch_add(sw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1u) {
            __hip_atomic_store(sw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // ready for the next launch
            unsigned* top = ticket + 32u * kTicketShards;
            if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u) {
                __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1u;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (is_last == 0u) return;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < batch; i += 256) s += __hip_atomic_load(loss_part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = sum64(s);
    if (lane == 0) hinge_part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((hinge_part[0] + hinge_part[1]) + hinge_part[2]) + hinge_part[3];
}

template <typename IdxT, typename ElemT>
static int launch_fused(const ElemT* ent, int64_t eld, const ElemT* rel, int dim, const void* h, const void* r, const void* t,
                        const void* neg_t, int64_t batch, int n_neg, float margin, float gscale, float* pos, float* neg,
                        float* loss_part, float* loss, unsigned* ticket, float* d_ent, int64_t dld, float* d_rel, hipStream_t st) {
    // waves per triple: enough short waves to keep ~16 per CU (256 CUs) while a triple still has >= 8 negatives per wave
    int wpt = 1;
    if (batch * 4 <= 4096 && n_neg >= 32) wpt = 4;
    else if (batch * 2 <= 4096 && n_neg >= 16) wpt = 2;
    const dim3 grid((unsigned)ceil_div(batch, 4 / wpt)), block(256);
    const int ns = (dim + 63) / 64;
#define SNAG_FUSED_W(NS, W)                                                                                                      \
    hipLaunchKernelGGL((score_fused_kernel<IdxT, NS, W, ElemT>), grid, block, 0, st, ent, eld, rel, dim, (const IdxT*)h, (const IdxT*)r, \
                       (const IdxT*)t, (const IdxT*)neg_t, batch, n_neg, margin, gscale, pos, neg, loss_part, loss, ticket,     \
                       d_ent, dld, d_rel)
#define SNAG_FUSED(NS)                                \
    do {                                              \
        if (wpt == 4) SNAG_FUSED_W(NS, 4);            \
        else if (wpt == 2) SNAG_FUSED_W(NS, 2);       \
        else SNAG_FUSED_W(NS, 1);                     \
    } while (0)
    switch (ns) {
        case 1: SNAG_FUSED(1); break;
        case 2: SNAG_FUSED(2); break;
        case 3: SNAG_FUSED(3); break;
        case 4: SNAG_FUSED(4); break;
        case 5: case 6: SNAG_FUSED(6); break;
        default: SNAG_FUSED(8); break;
    }
#undef SNAG_FUSED
#undef SNAG_FUSED_W
    return check_launch("snag_score_margin_fwd_bwd");
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_score_margin_fwd_bwd(const float* ent, int64_t ent_ld, int64_t n_ent, const float* rel, int64_t n_rel, int dim,
                                                  const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                  int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                  float* loss_part, float* loss, uint32_t* ticket, float* d_ent, int64_t d_ent_ld,
                                                  float* d_rel, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SNAG_REQUIRE(batch == 0 || (loss == nullptr) || (loss_part != nullptr && ticket != nullptr), SNAG_ERR_ARG,
                 "score_fwd_bwd: the summed loss needs loss_part and a ticket word");
    if (batch == 0) {           // nothing to score: the summed loss is zero, the gradients are untouched
        if (loss != nullptr) return hipMemsetAsync(loss, 0, sizeof(float), st) == hipSuccess ? SNAG_OK : fail(SNAG_ERR_LAUNCH, "score_fwd_bwd: memset failed");
        return SNAG_OK;
    }
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(pos && (n_neg == 0 || (neg && neg_t)), SNAG_ERR_ARG, "score_fwd_bwd: null score output or negative-tail ids");
    SNAG_REQUIRE(d_ent && d_rel, SNAG_ERR_ARG, "score_fwd_bwd: null gradient pointer");
    SNAG_REQUIRE(ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_ARG, "score_fwd_bwd: row stride smaller than dim=%d", dim);
    return idx_bytes == 4 ? launch_fused<int32_t, float>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, pos, neg, loss_part,
                                                         loss, ticket, d_ent, d_ent_ld, d_rel, st)
                          : launch_fused<int64_t, float>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, pos, neg, loss_part,
                                                         loss, ticket, d_ent, d_ent_ld, d_rel, st);
}

extern "C" SNAG_API int snag_score_margin_fwd_bwd_bf16(const uint16_t* ent, int64_t ent_ld, int64_t n_ent, const uint16_t* rel, int64_t n_rel, int dim,
                                                       const void* h, const void* r, const void* t, const void* neg_t, int idx_bytes,
                                                       int64_t batch, int n_neg, float margin, float grad_scale, float* pos, float* neg,
                                                       float* loss_part, float* loss, uint32_t* ticket, float* d_ent, int64_t d_ent_ld,
                                                       float* d_rel, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SNAG_REQUIRE(batch == 0 || (loss == nullptr) || (loss_part != nullptr && ticket != nullptr), SNAG_ERR_ARG,
                 "score_fwd_bwd_bf16: the summed loss needs loss_part and a ticket buffer");
    if (batch == 0) {           // nothing to score: the summed loss is zero, the gradients are untouched
        if (loss != nullptr) return hipMemsetAsync(loss, 0, sizeof(float), st) == hipSuccess ? SNAG_OK : fail(SNAG_ERR_LAUNCH, "score_fwd_bwd_bf16: memset failed");
        return SNAG_OK;
    }
    int rc = check_score_args(ent, n_ent, rel, n_rel, dim, h, r, t, neg_t, idx_bytes, batch, n_neg);
    if (rc) return rc;
    SNAG_REQUIRE(pos && (n_neg == 0 || (neg && neg_t)), SNAG_ERR_ARG, "score_fwd_bwd_bf16: null score output or negative-tail ids");
    SNAG_REQUIRE(d_ent && d_rel, SNAG_ERR_ARG, "score_fwd_bwd_bf16: null gradient pointer");
    SNAG_REQUIRE(ent_ld >= dim && d_ent_ld >= dim, SNAG_ERR_ARG, "score_fwd_bwd_bf16: row stride smaller than dim=%d", dim);
    return idx_bytes == 4 ? launch_fused<int32_t, uint16_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, pos, neg, loss_part,
                                                            loss, ticket, d_ent, d_ent_ld, d_rel, st)
                          : launch_fused<int64_t, uint16_t>(ent, ent_ld, rel, dim, h, r, t, neg_t, batch, n_neg, margin, grad_scale, pos, neg, loss_part,
                                                            loss, ticket, d_ent, d_ent_ld, d_rel, st);
}
