// (f3) on-device negative sampling, "distinct" form: the in-tree sampler's two properties the with-replacement kernel of misc.hip
// does not have (generate_neg_triples_fast, model/MSNEA_tools.py:59-90): candidates come from the KG's OWN entity list (:64-65),
// and the negatives of one positive are distinct (random.sample draws without replacement, :69,:72).
//
// One wave per positive; lane = slot inside a block of 64 slots.  Rounds = the reference's tries (max_try = 10, :59): in round a
// every still-pending slot k draws  cand = ent_list[(philox(seed, (b*K + k)*16 + a).x * n_list) >> 32];  it is accepted when it is
// not a known triple, not already accepted for this positive, and no lower pending slot accepted the same value in this round; the
// last round accepts whatever it drew (:76-78 keep the last try unfiltered).  oracle/sampler.py restates this rule and the Philox
// stream bit for bit (the reference's python `random` stream itself cannot be matched).
#include "common.hip.h"
#include "philox.hip.h"

namespace snag {

constexpr int kMaxDistinct = 1024;       // slots per positive (LDS: 4 KB per wave)

__global__ __launch_bounds__(256) void sample_distinct_kernel(uint64_t seed, const uint64_t* __restrict__ tick, const int32_t* __restrict__ h,
                                                              const int32_t* __restrict__ r, const int32_t* __restrict__ t, int64_t B, int K,
                                                              int n_ent, int n_rel, int mode, const int64_t* __restrict__ known, int64_t n_known,
                                                              const int32_t* __restrict__ ent_list, int n_list,
                                                              int32_t* __restrict__ neg_h, int32_t* __restrict__ neg_t) {
    __shared__ int vals_all[4][kMaxDistinct];
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    if (tick != nullptr) seed += tick[0] * 0x9E3779B97F4A7C15ull;
    int* vals = vals_all[wave];
    const int hb = h[b], rb = r[b], tb = t[b];
    const bool corrupt_head = (mode == 1) && ((philox4x32_10(seed ^ 0xC0FFEEull, (uint64_t)b).x & 1u) != 0);
    for (int k = lane; k < K; k += 64) vals[k] = -1;
    for (int c0 = 0; c0 < K; c0 += 64) {
        const int k = c0 + lane;
        bool pending = k < K;
        const int seen = min(K, c0 + 64);              // slots that may already hold an accepted value
        for (int a = 0; a < 10; ++a) {
            if (__ballot(pending) == 0ull) break;
            const bool last = (a == 9);
            int cand = 0;
            bool ok = false;
            if (pending) {
                const uint32_t u = philox4x32_10(seed, (uint64_t)(b * K + k) * 16 + a).x;
                const int j = (int)(((uint64_t)u * (uint64_t)n_list) >> 32);
                cand = ent_list != nullptr ? ent_list[j] : j;
                ok = true;
                if (!last) {
                    if (known != nullptr) {
                        const int64_t key = corrupt_head ? ((int64_t)cand * n_rel + rb) * n_ent + tb : ((int64_t)hb * n_rel + rb) * n_ent + cand;
                        ok = !key_known(known, n_known, key);
                    }
                }
            }
            if (!last) {
                // already accepted for this positive (earlier blocks of slots, earlier rounds of this block)
                for (int i = 0; i < seen; ++i) ok = ok && (vals[i] != cand);
                // same value drawn twice in this round: the lowest pending slot keeps it
                for (int j = 0; j < 64; ++j) {
                    const int cj = __builtin_amdgcn_readlane(cand, j);
                    const int okj = __builtin_amdgcn_readlane((int)ok, j);
                    if (okj && j < lane && cj == cand) ok = false;
                }
            }
            if (pending && ok) {
                vals[k] = cand;
                pending = false;
            }
            // the LDS writes of this round must be visible to the next round's membership test (same wave: a wait is enough)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    }
    for (int k = lane; k < K; k += 64) {
        const int cand = vals[k];
        if (neg_h != nullptr) neg_h[b * K + k] = corrupt_head ? cand : hb;
        neg_t[b * K + k] = corrupt_head ? tb : cand;
    }
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_sample_negatives_distinct(uint64_t seed, const uint64_t* tick, const int32_t* h, const int32_t* r, const int32_t* t,
                                                       int64_t batch, int n_neg, int n_ent, int n_rel, int mode, const int64_t* known_sorted,
                                                       int64_t n_known, const int32_t* ent_list, int n_list, int32_t* neg_h, int32_t* neg_t,
                                                       void* stream) {
    SNAG_REQUIRE(batch >= 0 && n_neg >= 0 && n_ent >= 1 && n_rel >= 1, SNAG_ERR_ARG, "sample_negatives_distinct: bad sizes");
    SNAG_REQUIRE(mode == 0 || mode == 1, SNAG_ERR_ARG, "sample_negatives_distinct: mode=%d (0 = tails, 1 = head-or-tail)", mode);
    if (batch == 0 || n_neg == 0) return SNAG_OK;
    SNAG_REQUIRE(h && r && t && neg_t, SNAG_ERR_ARG, "sample_negatives_distinct: null pointer");
    SNAG_REQUIRE(mode == 0 || neg_h != nullptr, SNAG_ERR_ARG, "sample_negatives_distinct: head-or-tail mode needs neg_h");
    SNAG_REQUIRE((known_sorted != nullptr) == (n_known > 0), SNAG_ERR_ARG, "sample_negatives_distinct: known_sorted and n_known go together");
    SNAG_REQUIRE((int64_t)n_ent * n_rel < (1LL << 62) / n_ent, SNAG_ERR_UNSUPPORTED, "sample_negatives_distinct: triple key overflows int64");
    if (ent_list == nullptr) n_list = n_ent;
    SNAG_REQUIRE(n_list >= 1, SNAG_ERR_ARG, "sample_negatives_distinct: empty entity list");
    SNAG_REQUIRE(n_neg <= kMaxDistinct, SNAG_ERR_UNSUPPORTED, "sample_negatives_distinct: n_neg=%d > %d", n_neg, kMaxDistinct);
    SNAG_REQUIRE(n_neg <= n_list, SNAG_ERR_ARG, "sample_negatives_distinct: %d distinct negatives out of %d candidates (random.sample would raise too)", n_neg, n_list);
    hipLaunchKernelGGL(sample_distinct_kernel, dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, (hipStream_t)stream, seed, tick, h, r, t, batch,
                       n_neg, n_ent, n_rel, mode, known_sorted, n_known, ent_list, n_list, neg_h, neg_t);
    return check_launch("snag_sample_negatives_distinct");
}
