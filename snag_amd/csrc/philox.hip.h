// Philox4x32-10 and the sorted-key lookup shared by misc.hip (dropout masks, noise, the with-replacement sampler) and sampler.hip.
// oracle/sampler.py restates both bit for bit.
#pragma once
#include "common.hip.h"

namespace snag {

// ---- Philox4x32-10 (counter-based: element i of stream `seed` is a pure function of (seed, i)) ----
struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t ctr) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5a17c0deu, c3 = 0x9e3779b9u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// binary search of one key in the ascending int64 key table of the known triples
// Membership in a sorted key list by 16-ARY search: fifteen probes per round trip, the range shrinks 16-fold per step, the last <= 16
// candidates are compared side by side -- 5 dependent round trips at 100k keys where a binary search makes 17 (the sampler's chain of
// dependent loads was the longest role of the step prologue).
__device__ __forceinline__ bool key_known(const int64_t* __restrict__ keys, int64_t n, int64_t key) {
    if (n <= 0) return false;
    int64_t lo = 0, hi = n;                 // the key, if present, sits in [lo, hi); never empty
    while (hi - lo > 16) {
        const int64_t step = (hi - lo) >> 4;
        int64_t v[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) v[j] = keys[lo + (j + 1) * step];
        int c = 0;                          // probes at or below the key: keys[lo + c step] <= key < keys[lo + (c + 1) step]
#pragma unroll
        for (int j = 0; j < 15; ++j) c += v[j] <= key ? 1 : 0;
        if (c < 15) hi = lo + (c + 1) * step;
        lo += c * step;
    }
    const int64_t last = hi - lo - 1;
    bool found = false;
    int64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = keys[lo + (j < last ? j : last)];
#pragma unroll
    for (int j = 0; j < 16; ++j) found = found || v[j] == key;
    return found;
}

}  // namespace snag
