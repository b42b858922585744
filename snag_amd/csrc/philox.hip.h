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
__device__ __forceinline__ bool key_known(const int64_t* __restrict__ keys, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t v = keys[mid];
        if (v == key) return true;
        if (v < key) lo = mid + 1; else hi = mid;
    }
    return false;
}

}  // namespace snag
