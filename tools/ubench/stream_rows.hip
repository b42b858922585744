// Microbenchmark: how fast can the first projection's operand -- a [15000, 4096] fp32 table, rows 16 KiB apart -- be pulled into the CUs, by
// access shape?  Every variant reads the whole table exactly once (cold: a 1 GiB READ sweep of another buffer in front of every launch, so no
// dirty lines are written back meanwhile) and XORs what it loads into one word per lane.
//   lin      : plain grid-stride dwordx4 streaming read (the chip's ceiling for this buffer)
//   frag     : the engine's FragLoader shape -- workgroup = 8 waves x 32 rows, lane (li, lh) loads 16 B at k-tile*64 + lh*16 and + 32 of row li,
//              S register sets in flight, optional s_barrier per k-tile, K split over `splits` workgroups per row tile
//   quad     : 4 adjacent lanes cover a row's 64 bytes of the k-tile (16 rows per wave-instruction, 2 instructions per 32 rows)
//   line     : 32-deep k-tiles -- lane (li, lh) loads the 64 contiguous bytes lh*64.. of row li's 128-byte line (4 dwordx4)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/stream_rows.hip -o tools/ubench/stream_rows && tools/ubench/stream_rows
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void lin_kernel(const uint4* __restrict__ p, int64_t n16, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// MODE 0 frag, 1 quad, 2 line.  S sets in flight; BAR: s_barrier per k-tile; grid = row tiles x splits
template <int MODE, int S, bool BAR>
__global__ __launch_bounds__(512) void rows_kernel(const char* __restrict__ base, int nrows, int64_t pitch, int ktiles_per_split, int splits, uint32_t* __restrict__ out) {
    const int tile = blockIdx.x / splits, z = blockIdx.x % splits;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = tile * 256 + wave * 32;
    constexpr int TB = MODE == 2 ? 128 : 64;                       // bytes per row and k-tile
    constexpr int NL = MODE == 2 ? 4 : 2;                          // loads per lane and k-tile
    uint32_t voff[NL];
    if (MODE == 0) {
        const int li = lane & 31, lh = lane >> 5;
        const uint32_t r = (uint32_t)(min(row0 + li, nrows - 1) - min(row0, nrows - 1));
        voff[0] = r * (uint32_t)pitch + lh * 16;
        voff[1] = voff[0] + 32;
    } else if (MODE == 1) {
        const int r16 = lane >> 2, c = lane & 3;
        for (int i = 0; i < 2; ++i) {
            const uint32_t r = (uint32_t)(min(row0 + r16 + 16 * i, nrows - 1) - min(row0, nrows - 1));
            voff[i] = r * (uint32_t)pitch + c * 16;
        }
    } else {
        const int li = lane & 31, lh = lane >> 5;
        const uint32_t r = (uint32_t)(min(row0 + li, nrows - 1) - min(row0, nrows - 1));
        for (int i = 0; i < NL; ++i) voff[i % NL] = r * (uint32_t)pitch + lh * 64 + i * 16;
    }
    const char* ub = base + (int64_t)min(row0, nrows - 1) * pitch + (int64_t)z * ktiles_per_split * TB;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ub), 0, (int)0xffffffffu, 0x00020000);
    using u4 = __attribute__((ext_vector_type(4))) uint32_t;
    u4 r[S][NL];
    uint32_t acc = 0;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < NL; ++i) r[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], (uint32_t)s * TB, 0);
    const int nk = ktiles_per_split;
    for (int kt = 0; kt < nk; kt += S) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc ^= r[s][i][0] ^ r[s][i][1] ^ r[s][i][2] ^ r[s][i][3];
            const int nxt = kt + s + S;
            const uint32_t so = (uint32_t)(nxt < nk ? nxt : nk - 1) * TB;            // (past the end: re-read the last tile -- L2 hits)
#pragma unroll
            for (int i = 0; i < NL; ++i) r[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], so, 0);
            if (BAR) __builtin_amdgcn_s_barrier();
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static float med(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int N = 15000, F = 4096;
    const int64_t pitch = (int64_t)F * 4, bytes = (int64_t)N * pitch;
    char *tab, *sweep;
    uint32_t* out;
    CK(hipMalloc(&tab, bytes + 4096));
    CK(hipMalloc(&sweep, 1LL << 30));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(tab, 1, bytes));
    CK(hipMemset(sweep, 2, 1LL << 30));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto flush = [&]() { hipLaunchKernelGGL(lin_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)sweep, (1LL << 30) / 16, out); };
    auto timeit = [&](const char* name, auto launch, double mb) {
        std::vector<float> t;
        for (int rep = 0; rep < 9; ++rep) {
            flush();
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) t.push_back(ms * 1e3f);
        }
        const float us = med(t);
        printf("%-44s %7.1f us  %6.2f TB/s\n", name, us, mb / us);
        fflush(stdout);
    };
    const double mb = bytes / 1e6;
    for (int blocks : {1024, 2048, 4096, 8192})
        timeit((std::string("lin grid ") + std::to_string(blocks)).c_str(), [&]() { hipLaunchKernelGGL(lin_kernel, dim3(blocks), dim3(256), 0, 0, (const uint4*)tab, bytes / 16, out); }, mb);
    const int tiles = (N + 255) / 256;
#define ROWS(MODE, S, BAR, SPLITS)                                                                                                   \
    do {                                                                                                                             \
        const int TB = MODE == 2 ? 128 : 64, nkt = F * 4 / TB, kps = (nkt + SPLITS - 1) / SPLITS;                                     \
        char nm[96];                                                                                                                 \
        snprintf(nm, sizeof nm, "%s S=%d bar=%d splits=%d (%d wg x %d kt)", MODE == 0 ? "frag" : MODE == 1 ? "quad" : "line", S, (int)BAR, SPLITS, tiles * SPLITS, kps); \
        timeit(nm, [&]() { hipLaunchKernelGGL((rows_kernel<MODE, S, BAR>), dim3(tiles * SPLITS), dim3(512), 0, 0, (const char*)tab, N, pitch, kps, SPLITS, out); }, mb); \
    } while (0)
    ROWS(0, 4, true, 3); ROWS(0, 4, false, 3); ROWS(0, 8, false, 3); ROWS(0, 4, true, 4); ROWS(0, 4, false, 4); ROWS(0, 8, false, 4);
    ROWS(0, 4, false, 8); ROWS(0, 4, false, 16); ROWS(0, 8, false, 8);
    ROWS(1, 4, true, 3); ROWS(1, 4, false, 3); ROWS(1, 8, false, 4); ROWS(1, 4, false, 8); ROWS(1, 8, false, 8);
    ROWS(2, 4, true, 3); ROWS(2, 4, false, 3); ROWS(2, 4, false, 4); ROWS(2, 2, false, 4); ROWS(2, 4, false, 8); ROWS(2, 2, false, 8);
    return 0;
}
