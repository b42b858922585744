// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 from registers, NACC independent accumulators per wave,
// W waves per block (one block per CU x OCC blocks).  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int threads) {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 4000;
    dim3 grid(256 * blocks_per_cu), block(threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, 10, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid.x * (threads / 64);
    const double flop = waves * iters * 8.0 * NACC * 4096.0;
    printf("NACC=%d blocks/CU=%d threads=%d : %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, threads, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256);
    run<2>(2, 256); run<2>(4, 256); run<1>(4, 256); run<4>(2, 256);
    return 0;
}
