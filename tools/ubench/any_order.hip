// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is not supported on GFX9xx for
// hipExtModuleLaunchKernel.)  A: 236 workgroups x 512 threads spinning ~T cycles (one per CU, like the eight-wave GEMM launches);
// B: 4,096 small workgroups spinning ~T/10.  Timed: A then B plainly; A then B with the flag on B; the same on two streams.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/any_order.hip -o tools/ubench/any_order && tools/ubench/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ __launch_bounds__(512) void spin_big(long long cycles, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0x12345678u) sink[0] = x;
}
__global__ __launch_bounds__(256) void spin_small(long long cycles, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0x12345678u) sink[1] = x;
}

int main() {
    unsigned* sink;
    hipMalloc(&sink, 16);
    hipStream_t s0, s1;
    hipStreamCreate(&s0);
    hipStreamCreate(&s1);
    hipEvent_t e0, e1, ef;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ef, hipEventDisableTiming);
    const long long TA = 10000, TB = 1000;       // clock64 ticks (100 MHz): 100 us and 10 us
    uint32_t* flag;
    hipMalloc(&flag, 8);
    hipMemset(flag, 0, 8);
    uint32_t seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0, s0);
            for (int it = 0; it < 10; ++it) {
                hipLaunchKernelGGL(spin_big, dim3(236), dim3(512), 0, s0, TA, sink);
                if (mode == 0) hipLaunchKernelGGL(spin_small, dim3(4096), dim3(256), 0, s0, TB, sink);
                if (mode == 1) hipExtLaunchKernelGGL(spin_small, dim3(4096), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, TB, sink);
                if (mode == 3) {                     // the same fork / join through stream memory operations instead of events
                    ++seq;
                    hipStreamWriteValue32(s0, flag, seq, 0);          // (in the real step: behind the previous step's last launch)
                    hipStreamWaitValue32(s1, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
                    hipLaunchKernelGGL(spin_small, dim3(4096), dim3(256), 0, s1, TB, sink);
                    hipStreamWriteValue32(s1, flag + 1, seq, 0);
                    hipStreamWaitValue32(s0, flag + 1, seq, hipStreamWaitValueGte, 0xffffffffu);
                }
                if (mode == 2) {
                    hipEventRecord(ef, s0);          // fork behind the previous iteration, join in front of the next one
                    hipLaunchKernelGGL(spin_small, dim3(4096), dim3(256), 0, s1, TB, sink);
                    hipEventRecord(ef, s1);
                    hipStreamWaitEvent(s0, ef, 0);
                }
            }
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d (%s): %.1f us per (big + small) pair\n", mode, mode == 0 ? "plain, one stream" : mode == 1 ? "hipExtAnyOrderLaunch on the small one" : mode == 2 ? "small one on a second stream (join by event)" : "small one on a second stream (fork and join by hipStreamWrite/WaitValue32)",
               best * 100.f);
    }
    return 0;
}
