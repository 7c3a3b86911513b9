// Does hipExtAnyOrderLaunch let two INDEPENDENT kernels of one stream overlap on gfx950 / ROCm 7.2?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/anyorder.hip -o scripts/micro/anyorder && scripts/micro/anyorder
// Two spin kernels (64 workgroups, ~T us each) back to back: both ordinary launches, then the second one with the flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, int* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}

static float run(hipStream_t st, int flagged, int reps, unsigned long long ticks) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, ticks, (int*)nullptr);
        if (flagged) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, (int*)nullptr);
        else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, ticks, (int*)nullptr);
    }
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    const unsigned long long ticks = 3000;   // 100 MHz -> 30 us
    run(st, 0, 5, ticks);
    printf("pair of 30 us kernels, ordinary launches : %.1f us per pair\n", run(st, 0, 50, ticks));
    printf("second launch with hipExtAnyOrderLaunch  : %.1f us per pair\n", run(st, 1, 50, ticks));
    printf("ordinary again                           : %.1f us per pair\n", run(st, 0, 50, ticks));
    return 0;
}
