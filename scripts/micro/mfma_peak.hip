// Sustained fp32 MFMA ceiling (v_mfma_f32_32x32x2f32) on registers only: NACC independent accumulators per wave,
// W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f - a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[0] = s;
}
template <int NACC> void run(int blocks_per_cu) {
    float* d; hipMalloc(&d, 4);
    const int iters = 4096, grid = 256 * blocks_per_cu;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<NACC><<<grid, 256>>>(d, 64);
    hipEventRecord(s);
    k<NACC><<<grid, 256>>>(d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double fl = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d: %.1f TF/s (%.2f ms)\n", NACC, blocks_per_cu, fl / ms / 1e9, ms);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<1>(2); run<4>(2); run<4>(3); run<3>(3);
    return 0;
}
