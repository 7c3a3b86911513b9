// Does the fp32 MFMA ceiling hold under SUSTAINED load?  mfma_peak.hip measures a few milliseconds (boost clocks); a training step
// keeps the part busy for seconds.  The register-only v_mfma_f32_32x32x2f32 loop (4 accumulators, 2 waves per SIMD) for ~3 s, TF/s
// printed per 100 ms window; then the same with a 2 s pause in front (cool start).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_sustained.hip -o mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f - a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4);
    const int iters = 4096, grid = 512, per = 8;                      // one launch ~ 3.6 ms at 150 TF/s
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const double fl = (double)grid * 4 * iters * 8 * 4 * 4096.0 * per;
    for (int phase = 0; phase < 2; ++phase) {
        if (phase) { printf("-- after a 2 s pause\n"); sleep(2); }
        k<<<grid, 256>>>(d, 64);
        hipDeviceSynchronize();
        for (int w = 0; w < 100; ++w) {
            hipEventRecord(s);
            for (int i = 0; i < per; ++i) k<<<grid, 256>>>(d, iters);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            if (w < 5 || w % 10 == 9) printf("window %3d (t ~ %.2f s): %.1f TF/s\n", w, (w + 1) * ms * 1e-3, fl / ms / 1e9);
        }
    }
    return 0;
}
