// Ceiling of a bf16x3 split product on registers: per 32x32x16 fp32-equivalent block THREE v_mfma_f32_32x32x16_bf16
// (hi*hi + hi*lo + lo*hi) against EIGHT v_mfma_f32_32x32x2_f32.  Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(threadIdx.x * 1e-3f + q); b[q] = (__bf16)(1.f - q * 0.1f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int wps = 1; wps <= 3; ++wps) {
        const int iters = 4096, grid = 256 * wps;
        k_bf16<4><<<grid, 256>>>(d, 64);
        (void)hipEventRecord(s);
        k_bf16<4><<<grid, 256>>>(d, iters);
        (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e);
        const double n_mfma = (double)grid * 4 * iters * 8 * 4;                 // wave-level MFMA instructions
        const double raw = n_mfma * 32.0 * 32 * 16 * 2 / ms / 1e9;                 // bf16 TFLOP/s
        printf("waves/SIMD=%d: %.0f TFLOP/s raw bf16 (32x32x16) -> %.0f TFLOP/s fp32-equivalent with the 3-product split\n", wps, raw, raw / 3);
    }
    return 0;
}
