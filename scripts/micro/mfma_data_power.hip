// Does the MFMA rate depend on WHAT is multiplied?  (It does on power-limited parts: operand bits that toggle cost energy, the clock follows.)
// The register-only v_mfma_f32_32x32x2f32 loop of mfma_sustained.hip (4 accumulators, 2 waves per SIMD, ~0.6 s per mode) with
//   mode 0: the same two small constants per lane for every MFMA (what mfma_peak / mfma_sustained measure),
//   mode 1: eight random operand pairs per lane cycling (random mantissas and signs, |x| ~ 1: what a GEMM on activations feeds),
//   mode 2: zeros,
// and the bf16 loop v_mfma_f32_32x32x16_bf16 with constant / random operands.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_data_power.hip -o mfma_data_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void kf(float* out, const float* rnd, int iters, int mode) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        if (mode == 0) { a[u] = threadIdx.x * 1e-3f; b[u] = 1.f - a[u]; }
        else if (mode == 1) { a[u] = rnd[(threadIdx.x * 8 + u) * 2]; b[u] = rnd[(threadIdx.x * 8 + u) * 2 + 1]; }
        else { a[u] = 0.f; b[u] = 0.f; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256) void kb(float* out, const float* rnd, int iters, int mode) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
    for (int u = 0; u < 4; ++u)
        for (int e = 0; e < 8; ++e) {
            const float x = mode == 0 ? 0.5f : rnd[((threadIdx.x * 4 + u) * 8 + e) * 2], y = mode == 0 ? 0.25f : rnd[((threadIdx.x * 4 + u) * 8 + e) * 2 + 1];
            a[u][e] = (__bf16)x; b[u][e] = (__bf16)y;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float *d, *rnd;
    (void)hipMalloc(&d, 4);
    const int NR = 256 * 64 * 2;
    float* h = (float*)malloc(NR * 4);
    srand(7);
    for (int i = 0; i < NR; ++i) h[i] = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 1.3f;      // signs and mantissas random; sums stay bounded-ish (random walk)
    (void)hipMalloc(&rnd, NR * 4); (void)hipMemcpy(rnd, h, NR * 4, hipMemcpyHostToDevice);
    const int iters = 4096, grid = 512, per = 8;
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    const char* names[3] = {"constant operands", "random operands", "zeros"};
    for (int mode = 0; mode < 3; ++mode) {
        const double fl = (double)grid * 4 * iters * 8 * 4 * 4096.0 * per;
        kf<<<grid, 256>>>(d, rnd, 64, mode); (void)hipDeviceSynchronize();
        double best = 0, last = 0;
        for (int w = 0; w < 12; ++w) {
            (void)hipEventRecord(s);
            for (int i = 0; i < per; ++i) kf<<<grid, 256>>>(d, rnd, iters, mode);
            (void)hipEventRecord(e); (void)hipEventSynchronize(e);
            float ms; (void)hipEventElapsedTime(&ms, s, e);
            last = fl / ms / 1e9; if (last > best) best = last;
        }
        printf("fp32 32x32x2  %-18s: %.1f TF/s in the last 50-ms window (best %.1f)\n", names[mode], last, best);
    }
    for (int mode = 0; mode < 2; ++mode) {
        const double fl = (double)grid * 4 * iters * 8 * 4 * 32768.0 * per;
        kb<<<grid, 256>>>(d, rnd, 64, mode); (void)hipDeviceSynchronize();
        double best = 0, last = 0;
        for (int w = 0; w < 12; ++w) {
            (void)hipEventRecord(s);
            for (int i = 0; i < per; ++i) kb<<<grid, 256>>>(d, rnd, iters, mode);
            (void)hipEventRecord(e); (void)hipEventSynchronize(e);
            float ms; (void)hipEventElapsedTime(&ms, s, e);
            last = fl / ms / 1e9; if (last > best) best = last;
        }
        printf("bf16 32x32x16 %-18s: %.1f TF/s in the last window (best %.1f)\n", names[mode], last, best);
    }
    return 0;
}
