// Does global_load_lds_dwordx4 accept a source address that is only 4-byte aligned (conv taps shifted by one pixel), and at
// what rate?  Each wavefront DMAs 1-KiB pieces (64 lanes x 16 B) of a long row into LDS, reads them back and checksums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma_piece(unsigned lds_dst, const void* sbase, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

__global__ __launch_bounds__(256) void k(const float* src, float* out, int shift, int pieces) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + wave * 4096;
    float acc = 0.f;
    const float* base = src + ((long)blockIdx.x * 4 + wave) * (long)pieces * 256 + shift;
    for (int p = 0; p < pieces; p += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_piece(lds0 + q * 1024, base + (long)(p + q) * 256, lane * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(lds + wave * 1024 + q * 256 + lane * 4);
            acc += v.x + 2 * v.y + 3 * v.z + 4 * v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int blocks = 1024, pieces = 64;
    const long n = (long)blocks * 4 * pieces * 256 + 16;
    std::vector<float> h(n);
    for (long i = 0; i < n; ++i) h[i] = (float)((i * 7) % 1001) * 0.001f;
    float *d, *o;
    (void)hipMalloc(&d, n * 4);
    hipMalloc(&o, blocks * 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> r(blocks * 256);
    for (int shift = 0; shift < 4; ++shift) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 16384, 0, d, o, shift, pieces);
        hipEventRecord(a);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 16384, 0, d, o, shift, pieces);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(r.data(), o, blocks * 256 * 4, hipMemcpyDeviceToHost);
        // check a few threads
        double worst = 0;
        for (int t = 0; t < blocks * 256; t += 997) {
            const int blk = t / 256, th = t % 256, wave = th / 64, lane = th % 64;
            double ref = 0;
            for (int p = 0; p < pieces; ++p) {
                const long e0 = ((long)blk * 4 + wave) * (long)pieces * 256 + shift + (long)p * 256 + lane * 4;
                ref += h[e0] + 2 * h[e0 + 1] + 3 * h[e0 + 2] + 4 * h[e0 + 3];
            }
            const double e = fabs(ref - r[t]) / (fabs(ref) + 1e-9);
            if (e > worst) worst = e;
        }
        printf("shift %d floats: %.1f us per launch, %.2f TB/s, worst rel err %.2e\n", shift, ms / 20 * 1e3,
               (double)blocks * 4 * pieces * 1024 / (ms / 20 * 1e-3) / 1e12, worst);
    }
    return 0;
}
