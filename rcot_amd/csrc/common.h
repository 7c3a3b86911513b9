// Shared helpers for the rcot_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#define RCOT_OK 0
#define RCOT_EINVAL (-1)      // bad shape / alignment / null pointer
#define RCOT_EWORKSPACE (-2)  // caller-provided workspace too small
#define RCOT_EUNSUPPORTED (-3) // no kernel of this entry point takes the shape

#define RCOT_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

namespace rcot {
// Per-launch device time stamps (rcot_profile_begin / rcot_profile_end, api.hip): while the calling thread collects, every launch of the
// library goes through hipExtLaunchKernelGGL with a start and a stop event of its own — the dispatch packet's own begin / end times,
// what rocprofv3 --kernel-trace lists, with nothing added between the kernels.
bool prof_on();
void prof_slot(const void* fn, hipEvent_t* e0, hipEvent_t* e1);
template <typename F> inline const void* prof_fn(F f) { return reinterpret_cast<const void*>(f); }
}  // namespace rcot

#define RCOT_LAUNCH(kernel, grid, block, smem, stream, ...)                                              \
    do {                                                                                                  \
        if (rcot::prof_on()) {                                                                            \
            hipEvent_t e0__, e1__;                                                                        \
            rcot::prof_slot(rcot::prof_fn(kernel), &e0__, &e1__);                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, e0__, e1__, 0, __VA_ARGS__);         \
        } else {                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                           \
        }                                                                                                 \
    } while (0)

namespace rcot {

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64). `red` must hold NT/64 floats.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// Unsigned division by a runtime-invariant divisor (host precomputes magic/shift).
struct FastDiv {
    uint32_t d, magic, shift;
    __host__ void init(uint32_t div) {
        d = div ? div : 1;
        shift = 0;
        while ((1ull << shift) < d) ++shift;
        magic = (uint32_t)(((1ull << 32) * ((1ull << shift) - d)) / d + 1);
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        return (uint32_t)(((uint64_t)__umulhi(n, magic) + n) >> shift);
    }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
        q = div(n);
        r = n - q * d;
    }
};

__device__ __forceinline__ float gelu_erf(float a) {
    return 0.5f * a * (1.0f + erff(a * 0.70710678118654752440f));
}
// d/da [a * Phi(a)] = Phi(a) + a * phi(a)
__device__ __forceinline__ float gelu_erf_grad(float a) {
    const float cdf = 0.5f * (1.0f + erff(a * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * a * a);
    return cdf + a * pdf;
}

// gelu(a) and gelu'(a) from ONE exponential: Phi via erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1/(1 + p x)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7, i.e. ~2 ulp of Phi), and phi(a) = exp(-a^2/2)/sqrt(2 pi) is the same
// exponential.  Used by the backward gate, which is VALU-bound with erff + expf per pixel.
__device__ __forceinline__ void gelu_and_grad(float a, float& g, float& dg) {
    const float e = __expf(-0.5f * a * a);
    const float x = fabsf(a) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, x, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float erfv = copysignf(fmaf(-q * t, e, 1.0f), a);
    const float cdf = fmaf(0.5f, erfv, 0.5f);
    g = a * cdf;
    dg = fmaf(a * 0.39894228040143267794f, e, cdf);
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Name of the kernel symbol the calling thread's last dispatcher chose (printf-style; api.hip).  bench.py reads it back through
// rcot_last_kernel() to attribute the time of an entry point to the kernel family that ran (several families serve one entry point).
void note_kernel(const char* fmt, ...);
inline const char* tf(bool b) { return b ? "true" : "false"; }

}  // namespace rcot
