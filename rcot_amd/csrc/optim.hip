// Fused flat-buffer optimizers (reference: torch.optim.RMSprop / Adam defaults, trainer.py:121-126).
// Parameters, gradients and state of one network live in contiguous fp32 buffers, so one launch
// updates the whole network at HBM speed (5 streams of 4 B per element for RMSprop).
#include "common.h"
#include "../../include/rcot_hip.h"

namespace {

__global__ __launch_bounds__(256) void rmsprop_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                      float4* __restrict__ sq, long n4, float lr, float alpha, float oma,
                                                      float eps, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pv = p[i], gv = g[i], sv = sq[i];
#define RCOT_RMS(c)                                        \
    {                                                      \
        const float gg = gv.c * gscale;                    \
        sv.c = alpha * sv.c + oma * gg * gg;     \
        pv.c -= lr * gg / (sqrtf(sv.c) + eps);             \
    }
        RCOT_RMS(x) RCOT_RMS(y) RCOT_RMS(z) RCOT_RMS(w)
#undef RCOT_RMS
        p[i] = pv;
        sq[i] = sv;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                   float4* __restrict__ m, float4* __restrict__ v, long n4, float lr,
                                                   float b1, float b2, float omb1, float omb2, float eps, float step_size, float rsqrt_bc2,
                                                   float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pv = p[i], gv = g[i], mv = m[i], vv = v[i];
#define RCOT_ADAM(c)                                                     \
    {                                                                    \
        const float gg = gv.c * gscale;                                  \
        mv.c = b1 * mv.c + omb1 * gg;                              \
        vv.c = b2 * vv.c + omb2 * gg * gg;                         \
        pv.c -= step_size * mv.c / (sqrtf(vv.c) * rsqrt_bc2 + eps);     \
    }
        RCOT_ADAM(x) RCOT_ADAM(y) RCOT_ADAM(z) RCOT_ADAM(w)
#undef RCOT_ADAM
        p[i] = pv;
        m[i] = mv;
        v[i] = vv;
    }
}

inline bool ok16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

// Hyper-parameters arrive as doubles so that 1-alpha / 1-beta / bias corrections are formed in double
// precision exactly as torch.optim does on the host (1 - 0.999f would be off by 1.3e-5 relative).
int rcot_rmsprop_step(float* p, const float* g, float* sq, long n, double lr, double alpha, double eps,
                      double grad_scale, void* stream) {
    if (!p || !g || !sq || n <= 0 || (n & 3) || !ok16(p) || !ok16(g) || !ok16(sq)) return RCOT_EINVAL;
    const long n4 = n >> 2;
    long grid = (n4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    RCOT_LAUNCH(rmsprop_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)sq, n4, (float)lr, (float)alpha, (float)(1.0 - alpha), (float)eps, (float)grad_scale);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                   int step, double grad_scale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || (n & 3) || step <= 0 || !ok16(p) || !ok16(g) || !ok16(m) || !ok16(v))
        return RCOT_EINVAL;
    const long n4 = n >> 2;
    long grid = (n4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    RCOT_LAUNCH(adam_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, n4, (float)lr, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2),
                       (float)eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)grad_scale);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // extern "C"
