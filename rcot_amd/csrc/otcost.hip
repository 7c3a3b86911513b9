// Fourier residual-guided OT cost of the generator step (reference: trainer.py:320-343; math: SURVEY.md A.5).
//   res = degraded - T(x);  rmse = sqrt(mean res^2) over the (global) batch
//   de_id < 3 : f_i = mean|FFT2(res_i)|^2 / 2  ==  sum(res_i^2)/6          (Parseval: no FFT needed)
//   else      : f_i = mean|FFT2(res_i)|        -> real 2-D FFT, line by line through LDS
//   loss_T   += sigma*(rmse + sum_i f_i) [+ Sigma*mean|T(x)-y|]
// Elementwise / bandwidth-bound work, no MFMA.  The spectrum branch runs three line-FFT passes
// (rows forward; columns forward -> |F|, F/|F| -> columns inverse; rows inverse) over an L2-resident
// complex scratch; each wavefront owns one line in LDS (radix-2, in place).
#include "common.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace {

constexpr int MAXP = 1024;         // longest FFT line
constexpr int LPB = 4;             // lines per block (one per wavefront)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// In-place radix-2 DIT FFT of one line of length P = 1<<logP held in LDS (owned by one wavefront).
// sign = -1 forward, +1 inverse (unnormalised).
__device__ void fft_line(float2* line, int P, int logP, float sign, int lane) {
    for (int i = lane; i < P; i += 64) {
        const int r = (int)(__brev((unsigned)i) >> (32 - logP));
        if (i < r) { const float2 t = line[i]; line[i] = line[r]; line[r] = t; }
    }
    __syncthreads();
    for (int s = 0; s < logP; ++s) {
        const int half = 1 << s;
        for (int q = lane; q < (P >> 1); q += 64) {
            const int grp = q >> s, pos = q & (half - 1);
            const int i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
            float sn, cs;
            sincospif(sign * (float)pos / (float)half, &sn, &cs);
            const float2 a = line[i0], bt = cmul(line[i1], make_float2(cs, sn));
            line[i0] = make_float2(a.x + bt.x, a.y + bt.y);
            line[i1] = make_float2(a.x - bt.x, a.y - bt.y);
        }
        __syncthreads();
    }
}

// sums[b] += sum res^2 ; sums[B+b] += sum |out-tgt| ; sums[2B] += total res^2 ; sums[2B+1] += total |out-tgt|
__global__ __launch_bounds__(256) void ot_reduce_kernel(const float* __restrict__ deg, const float* __restrict__ out,
                                                        const float* __restrict__ tgt, float* __restrict__ sums, int B,
                                                        long per) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long base = (long)b * per;
    float s2 = 0.f, l1 = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const float o = out[base + i];
        const float r = deg[base + i] - o;
        s2 += r * r;
        if (tgt) l1 += fabsf(o - tgt[base + i]);
    }
    s2 = block_sum<256>(s2, red);
    l1 = block_sum<256>(l1, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[b], s2);
        atomicAdd(&sums[2 * B], s2);
        if (tgt) { atomicAdd(&sums[B + b], l1); atomicAdd(&sums[2 * B + 1], l1); }
    }
}

// pass 1: forward FFT of every row of res (planes of samples with de_id >= 3 only)
__global__ __launch_bounds__(256) void ot_rows_fwd_kernel(const float* __restrict__ deg, const float* __restrict__ out,
                                                          const int* __restrict__ de_id, float2* __restrict__ scr,
                                                          int H, int W, int logW) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int plane = blockIdx.y, b = plane / 3;
    if (de_id[b] < 3) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * LPB + wave;
    float2* line = sm + wave * W;
    const long base = ((long)plane * H + row) * W;
    if (row < H)
        for (int i = lane; i < W; i += 64) line[i] = make_float2(deg[base + i] - out[base + i], 0.f);
    __syncthreads();
    fft_line(line, W, logW, -1.f, lane);
    if (row < H)
        for (int i = lane; i < W; i += 64) scr[base + i] = line[i];
}

// pass 2: forward FFT of every column -> |F| (sum into spec[b]) and U = F/|F| -> inverse FFT along the column
__global__ __launch_bounds__(256) void ot_cols_kernel(const int* __restrict__ de_id, float2* __restrict__ scr,
                                                      float* __restrict__ spec, int H, int W, int logH) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    __shared__ float red[4];
    const int plane = blockIdx.y, b = plane / 3;
    if (de_id[b] < 3) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = blockIdx.x * LPB + wave;
    float2* line = sm + wave * H;
    float2* p = scr + (long)plane * H * W + col;
    if (col < W)
        for (int i = lane; i < H; i += 64) line[i] = p[(long)i * W];
    __syncthreads();
    fft_line(line, H, logH, -1.f, lane);
    float s = 0.f;
    if (col < W)
        for (int i = lane; i < H; i += 64) {
            const float2 f = line[i];
            const float mag = sqrtf(f.x * f.x + f.y * f.y);
            s += mag;
            line[i] = mag > 0.f ? make_float2(f.x / mag, f.y / mag) : make_float2(0.f, 0.f);
        }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) atomicAdd(&spec[b], s);
    __syncthreads();
    fft_line(line, H, logH, 1.f, lane);
    if (col < W)
        for (int i = lane; i < H; i += 64) p[(long)i * W] = line[i];
}

// pass 3: inverse FFT of every row; gF = Re(.) / (3*H*W)   ( == Re(ifft2(U)) / 3 )
__global__ __launch_bounds__(256) void ot_rows_inv_kernel(const int* __restrict__ de_id, const float2* __restrict__ scr,
                                                          float* __restrict__ gF, int H, int W, int logW) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int plane = blockIdx.y, b = plane / 3;
    if (de_id[b] < 3) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * LPB + wave;
    float2* line = sm + wave * W;
    const long base = ((long)plane * H + row) * W;
    if (row < H)
        for (int i = lane; i < W; i += 64) line[i] = scr[base + i];
    __syncthreads();
    fft_line(line, W, logW, 1.f, lane);
    const float sc = 1.0f / (3.0f * (float)H * (float)W);
    if (row < H)
        for (int i = lane; i < W; i += 64) gF[base + i] = line[i].x * sc;
}

// dout += -sigma*( res/(Mg*rmse) + [de_id<3 ? res/3 : gF] ) + Sigma*sign(out-tgt)/Mg
// scal[0]=rmse (global), scal[1]=local Fourier penalty sum, scal[2]=local sum|out-tgt| / Mg
__global__ __launch_bounds__(256) void ot_grad_kernel(const float* __restrict__ deg, const float* __restrict__ out,
                                                      const float* __restrict__ tgt, const int* __restrict__ de_id,
                                                      const float* __restrict__ gF, const float* __restrict__ sums,
                                                      const float* __restrict__ spec, float* __restrict__ dout,
                                                      float* __restrict__ scal, int B, long per, float sigma,
                                                      float Sigma, float Mg) {
    const float tot = sums[2 * B];
    const float rmse = sqrtf(tot / Mg);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float four = 0.f;
        for (int i = 0; i < B; ++i) four += de_id[i] < 3 ? sums[i] / 6.0f : spec[i] / (float)per;
        scal[0] = rmse;
        scal[1] = four;
        scal[2] = sums[2 * B + 1] / Mg;
    }
    const int b = blockIdx.y;
    const bool l2 = de_id[b] < 3;
    const long base = (long)b * per;
    const float k_rmse = rmse > 0.f ? 1.0f / (Mg * rmse) : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const float o = out[base + i];
        const float r = deg[base + i] - o;
        float g = -sigma * (r * k_rmse + (l2 ? r * (1.0f / 3.0f) : gF[base + i]));
        if (tgt) {
            const float d = o - tgt[base + i];
            g += Sigma / Mg * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        dout[base + i] += g;
    }
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

}  // namespace

extern "C" {

int rcot_ot_reduce(const float* degraded, const float* restored, const float* target, float* sums, int B, long per,
                   void* stream) {
    if (!degraded || !restored || !sums || B <= 0 || per <= 0 || B > 65535) return RCOT_EINVAL;
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * (2 * (size_t)B + 2), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    int gx = (int)((per + 256 * 8 - 1) / (256 * 8));
    if (gx < 1) gx = 1;
    RCOT_LAUNCH(ot_reduce_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, degraded, restored, target, sums,
                       B, per);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_ot_spectrum(const float* degraded, const float* restored, const int* de_id, float* gF, float* spec, float* ws,
                     size_t ws_bytes, int B, int H, int W, void* stream) {
    if (!degraded || !restored || !de_id || !gF || !spec || !ws || B <= 0 || B * 3 > 65535) return RCOT_EINVAL;
    const int lh = ilog2(H), lw = ilog2(W);
    if (lh < 1 || lw < 1 || H > MAXP || W > MAXP) return RCOT_EINVAL;
    if (ws_bytes < sizeof(float2) * (size_t)B * 3 * H * W) return RCOT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(spec, 0, sizeof(float) * (size_t)B, st);
    if (e != hipSuccess) return (int)e;
    float2* scr = reinterpret_cast<float2*>(ws);
    RCOT_LAUNCH(ot_rows_fwd_kernel, dim3(cdiv(H, LPB), B * 3), dim3(256), sizeof(float2) * LPB * W, st, degraded,
                       restored, de_id, scr, H, W, lw);
    RCOT_LAUNCH_CHECK();
    RCOT_LAUNCH(ot_cols_kernel, dim3(cdiv(W, LPB), B * 3), dim3(256), sizeof(float2) * LPB * H, st, de_id, scr, spec,
                       H, W, lh);
    RCOT_LAUNCH_CHECK();
    RCOT_LAUNCH(ot_rows_inv_kernel, dim3(cdiv(H, LPB), B * 3), dim3(256), sizeof(float2) * LPB * W, st, de_id, scr,
                       gF, H, W, lw);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_ot_grad(const float* degraded, const float* restored, const float* target, const int* de_id, const float* gF,
                 const float* sums, const float* spec, float* dout, float* scal, int B, long per, float sigma,
                 float Sigma, long global_batch, void* stream) {
    if (!degraded || !restored || !de_id || !sums || !spec || !dout || !scal || B <= 0 || per <= 0 || global_batch <= 0 ||
        B > 65535)
        return RCOT_EINVAL;
    int gx = (int)((per + 256 * 4 - 1) / (256 * 4));
    if (gx < 1) gx = 1;
    RCOT_LAUNCH(ot_grad_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, degraded, restored, target, de_id,
                       gF, sums, spec, dout, scal, B, per, sigma, Sigma, (float)global_batch * (float)per);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // extern "C"
