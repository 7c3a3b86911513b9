// MDTA's channel-attention core works on tiny (C/heads)^2 matrices per (image, head); the H*W-long
// reductions are done by the GEMM engine (rcot_bmm_nt), everything in between lives here:
//   forward : Graw, |q|^2, |k|^2  -> Gn = Graw/(nq nk^T), A = softmax_rows(tau*Gn),
//             Mf = W_o * blockdiag(A)           (so that  y = Mf * V  is ONE 1x1 projection)
//   backward: dMf -> dW_o (per-image partial), dA, dS, dtau (partial), and the small matrices that turn
//             the gradient wrt the normalised q,k back into  dQ = Eq*K + Dq.Q ,  dK = Eq^T*Q + Dk.K .
// One workgroup per (head, image); rows of the c x c matrices are owned by wavefronts and reduced with
// wave shuffles (reference: Net_Restormer.py:39-49; math: SURVEY.md Appendix A.2).
#include "common.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace {

constexpr int CMAX = 96;
constexpr int LDA = CMAX + 1;

__device__ __forceinline__ float clamp_norm(float sumsq) { return fmaxf(sqrtf(sumsq), 1e-12f); }

__global__ __launch_bounds__(256) void attn_fwd_small_kernel(const float* __restrict__ Graw, const float* __restrict__ sq,
                                                             const float* __restrict__ temp, const float* __restrict__ Wo,
                                                             float* __restrict__ Gn, float* __restrict__ A,
                                                             float* __restrict__ Mf, int heads, int c) {
    __shared__ float S[CMAX * LDA];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = heads * c;
    const long off = ((long)b * heads + h) * c * c;
    const float tau = temp[h];
    const float* sqq = sq + (long)b * 2 * C + h * c;
    const float* sqk = sqq + C;
    for (int e = tid; e < c * c; e += 256) {
        const int i = e / c, j = e - i * c;
        const float g = Graw[off + e] / (clamp_norm(sqq[i]) * clamp_norm(sqk[j]));
        Gn[off + e] = g;
        S[i * LDA + j] = g * tau;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < c; i += 4) {
        const float v0 = lane < c ? S[i * LDA + lane] : -INFINITY;
        const float v1 = lane + 64 < c ? S[i * LDA + lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = lane < c ? expf(v0 - mx) : 0.f;
        const float e1 = lane + 64 < c ? expf(v1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < c) { S[i * LDA + lane] = e0 * inv; A[off + i * c + lane] = e0 * inv; }
        if (lane + 64 < c) { S[i * LDA + lane + 64] = e1 * inv; A[off + i * c + lane + 64] = e1 * inv; }
    }
    __syncthreads();
    // Mf[b][m][h*c + j] = sum_i Wo[m][h*c + i] * A[i][j]
    float* Mb = Mf + (long)b * C * C;
    for (int e = tid; e < C * c; e += 256) {
        const int m = e / c, j = e - m * c;
        const float* wrow = Wo + (long)m * C + h * c;
        float acc = 0.f;
        for (int i = 0; i < c; ++i) acc += wrow[i] * S[i * LDA + j];
        Mb[(long)m * C + h * c + j] = acc;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ dM, const float* __restrict__ Wo,
                                                             const float* __restrict__ A, const float* __restrict__ Gn,
                                                             const float* __restrict__ sq, const float* __restrict__ temp,
                                                             float* __restrict__ dWo_part, float* __restrict__ dtemp_part,
                                                             float* __restrict__ Eq, float* __restrict__ Dq,
                                                             float* __restrict__ Dk, int heads, int c) {
    __shared__ float As[CMAX * LDA];
    __shared__ float Ds[CMAX * LDA];
    __shared__ float red[4];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = heads * c;
    const long off = ((long)b * heads + h) * c * c;
    const float tau = temp[h];
    const float* dMb = dM + (long)b * C * C;
    for (int e = tid; e < c * c; e += 256) {
        const int i = e / c, j = e - i * c;
        As[i * LDA + j] = A[off + e];
    }
    // dA[i][j] = sum_m Wo[m][hc+i] * dM[m][hc+j]
    for (int e = tid; e < c * c; e += 256) {
        const int i = e / c, j = e - i * c;
        float acc = 0.f;
        for (int m = 0; m < C; ++m) acc += Wo[(long)m * C + h * c + i] * dMb[(long)m * C + h * c + j];
        Ds[i * LDA + j] = acc;
    }
    __syncthreads();
    // dWo_part[b][m][hc+i] = sum_j dM[m][hc+j] * A[i][j]
    float* dWb = dWo_part + (long)b * C * C;
    for (int e = tid; e < C * c; e += 256) {
        const int m = e / c, i = e - m * c;
        const float* drow = dMb + (long)m * C + h * c;
        float acc = 0.f;
        for (int j = 0; j < c; ++j) acc += drow[j] * As[i * LDA + j];
        dWb[(long)m * C + h * c + i] = acc;
    }
    // dS = A .* (dA - rowsum(dA .* A))
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < c; i += 4) {
        const float a0 = lane < c ? As[i * LDA + lane] : 0.f, d0 = lane < c ? Ds[i * LDA + lane] : 0.f;
        const float a1 = lane + 64 < c ? As[i * LDA + lane + 64] : 0.f, d1 = lane + 64 < c ? Ds[i * LDA + lane + 64] : 0.f;
        const float rsum = wave_sum(a0 * d0 + a1 * d1);
        if (lane < c) Ds[i * LDA + lane] = a0 * (d0 - rsum);
        if (lane + 64 < c) Ds[i * LDA + lane + 64] = a1 * (d1 - rsum);
    }
    __syncthreads();
    // from here As holds dS .* Gn ; Ds holds dS
    const float* sqq = sq + (long)b * 2 * C + h * c;
    const float* sqk = sqq + C;
    float part = 0.f;
    for (int e = tid; e < c * c; e += 256) {
        const int i = e / c, j = e - i * c;
        const float ds = Ds[i * LDA + j];
        const float sg = ds * Gn[off + e];
        part += sg;
        As[i * LDA + j] = sg;
        Eq[off + e] = tau * ds / (clamp_norm(sqq[i]) * clamp_norm(sqk[j]));
    }
    part = block_sum<256>(part, red);      // (contains the barrier that publishes As)
    if (tid == 0) dtemp_part[(long)b * heads + h] = part;
    // Dq_i = -tau * sum_j (dS.*Gn)[i][j] / nq_i^2 ; Dk_j = -tau * sum_i (dS.*Gn)[i][j] / nk_j^2
    for (int i = wave; i < c; i += 4) {
        const float r0 = lane < c ? As[i * LDA + lane] : 0.f;
        const float r1 = lane + 64 < c ? As[i * LDA + lane + 64] : 0.f;
        const float c0 = lane < c ? As[lane * LDA + i] : 0.f;
        const float c1 = lane + 64 < c ? As[(lane + 64) * LDA + i] : 0.f;
        const float rs_ = wave_sum(r0 + r1), cs_ = wave_sum(c0 + c1);
        if (lane == 0) {
            const float q2 = sqq[i], k2 = sqk[i];
            Dq[(long)b * C + h * c + i] = q2 >= 1e-24f ? -tau * rs_ / q2 : 0.f;
            Dk[(long)b * C + h * c + i] = k2 >= 1e-24f ? -tau * cs_ / k2 : 0.f;
        }
    }
}

// dst = beta*dst + sum_b src[b*n + i]
__global__ void batch_reduce_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, long n, float beta) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += src[(long)b * n + i];
        dst[i] = (beta != 0.f ? beta * dst[i] : 0.f) + s;
    }
}

}  // namespace

extern "C" {

int rcot_attn_fwd_small(const float* Graw, const float* sq, const float* temp, const float* Wo, float* Gn, float* A,
                        float* Mf, int B, int heads, int c, void* stream) {
    if (!Graw || !sq || !temp || !Wo || !Gn || !A || !Mf || B <= 0 || heads <= 0 || c <= 0 || c > CMAX || B > 65535)
        return RCOT_EINVAL;
    hipLaunchKernelGGL(attn_fwd_small_kernel, dim3(heads, B), dim3(256), 0, (hipStream_t)stream, Graw, sq, temp, Wo, Gn,
                       A, Mf, heads, c);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_attn_bwd_small(const float* dM, const float* Wo, const float* A, const float* Gn, const float* sq,
                        const float* temp, float* dWo_part, float* dtemp_part, float* Eq, float* Dq, float* Dk, int B,
                        int heads, int c, void* stream) {
    if (!dM || !Wo || !A || !Gn || !sq || !temp || !dWo_part || !dtemp_part || !Eq || !Dq || !Dk || B <= 0 ||
        heads <= 0 || c <= 0 || c > CMAX || B > 65535)
        return RCOT_EINVAL;
    hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(heads, B), dim3(256), 0, (hipStream_t)stream, dM, Wo, A, Gn, sq, temp,
                       dWo_part, dtemp_part, Eq, Dq, Dk, heads, c);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_batch_reduce(const float* src, float* dst, int B, long n, float beta, void* stream) {
    if (!src || !dst || B <= 0 || n <= 0) return RCOT_EINVAL;
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(batch_reduce_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, src, dst, B, n, beta);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // extern "C"
