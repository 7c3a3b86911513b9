// MDTA's channel-attention core works on tiny (C/heads)^2 matrices per (image, head); the H*W-long
// reductions are done by the GEMM engine (rcot_bmm_nt), everything in between lives here:
//   forward : Graw, |q|^2, |k|^2  -> Gn = Graw/(nq nk^T), A = softmax_rows(tau*Gn)
//             (Mf = W_o * blockdiag(A), which makes  y = Mf * V  ONE 1x1 projection, is an rcot_bmm_nn call)
//   backward: dA (= W_o^T dMf per head block, an rcot_bmm_nn call) -> dS, dtau (partial), and the small
//             matrices that turn the gradient wrt the normalised q,k back into
//             dQ = Eq*K + Dq.Q ,  dK = Eq^T*Q + Dk.K .
// One workgroup per (head, image); rows of the c x c matrices are owned by wavefronts and reduced with
// wave shuffles (reference: Net_Restormer.py:39-49; math: SURVEY.md Appendix A.2).
#include "common.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace {

constexpr int CMAX = 96;
constexpr int LDA = CMAX + 1;

__device__ __forceinline__ float clamp_norm(float sumsq) { return fmaxf(sqrtf(sumsq), 1e-12f); }

// S > 0: Graw is the split-K slab set [z][S][c][ld] of rcot_bmm_nt_slabs and is summed here (fixed order) — the Gram product
// then needs no reduce launch of its own.  Grid (rows, heads, images), four wavefronts per row: each sums a quarter of the slabs
// (S can be 160: one wavefront alone would chain 40 dependent load batches), wavefront 0 finishes the row.
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ Graw, int S, int ld,
                                                           const float* __restrict__ sq,
                                                           const float* __restrict__ temp, float* __restrict__ Gn,
                                                           float* __restrict__ A, int heads, int c) {
    __shared__ float part[4][128];
    const int h = blockIdx.y, b = blockIdx.z;
    const int C = heads * c;
    const long off = ((long)b * heads + h) * c * c;
    const float tau = temp[h];
    const float* sqq = sq + (long)b * 2 * C + h * c;
    const float* sqk = sqq + C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x;
    const bool a0 = lane < c, a1 = lane + 64 < c;
    float r0 = 0.f, r1 = 0.f;
    if (S > 0) {
        const long slab = (long)c * ld;
        const float* w = Graw + ((long)b * heads + h) * S * slab + (long)i * ld + lane;
        float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
        int s = wave;
        for (; s + 12 < S; s += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (a0) q0[u] += w[(long)(s + 4 * u) * slab];
                if (a1) q1[u] += w[(long)(s + 4 * u) * slab + 64];
            }
        }
        for (; s < S; s += 4) {
            if (a0) q0[0] += w[(long)s * slab];
            if (a1) q1[0] += w[(long)s * slab + 64];
        }
        part[wave][lane] = (q0[0] + q0[1]) + (q0[2] + q0[3]);
        part[wave][lane + 64] = (q1[0] + q1[1]) + (q1[2] + q1[3]);
        __syncthreads();
        if (wave != 0) return;
        r0 = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        r1 = (part[0][lane + 64] + part[1][lane + 64]) + (part[2][lane + 64] + part[3][lane + 64]);
    } else {
        if (wave != 0) return;
        r0 = a0 ? Graw[off + i * c + lane] : 0.f;
        r1 = a1 ? Graw[off + i * c + lane + 64] : 0.f;
    }
    const float k0 = a0 ? clamp_norm(sqk[lane]) : 1.f, k1 = a1 ? clamp_norm(sqk[lane + 64]) : 1.f;
    const float nq = clamp_norm(sqq[i]);
    const float g0 = a0 ? r0 / (nq * k0) : 0.f;
    const float g1 = a1 ? r1 / (nq * k1) : 0.f;
    const float v0 = a0 ? g0 * tau : -INFINITY, v1 = a1 ? g1 * tau : -INFINITY;
    const float mx = wave_max(fmaxf(v0, v1));
    const float e0 = a0 ? expf(v0 - mx) : 0.f, e1 = a1 ? expf(v1 - mx) : 0.f;
    const float inv = 1.0f / wave_sum(e0 + e1);
    if (a0) { Gn[off + i * c + lane] = g0; A[off + i * c + lane] = e0 * inv; }
    if (a1) { Gn[off + i * c + lane + 64] = g1; A[off + i * c + lane + 64] = e1 * inv; }
}

// From dA (= W_o^T dM restricted to the head block): dS = A.*(dA - rowsum(dA.*A)); dtau partial; Eq; Dq; Dk.
// The three c x c inputs are first staged into LDS with all loads in flight at once (the kernel is pure latency:
// 8..64 workgroups), then rows are processed by wavefronts from LDS.
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ dA, const float* __restrict__ A,
                                                             const float* __restrict__ Gn, const float* __restrict__ sq,
                                                             const float* __restrict__ temp, float* __restrict__ dtemp_part,
                                                             float* __restrict__ Eq, float* __restrict__ EqT,
                                                             float* __restrict__ Dq, float* __restrict__ Dk, int heads, int c,
                                                             int nparts) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As_ = sm;                       // A, later dS.*Gn
    float* Ds_ = sm + CMAX * LDA;          // dA  (= sum of nparts partial matrices [b,h][part][c][c])
    float* Gs_ = sm + 2 * CMAX * LDA;      // Gn
    float* red = sm + 3 * CMAX * LDA;      // 4 floats
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = heads * c;
    const long off = ((long)b * heads + h) * c * c;
    const float tau = temp[h];
    const float* sqq = sq + (long)b * 2 * C + h * c;
    const float* sqk = sqq + C;
    // staging with many loads in flight (the kernel is pure latency): 4 elements per thread and pass, every load of a
    // pass issued before the first LDS store
    for (int e0 = tid; e0 < c * c; e0 += 1024) {
        float av[4], gv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            const bool ok = e < c * c;
            av[u] = ok ? A[off + e] : 0.f;
            gv[u] = ok ? Gn[off + e] : 0.f;
            dv[u] = 0.f;
        }
        for (int q = 0; q < nparts; ++q) {
            const float* dq = dA + (off * nparts) + (long)q * c * c;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256;
                dv[u] += e < c * c ? dq[e] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            if (e < c * c) {
                const int i = e / c, j = e - i * c;
                As_[i * LDA + j] = av[u];
                Ds_[i * LDA + j] = dv[u];
                Gs_[i * LDA + j] = gv[u];
            }
        }
    }
    __syncthreads();
    // rows are owned by 16-lane groups (16 rows in flight per workgroup, 6 columns per lane at c = 96); sums over a
    // row stay inside the group (xor 8,4,2,1)
    const int grp = tid >> 4, gl = tid & 15;
    constexpr int NT_ = CMAX / 16;
    float kn[NT_];
#pragma unroll
    for (int t = 0; t < NT_; ++t) kn[t] = (gl + 16 * t < c) ? clamp_norm(sqk[gl + 16 * t]) : 1.f;
    float part = 0.f;
    for (int i = grp; i < c; i += 16) {
        float pv[NT_], dv[NT_], acc = 0.f;
#pragma unroll
        for (int t = 0; t < NT_; ++t) {
            const int j = gl + 16 * t;
            pv[t] = j < c ? As_[i * LDA + j] : 0.f;
            dv[t] = j < c ? Ds_[i * LDA + j] : 0.f;
            acc += pv[t] * dv[t];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        const float nq = clamp_norm(sqq[i]);
#pragma unroll
        for (int t = 0; t < NT_; ++t) {
            const int j = gl + 16 * t;
            if (j < c) {
                const float sv = pv[t] * (dv[t] - acc);
                const float sg = sv * Gs_[i * LDA + j];
                part += sg;
                As_[i * LDA + j] = sg;
                const float e = tau * sv / (nq * kn[t]);
                Eq[off + i * c + j] = e;
                Ds_[i * LDA + j] = e;
            }
        }
    }
    part = block_sum<256>(part, red);      // (its barriers also publish As_ / Ds_)
    if (tid == 0) dtemp_part[(long)b * heads + h] = part;
    for (int i = grp; i < c; i += 16) {
        float rs_ = 0.f, cs_ = 0.f;
#pragma unroll
        for (int t = 0; t < NT_; ++t) {
            const int j = gl + 16 * t;
            if (j < c) {
                rs_ += As_[i * LDA + j];
                cs_ += As_[j * LDA + i];
                EqT[off + (long)i * c + j] = Ds_[j * LDA + i];      // row i of Eq^T = column i of Eq
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            rs_ += __shfl_xor(rs_, o, 64);
            cs_ += __shfl_xor(cs_, o, 64);
        }
        if (gl == 0) {
            const float q2 = sqq[i], k2 = sqk[i];
            Dq[(long)b * C + h * c + i] = q2 >= 1e-24f ? -tau * rs_ / q2 : 0.f;
            Dk[(long)b * C + h * c + i] = k2 >= 1e-24f ? -tau * cs_ / k2 : 0.f;
        }
    }
}

// ---- the three small products of the attention-matrix backward, one workgroup per (head, image, ROW CHUNK).  With
// W = W_o[:, head block] (C x c), D = dM[b][:, head block] (C x c) and A = A[b,h] (c x c), for the R rows of the chunk:
//     Mf[b][rows, head block] = W[rows] A              (K-major operand of dV = Mf^T dY)
//     dW_o part[b][rows, head block] = D[rows] A^T
//     dA part[b,h][chunk] = W[rows]^T D[rows]           (summed over chunks by attn_bwd_small_kernel while staging)
// A 16 x 16 thread grid holds (R/16) x (c/16) and (c/16)^2 register tiles over LDS copies of A, W[rows], D[rows].
// Replaces three rcot_bmm_* launches (each latency-bound on 8..64 workgroups) by one.
template <int CT>                          // c = 16 * CT  (48 or 96)
__global__ __launch_bounds__(256) void attn_bwd_chunk_kernel(const float* __restrict__ dM, const float* __restrict__ Wo,
                                                             const float* __restrict__ A, float* __restrict__ Mf,
                                                             float* __restrict__ dWo_part, float* __restrict__ dA_part,
                                                             int heads) {
    constexpr int c = 16 * CT, R = 3072 / c, RT = R / 16, LD = c + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As_ = sm;
    float* Ws = As_ + c * LD;              // W chunk  [R][LD]
    float* Dm = Ws + R * LD;               // D chunk  [R][LD]
    const int h = blockIdx.x, b = blockIdx.y, ch = blockIdx.z, tid = threadIdx.x;
    const int C = heads * c, m0 = ch * R;
    const int ty = tid >> 4, tx = tid & 15;
    const long off = ((long)b * heads + h) * c * c;
    const float* Wh = Wo + h * c;                               // W[m][i] = Wh[m*C + i]
    const float* Dh = dM + (long)b * C * C + h * c;
    float* Mfh = Mf + (long)b * C * C + h * c;
    float* dWh = dWo_part + (long)b * C * C + h * c;
#pragma unroll 4
    for (int e = tid; e < c * c; e += 256) {                  // c*c and R*c are multiples of 256: full unrolled passes
        const int i = e / c, j = e - i * c;
        As_[i * LD + j] = A[off + e];
    }
#pragma unroll 4
    for (int e = tid; e < R * c; e += 256) {
        const int r = e / c, i = e - r * c;
        const bool ok = m0 + r < C;
        Ws[r * LD + i] = ok ? Wh[(long)(m0 + r) * C + i] : 0.f;
        Dm[r * LD + i] = ok ? Dh[(long)(m0 + r) * C + i] : 0.f;
    }
    __syncthreads();
    float acc1[RT][CT], acc3[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < CT; ++j) { acc1[r][j] = 0.f; acc3[r][j] = 0.f; }
    for (int k = 0; k < c; ++k) {
        float wv[RT], dv[RT], a1[CT], a3[CT];
#pragma unroll
        for (int r = 0; r < RT; ++r) { wv[r] = Ws[(ty * RT + r) * LD + k]; dv[r] = Dm[(ty * RT + r) * LD + k]; }
#pragma unroll
        for (int j = 0; j < CT; ++j) { a1[j] = As_[k * LD + tx * CT + j]; a3[j] = As_[(tx * CT + j) * LD + k]; }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                acc1[r][j] = fmaf(wv[r], a1[j], acc1[r][j]);
                acc3[r][j] = fmaf(dv[r], a3[j], acc3[r][j]);
            }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int m = m0 + ty * RT + r;
        if (m < C) {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                Mfh[(long)m * C + tx * CT + j] = acc1[r][j];
                dWh[(long)m * C + tx * CT + j] = acc3[r][j];
            }
        }
    }
    float acc2[CT][CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc2[i][j] = 0.f;
    for (int r = 0; r < R; ++r) {
        float wv[CT], dv[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) { wv[i] = Ws[r * LD + ty * CT + i]; dv[i] = Dm[r * LD + tx * CT + i]; }
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc2[i][j] = fmaf(wv[i], dv[j], acc2[i][j]);
    }
    float* dp = dA_part + (off * gridDim.z) + (long)ch * c * c;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) dp[(ty * CT + i) * c + tx * CT + j] = acc2[i][j];
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

int rcot_attn_softmax(const float* Graw, int S, int ld, const float* sq, const float* temp, float* Gn, float* A, int B,
                      int heads, int c, void* stream) {
    if (!Graw || !sq || !temp || !Gn || !A || B <= 0 || heads <= 0 || c <= 0 || c > CMAX || B > 65535 || S < 0 || (S > 0 && ld < c))
        return RCOT_EINVAL;
    RCOT_LAUNCH(attn_softmax_kernel, dim3(c, heads, B), dim3(256), 0, (hipStream_t)stream, Graw, S, ld, sq, temp,
                       Gn, A, heads, c);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_attn_bwd_small(const float* dA, const float* A, const float* Gn, const float* sq, const float* temp,
                        float* dtemp_part, float* Eq, float* EqT, float* Dq, float* Dk, int B, int heads, int c,
                        void* stream) {
    if (!dA || !A || !Gn || !sq || !temp || !dtemp_part || !Eq || !EqT || !Dq || !Dk || B <= 0 || heads <= 0 || c <= 0 ||
        c > CMAX || B > 65535)
        return RCOT_EINVAL;
    static bool once = (hipFuncSetAttribute((const void*)attn_bwd_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024) == hipSuccess);
    (void)once;
    const size_t smem = sizeof(float) * (3 * CMAX * LDA + 4);
    RCOT_LAUNCH(attn_bwd_small_kernel, dim3(heads, B), dim3(256), smem, (hipStream_t)stream, dA, A, Gn, sq, temp,
                       dtemp_part, Eq, EqT, Dq, Dk, heads, c, 1);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_attn_bwd_fused(const float* dM, const float* Wo, const float* A, const float* Gn, const float* sq, const float* temp,
                        float* Mf, float* dWo_part, float* dtemp_part, float* Eq, float* EqT, float* Dq, float* Dk, int B,
                        int heads, int c, void* ws, long ws_bytes, void* stream) {
    if (!dM || !Wo || !A || !Gn || !sq || !temp || !Mf || !dWo_part || !dtemp_part || !Eq || !EqT || !Dq || !Dk || !ws ||
        B <= 0 || heads <= 0 || B > 65535)
        return RCOT_EINVAL;
    if (c != 48 && c != 96) return RCOT_EINVAL;              // the head widths of the Restormer configuration
    const int C = heads * c, R = 3072 / c, LD = c + 1;
    const int nch = cdiv(C, R);
    if (nch > 65535 || (long)B * heads * nch * c * c * (long)sizeof(float) > ws_bytes) return RCOT_EINVAL;
    float* dA_part = static_cast<float*>(ws);
    const size_t smem = sizeof(float) * ((size_t)c * LD + 2 * R * LD);
    const dim3 grid(heads, B, nch);
    if (c == 48) {
        static bool once = (hipFuncSetAttribute((const void*)attn_bwd_chunk_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                160 * 1024) == hipSuccess);
        (void)once;
        RCOT_LAUNCH(attn_bwd_chunk_kernel<3>, grid, dim3(256), smem, (hipStream_t)stream, dM, Wo, A, Mf, dWo_part, dA_part,
                           heads);
    } else {
        static bool once = (hipFuncSetAttribute((const void*)attn_bwd_chunk_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                160 * 1024) == hipSuccess);
        (void)once;
        RCOT_LAUNCH(attn_bwd_chunk_kernel<6>, grid, dim3(256), smem, (hipStream_t)stream, dM, Wo, A, Mf, dWo_part, dA_part,
                           heads);
    }
    RCOT_LAUNCH_CHECK();
    static bool once2 = (hipFuncSetAttribute((const void*)attn_bwd_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024) == hipSuccess);
    (void)once2;
    const size_t smem2 = sizeof(float) * (3 * CMAX * LDA + 4);
    RCOT_LAUNCH(attn_bwd_small_kernel, dim3(heads, B), dim3(256), smem2, (hipStream_t)stream, dA_part, A, Gn, sq, temp,
                       dtemp_part, Eq, EqT, Dq, Dk, heads, c, nch);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_batch_reduce(const float* src, float* dst, int B, long n, float beta, void* stream) {
    if (!src || !dst || B <= 0 || n <= 0) return RCOT_EINVAL;
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    RCOT_LAUNCH(batch_reduce_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, src, dst, B, n, beta);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // extern "C"
