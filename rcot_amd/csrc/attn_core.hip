// MDTA attention core for the levels whose images are small (H*W <= 4096: the 64x64, 32x32 and 16x16 levels of T_net).
//
// Forward (Net_Restormer.py:39-45,49), from u = dwconv3x3(qkv(LN x)) = [q | k | v]:
//     |q_i|^2, |k_j|^2 (F.normalize),  G = q k^T,  Gn = G / (|q| |k|^T),  A = softmax_rows(tau Gn),
//     MfT = (W_o blockdiag(A))^T      (the K-major operand of  y = x + MfT^T v,  one projection instead of attn@v + project_out)
// used to be four launches (rcot_row_sumsq, rcot_bmm_nt[_slabs], rcot_attn_softmax, rcot_bmm_nn), every one of them a few
// microseconds of work behind a launch: on these levels a transformer block is launch latency, not traffic.  Here:
//   * qk_stats_kernel: one workgroup per (pixel range, head, image) stages 256-pixel chunks of its q and k rows in LDS once and
//     takes both the row sums of squares and the Gram block from them — exact fp32 on v_mfma_f32_16x16x4_f32 (the product is
//     c x c x pixels: too small to be worth a bf16 split);
//   * with ONE pixel range per image (16x16 level) the same workgroup goes on to the softmax and the W_o fold: one launch;
//   * otherwise the partial Gram blocks / sums go to the workspace and softmax_fold_kernel — one workgroup per (head, image,
//     slice of output columns) — adds them in a fixed order, then softmax + fold (also MFMA fp32).
#include <cstdlib>
#include "common.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float clamp_norm(float sumsq) { return fmaxf(sqrtf(sumsq), 1e-12f); }

// These kernels run once per wavefront through code that a cold instruction cache has to fetch: they are written for FEW
// instructions (rolled loops wherever the data does not have to be in flight at once, no integer divisions) — a first, fully
// unrolled version (3500 instructions) spent 12 us per launch before doing any work.
template <int CT>                                   // channels per head padded to 16 CT (c = 24 -> 2, 48 -> 3, 96 -> 6)
struct AC {
    static constexpr int cp = 16 * CT;
    static constexpr int c = CT == 2 ? 24 : cp;                   // the head widths of T_net (compile-time: every bound check folds)
    static constexpr int LDA = (cp % 32 == 16) ? cp : cp + 16;    // == 16 (mod 32): the 4 k-rows x 16 columns of an MFMA operand hit 64 banks
    static constexpr int PX = CT == 6 ? 128 : 256;                // pixels per LDS chunk
    static constexpr int Q4 = PX / 4;
    static constexpr int LDQ = PX + 8;                            // == 8 (mod 64): conflict-free ds_read_b128 of 16 rows x 4 pixel quads
    static constexpr int NW = CT == 6 ? 8 : 4;
    static constexpr int NT = 64 * NW;
    static constexpr int RPP = NT / Q4;                           // rows one pass of the workgroup loads (4; 16 for c = 96)
    static constexpr int KQ = cp / RPP;                           // passes per operand (q rows, then k rows)
    static constexpr int MTW = CT == 6 ? 1 : 2;                   // 16-column tiles of the fold per wavefront
    static constexpr int MT_WG = NW * MTW;                        // ... per workgroup (column slice)
    static constexpr int TILE = 2 * cp * LDQ;
    static constexpr int RW = NW * cp * cp <= TILE ? NW : TILE / (cp * cp);   // wavefronts whose Gram blocks fit the pixel tile at once
    static_assert(RW >= 1 && NW % RW == 0 && cp % RPP == 0 && c % RPP == 0 && c % NW == 0, "tiling");
    static constexpr size_t smem_stats = sizeof(float) * (size_t)(TILE + cp * LDA + 2 * cp);
    static constexpr size_t smem_fold = sizeof(float) * (size_t)(cp * LDA + 2 * cp);
};

// W_o^T fragments of this wavefront's fold tiles mt0 + wave + NW t (t < MTW): requested as early as possible, they do not depend
// on anything the kernel computes.  B-operand lane (k = l >> 4, m = l & 15).
template <int CT>
__device__ __forceinline__ void fold_prefetch(float (&bfr)[AC<CT>::MTW][AC<CT>::cp / 4], const float* __restrict__ WoT, long ldwt, int h,
                                              int mt0, int mt1) {
    constexpr int cp = AC<CT>::cp, c = AC<CT>::c, NW = AC<CT>::NW, MTW = AC<CT>::MTW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int mt = mt0 + wave + NW * t;
        const float* Wl = WoT + ((long)h * c + kq) * ldwt + 16 * mt + r16;
#pragma unroll
        for (int s = 0; s < cp / 4; ++s) bfr[t][s] = (mt < mt1 && 4 * s + kq < c) ? Wl[(long)(4 * s) * ldwt] : 0.f;
    }
}

// Gs [cp][LDA]: raw Gram block (rows/columns >= c arbitrary), sqs [2 cp]: |q_i|^2 at i, |k_j|^2 at cp + j.  On return Gs holds A
// (zero padded); Gn and A are written when `small_out`; the fold covers the 16-column tiles [mt0, mt1) of MfT.
// Softmax: a row per 16-lane group (CT columns per lane), so the shuffle chains of 16 (32) rows run side by side.
template <int CT>
__device__ __forceinline__ void attn_tail(float* Gs, const float* sqs, const float (&bfr)[AC<CT>::MTW][AC<CT>::cp / 4], int h, int b,
                                          int heads, const float* __restrict__ temp, float* __restrict__ Gn,
                                          float* __restrict__ A, float* __restrict__ MfT, long ldm, long sMb, int mt0, int mt1,
                                          bool small_out) {
    constexpr int cp = AC<CT>::cp, c = AC<CT>::c, LDA = AC<CT>::LDA, NW = AC<CT>::NW, NT = AC<CT>::NT, MTW = AC<CT>::MTW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef AC_NO_TAIL
    if (sMb != -12345) return;
#endif
    const long off = ((long)b * heads + h) * c * c;
    const float tau = temp[h];
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    float kn[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) kn[t] = gl + 16 * t < c ? clamp_norm(sqs[cp + gl + 16 * t]) : 1.f;
#pragma unroll 1
    for (int i = grp; i < cp; i += NT / 16) {
        float* row = Gs + i * LDA;
        const float nq = clamp_norm(sqs[i]);
        float g[CT], e[CT], mx = -INFINITY, sum = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const bool ok = i < c && gl + 16 * t < c;
            g[t] = ok ? row[gl + 16 * t] / (nq * kn[t]) : 0.f;
            mx = fmaxf(mx, ok ? g[t] * tau : -INFINITY);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            e[t] = (i < c && gl + 16 * t < c) ? expf(g[t] * tau - mx) : 0.f;
            sum += e[t];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = i < c ? 1.0f / sum : 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int j = gl + 16 * t;
            row[j] = e[t] * inv;                                           // zero in the padding rows / columns
            if (small_out && i < c && j < c) {
                Gn[off + i * c + j] = g[t];
                A[off + i * c + j] = e[t] * inv;
            }
        }
    }
    __syncthreads();
    // MfT[h c + j][m] = sum_i A[i][j] W_o^T[h c + i][m]  as 16x16x4 fp32 MFMAs: D[j][m], A-operand lane (j = l & 15, k = l >> 4)
    const int r16 = lane & 15, kq = lane >> 4;
    float* Mh = MfT + (long)b * sMb + (long)h * c * ldm;
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int mt = mt0 + wave + NW * t;
        if (mt >= mt1) break;                                              // wave-uniform
#pragma unroll 1
        for (int jt = 0; jt < CT; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* Ga = Gs + kq * LDA + 16 * jt + r16;
#pragma unroll
            for (int s = 0; s < cp / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ga[4 * s * LDA], bfr[t][s], acc, 0, 0, 0);
            float* Md = Mh + (long)(16 * jt + 4 * kq) * ldm + 16 * mt + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * jt + 4 * kq + r < c) Md[(long)r * ldm] = acc[r];
        }
    }
}

// PARTIAL (gpart != null): grid (S pixel ranges, heads, B), gpart [z][S][c][c], sqpart [z][S][2c], z = b heads + h.
// FUSED (gpart == null): grid (MS column slices, heads, B): every workgroup takes the statistics of the whole image (pxw = N)
// and goes on to softmax + fold of its slice; slice 0 writes sq, Gn, A.
template <int CT>
__global__ __launch_bounds__(AC<CT>::NT) void qk_stats_kernel(const float* __restrict__ u, long sUb, int heads, int N, int pxw,
                                                              float* __restrict__ gpart, float* __restrict__ sqpart,
                                                              const float* __restrict__ temp, const float* __restrict__ WoT, long ldwt,
                                                              float* __restrict__ sq, float* __restrict__ Gn, float* __restrict__ A,
                                                              float* __restrict__ MfT, long ldm, long sMb) {
    using K = AC<CT>;
    constexpr int cp = K::cp, c = K::c, LDA = K::LDA, PX = K::PX, LDQ = K::LDQ, NW = K::NW, NT = K::NT, Q4 = K::Q4, RPP = K::RPP, KQ = K::KQ;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* T = sm;
    float* Gs = T + K::TILE;
    float* sqs = Gs + cp * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const bool fused = gpart == nullptr;
    const int s = fused ? 0 : blockIdx.x, S = fused ? 1 : gridDim.x;
    const int C = heads * c;
    const int mt0 = fused ? blockIdx.x * K::MT_WG : 0, mt1 = fused ? min(C / 16, mt0 + K::MT_WG) : 0;
    float bfr[K::MTW][cp / 4];
    if (fused) fold_prefetch<CT>(bfr, WoT, ldwt, h, mt0, mt1);
    // load geometry: thread = (row rsub of a pass, pixel quad c4); pass k covers rows rsub + RPP k of q, then of k
    const int rsub = tid / Q4, c4 = tid % Q4;
    const float* qsrc = u + (long)b * sUb + ((long)h * c + rsub) * N + 4 * c4 + s * pxw;
    const float* ksrc = qsrc + (long)C * N;
    float* tdst = T + rsub * LDQ + 4 * c4;
    for (int i = tid; i < 2 * cp; i += NT) sqs[i] = 0.f;
    if (c < cp) {                                                        // rows c..cp-1 of both halves stay zero
#pragma unroll 1
        for (int r = c + wave; r < cp; r += NW)
            for (int x = lane; x < LDQ; x += 64) T[r * LDQ + x] = T[(cp + r) * LDQ + x] = 0.f;
    }
    f32x4 acc[CT][CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll 1
    for (int p0 = 0; p0 < pxw; p0 += PX) {
        f32x4 v[2 * KQ];                                                 // the whole chunk in flight at once
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const bool ok = RPP * k < c;                             // c % RPP == 0: the same for every row of the pass
#ifndef AC_NO_LOAD
            v[k] = ok ? *reinterpret_cast<const f32x4*>(qsrc + (long)(RPP * k) * N + p0) : f32x4{0.f, 0.f, 0.f, 0.f};
            v[KQ + k] = ok ? *reinterpret_cast<const f32x4*>(ksrc + (long)(RPP * k) * N + p0) : f32x4{0.f, 0.f, 0.f, 0.f};
#else
            v[k] = v[KQ + k] = f32x4{(float)ok, 0.f, 0.f, 0.f};
#endif
        }
        __syncthreads();                                                 // the previous chunk has been consumed
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            if (RPP * k < c) {
                *reinterpret_cast<f32x4*>(tdst + (RPP * k) * LDQ) = v[k];
                *reinterpret_cast<f32x4*>(tdst + (cp + RPP * k) * LDQ) = v[KQ + k];
            }
        }
        __syncthreads();
#ifndef AC_NO_SUMSQ
        // row sums of squares: four lanes per row, interleaved pixel quads
#pragma unroll 1
        for (int r = tid >> 2; r < 2 * cp; r += NT / 4) {
            const float* row = T + r * LDQ + 4 * (tid & 3);
            float a = 0.f;
#pragma unroll 4
            for (int t = 0; t < Q4 / 4; ++t) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(row + 16 * t);
                a += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
            }
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            if ((tid & 3) == 0) sqs[r] += a;
        }
#endif
#ifndef AC_NO_MFMA
        // Gram block: wavefront w takes the 16-pixel groups w, w + NW, ...; lane (r16, kq) reads pixels 16 g + 4 kq .. + 3 of its row
#pragma unroll 1
        for (int g = wave; g < PX / 16; g += NW) {
            f32x4 af[CT], bf[CT];
            const float* Tl = T + r16 * LDQ + 16 * g + 4 * kq;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                af[t] = *reinterpret_cast<const f32x4*>(Tl + (16 * t) * LDQ);
                bf[t] = *reinterpret_cast<const f32x4*>(Tl + (cp + 16 * t) * LDQ);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < CT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
#endif
    }
    // the wavefronts leave their blocks in the (now free) pixel tile, RW at a time; each round is added into Gs in a fixed order
    constexpr int RW = K::RW;
#pragma unroll 1
    for (int rd = 0; rd < NW / RW; ++rd) {
        __syncthreads();
        if (wave / RW == rd) {
            float* Wv = T + (wave % RW) * cp * cp + (4 * kq) * cp + r16;
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Wv[(16 * i + r) * cp + 16 * j] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll 1
        for (int i = wave; i < c; i += NW)
            for (int j = lane; j < c; j += 64) {
                float a = rd == 0 ? 0.f : Gs[i * LDA + j];
#pragma unroll
                for (int w = 0; w < RW; ++w) a += T[w * cp * cp + i * cp + j];
                Gs[i * LDA + j] = a;
            }
    }
    const long z = (long)b * heads + h;
    if (!fused) {
        float* gp = gpart + (z * S + s) * c * c;
#pragma unroll 1
        for (int i = wave; i < c; i += NW)                                  // (each thread re-reads what it wrote itself)
            for (int j = lane; j < c; j += 64) gp[i * c + j] = Gs[i * LDA + j];
        __syncthreads();
        float* sp = sqpart + (z * S + s) * 2 * c;
        for (int r = tid; r < 2 * c; r += NT) sp[r] = sqs[r < c ? r : cp + r - c];
        return;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int r = tid; r < 2 * c; r += NT) sq[(long)b * 2 * C + (r < c ? h * c + r : C + h * c + r - c)] = sqs[r < c ? r : cp + r - c];
    attn_tail<CT>(Gs, sqs, bfr, h, b, heads, temp, Gn, A, MfT, ldm, sMb, mt0, mt1, blockIdx.x == 0);
}

// grid (MS column slices, heads, B): sums the S partial Gram blocks / sums of squares, then softmax + fold of its slice.
template <int CT>
__global__ __launch_bounds__(AC<CT>::NT) void softmax_fold_kernel(const float* __restrict__ gpart, const float* __restrict__ sqpart, int S,
                                                                  int heads, const float* __restrict__ temp,
                                                                  const float* __restrict__ WoT, long ldwt, float* __restrict__ sq,
                                                                  float* __restrict__ Gn, float* __restrict__ A, float* __restrict__ MfT,
                                                                  long ldm, long sMb) {
    using K = AC<CT>;
    constexpr int cp = K::cp, c = K::c, LDA = K::LDA, NT = K::NT, NW = K::NW, NR = cp / NW, NJ = (cp + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Gs = sm;
    float* sqs = Gs + cp * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ms = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int C = heads * c;
    const int mt0 = ms * K::MT_WG, mt1 = min(C / 16, mt0 + K::MT_WG);
    float bfr[K::MTW][cp / 4];
    fold_prefetch<CT>(bfr, WoT, ldwt, h, mt0, mt1);
    const long z = (long)b * heads + h;
    const int cc = c * c;
    const float* gp = gpart + z * S * cc;
    // element (i = wave + NW r, j = lane + 64 q): NR * NJ sums per thread, the loads of two partial blocks in flight
    float a[NR][NJ];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < NJ; ++q) a[r][q] = 0.f;
#pragma unroll 2
    for (int s = 0; s < S; ++s) {                                         // fixed order
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const int i = wave + NW * r, j = lane + 64 * q;
                a[r][q] += (i < c && j < c) ? gp[(long)s * cc + i * c + j] : 0.f;
            }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int i = wave + NW * r, j = lane + 64 * q;
            if (i < c && j < c) Gs[i * LDA + j] = a[r][q];
        }
    const float* sp = sqpart + z * S * 2 * c;
    for (int r = tid; r < 2 * c; r += NT) {
        float t = sp[r];
        for (int s = 1; s < S; ++s) t += sp[(long)s * 2 * c + r];
        sqs[r < c ? r : cp + r - c] = t;
        if (ms == 0) sq[(long)b * 2 * C + (r < c ? h * c + r : C + h * c + r - c)] = t;
    }
    __syncthreads();
    attn_tail<CT>(Gs, sqs, bfr, h, b, heads, temp, Gn, A, MfT, ldm, sMb, mt0, mt1, ms == 0);
}

template <int CT>
int launch_core_fwd(const float* u, long sUb, const float* temp, const float* WoT, long ldwt, float* sq, float* Gn, float* A, float* MfT,
                    long ldm, long sMb, int B, int heads, int c, int N, float* ws, size_t ws_bytes, hipStream_t st) {
    using K = AC<CT>;
    const int C = heads * c;
    static bool once = (hipFuncSetAttribute((const void*)qk_stats_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) ==
                        hipSuccess);
    (void)once;
    const int ms = cdiv(C / 16, K::MT_WG);                         // column slices of the fold
    if (N == 256 && CT != 6) {                                     // one launch: every slice workgroup takes the statistics itself
        RCOT_LAUNCH(qk_stats_kernel<CT>, dim3(ms, heads, B), dim3(K::NT), K::smem_stats, st, u, sUb, heads, N, N, nullptr,
                           nullptr, temp, WoT, ldwt, sq, Gn, A, MfT, ldm, sMb);
        RCOT_LAUNCH_CHECK();
        return RCOT_OK;
    }
    // pixel ranges: 256 pixels per workgroup up to 32x32, 512 on the 64x64 level (fewer partial blocks to add); c = 96 at
    // N = 256 (the fold of c = 96 is too long to repeat per slice): two half ranges
    const int pxw = N == 256 ? 128 : (N > 16384 && N % 2048 == 0) ? 2048 : ((N > 1024 && N % 512 == 0) ? 512 : 256);
    const int S = N / pxw;
    const size_t Z = (size_t)B * heads;
    const size_t need = Z * S * ((size_t)c * c + 2 * c) * sizeof(float);
    if (!ws || ws_bytes < need) return RCOT_EWORKSPACE;
    float* gpart = ws;
    float* sqpart = ws + Z * S * c * c;
    RCOT_LAUNCH(qk_stats_kernel<CT>, dim3(S, heads, B), dim3(K::NT), K::smem_stats, st, u, sUb, heads, N, pxw, gpart, sqpart,
                       temp, WoT, ldwt, sq, Gn, A, MfT, ldm, sMb);
    RCOT_LAUNCH_CHECK();
    RCOT_LAUNCH(softmax_fold_kernel<CT>, dim3(ms, heads, B), dim3(K::NT), K::smem_fold, st, gpart, sqpart, S, heads, temp, WoT,
                       ldwt, sq, Gn, A, MfT, ldm, sMb);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the attention-matrix chain of one block, ONE launch, one workgroup per (head, image):
//     D = dM[b][:, head block] (C x c, = dY V^T; given dense or as S <= 8 split-K slabs that are summed while staging)
//     Mf[b][:, head block]      = W A            W = W_o[:, head block] (C x c)      (K-major operand of dV = Mf^T dY)
//     dW_o part[b][:, head blk] = D A^T
//     dA                        = W^T D          (c x c, summed over all C rows in registers)
//     dS = A.(dA - rowsum(dA.A));  dtau part = sum dS.Gn;  Eq = tau dS / (|q| |k|^T), Eq^T;  Dq, Dk
// replaces attn_bwd_chunk_kernel (one workgroup per row chunk, scalar FMAs from LDS: 17-33 us) + attn_bwd_small_kernel
// (10-50 us) and, with slabs, the reduce launch of dM.  The three products run on v_mfma_f32_16x16x4_f32 (exact fp32): a
// wavefront walks 16-row tiles of W and D staged in its own LDS slice and keeps its share of dA in registers.
template <int CT>
struct AB {
    static constexpr int cp = 16 * CT;
    static constexpr int c = CT == 2 ? 24 : cp;
    static constexpr int LDA = (cp % 32 == 16) ? cp : cp + 16;      // A / A^T as B operands: == 16 (mod 32)
    static constexpr int LD = cp + 2;                               // 16-row W / D tiles as A operands: LD/2 odd
    static constexpr int NW = 4, NT = 256;
    static constexpr int Q4 = c / 4;                                // 16-byte pieces per tile row
    static constexpr int NLD = (16 * Q4 + 63) / 64;                 // pieces per lane and tile
    static_assert(NW * 2 * 16 * LD >= cp * LDA, "Gn is staged in the tile region for the tail");
    static constexpr size_t smem = sizeof(float) * (size_t)(2 * cp * LDA + NW * 2 * 16 * LD + 2 * cp + 8);
};

template <int CT>
__global__ __launch_bounds__(256) void attn_bwd_core_kernel(const float* __restrict__ dM, int S, long ldd, long sDs, long sDb,
                                                            const float* __restrict__ Wo, const float* __restrict__ A,
                                                            const float* __restrict__ Gn, const float* __restrict__ sq,
                                                            const float* __restrict__ temp, float* __restrict__ Mf,
                                                            float* __restrict__ dWo_part, float* __restrict__ dtemp_part,
                                                            float* __restrict__ Eq, float* __restrict__ EqT, float* __restrict__ Dq,
                                                            float* __restrict__ Dk, int heads) {
    using K = AB<CT>;
    constexpr int cp = K::cp, c = K::c, LDA = K::LDA, LD = K::LD, NW = K::NW, NT = K::NT, Q4 = K::Q4, NLD = K::NLD;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;                               // A [i][j]
    float* ATs = As + cp * LDA;                   // A^T [j][i]; after the row loop: the dA accumulator
    float* tiles = ATs + cp * LDA;                // per wavefront: W tile [16][LD], D tile [16][LD]
    float* rsum = tiles + NW * 2 * 16 * LD;       // [cp] row sums, [cp] column sums of dS.Gn
    float* red = rsum + 2 * cp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int C = heads * c;
    const long off = ((long)b * heads + h) * c * c;
    // ---- A and A^T, zero padded; Gn rides along in registers until the tail.  ALL loads of a thread are issued before the first
    // store (a loop that loads and stores element by element chains one memory latency per trip: 9-36 of them here)
    constexpr int NEL = (cp * cp + NT - 1) / NT;
    float av[NEL], gv[NEL];
#pragma unroll
    for (int q = 0; q < NEL; ++q) {
        const int e = tid + q * NT;
        const int i = e / cp, j = e - i * cp;
        const bool ok = e < cp * cp && i < c && j < c;
        av[q] = ok ? A[off + i * c + j] : 0.f;
        gv[q] = ok ? Gn[off + i * c + j] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NEL; ++q) {
        const int e = tid + q * NT;
        const int i = e / cp, j = e - i * cp;
        if (e < cp * cp) {
            As[i * LDA + j] = av[q];
            ATs[j * LDA + i] = av[q];
        }
    }
    __syncthreads();
    const int r16 = lane & 15, kq = lane >> 4;
    float* Wt = tiles + wave * 2 * 16 * LD;
    float* Dt = Wt + 16 * LD;
    const float* Wh = Wo + h * c;                                       // W[m][i] = Wh[m * C + i]
    const float* Dh = dM + (long)b * sDb + h * c;                       // D[m][j] = sum_s Dh[s * sDs + m * ldd + j]
    float* Mfh = Mf + (long)b * C * C + h * c;
    float* dWh = dWo_part + (long)b * C * C + h * c;
    f32x4 dacc[CT][CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) dacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nmt = C / 16;
    // the global loads of tile mt + NW are issued before the products of tile mt (a wavefront walks its tiles alone: nothing
    // else would cover their latency)
    f32x4 wv[NLD], dv[NLD];
    auto fetch = [&](int mt) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = lane + 64 * q;
            const int r = idx / Q4, c4 = idx - r * Q4;
            const bool ok = idx < 16 * Q4 && mt < nmt;
            const long m = 16 * mt + r;
            wv[q] = ok ? *reinterpret_cast<const f32x4*>(Wh + m * C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 d = ok ? *reinterpret_cast<const f32x4*>(Dh + m * ldd + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            for (int sl = 1; sl < S; ++sl)
                if (ok) d += *reinterpret_cast<const f32x4*>(Dh + (long)sl * sDs + m * ldd + 4 * c4);
            dv[q] = d;
        }
    };
    fetch(wave);
#pragma unroll 1
    for (int mt = wave; mt < nmt; mt += NW) {
        // stage this wavefront's 16 rows of W and D (the slabs of D summed in a fixed order); 8-byte LDS stores (LD is even)
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = lane + 64 * q;
            const int r = idx / Q4, c4 = idx - r * Q4;
            if (idx < 16 * Q4) {
                float* wd = Wt + r * LD + 4 * c4;
                float* dd = Dt + r * LD + 4 * c4;
                *reinterpret_cast<float2*>(wd) = make_float2(wv[q][0], wv[q][1]);
                *reinterpret_cast<float2*>(wd + 2) = make_float2(wv[q][2], wv[q][3]);
                *reinterpret_cast<float2*>(dd) = make_float2(dv[q][0], dv[q][1]);
                *reinterpret_cast<float2*>(dd + 2) = make_float2(dv[q][2], dv[q][3]);
            }
        }
        fetch(mt + NW);
        if (c < cp) {                                                        // padding columns of the tiles (c = 24)
            for (int e = lane; e < 16 * (cp - c); e += 64) {
                const int r = e / (cp - c), x = c + e - r * (cp - c);
                Wt[r * LD + x] = 0.f;
                Dt[r * LD + x] = 0.f;
            }
        }
        // (own LDS slice, own lanes: wavefront-level ordering only — the compiler's lgkmcnt covers the store -> load order)
        __builtin_amdgcn_wave_barrier();
        // Mf tile = W A and dW_o tile = D A^T: D[m][n], A operand lane (m = r16, k = 4 s + kq), B operand lane (k, n = r16)
#pragma unroll 1
        for (int nt = 0; nt < CT; ++nt) {
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s_ = 0; s_ < cp / 4; ++s_) {
                const int k = 4 * s_ + kq;
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wt[r16 * LD + k], As[k * LDA + 16 * nt + r16], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Dt[r16 * LD + k], ATs[k * LDA + 16 * nt + r16], a2, 0, 0, 0);
            }
            const int n = 16 * nt + r16;
            if (n < c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long m = 16 * mt + 4 * kq + r;
                    Mfh[m * C + n] = a1[r];
                    dWh[m * C + n] = a2[r];
                }
            }
        }
        // dA += W^T D over the 16 rows: D[i][j], A operand lane (i = 16 it + r16, k = m = 4 s + kq), B operand lane (m, j)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            float wa[CT], db[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                wa[t] = Wt[(4 * s_ + kq) * LD + 16 * t + r16];
                db[t] = Dt[(4 * s_ + kq) * LD + 16 * t + r16];
            }
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) dacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], db[j], dacc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- dA: the wavefronts add their shares into the A^T region one after the other (fixed order)
    float* dAs = ATs;
#pragma unroll 1
    for (int w = 0; w < NW; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = dAs + (16 * i + 4 * kq + r) * LDA + 16 * j + r16;
                        *d = w == 0 ? dacc[i][j][r] : *d + dacc[i][j][r];
                    }
        }
    }
    __syncthreads();
    // ---- the c x c tail: a row per 16-lane group, CT columns per lane.  Gn goes to the (now free) tile region first.
    float* Gs = tiles;
#pragma unroll
    for (int q = 0; q < NEL; ++q) {
        const int e = tid + q * NT;
        const int i = e / cp, j = e - i * cp;
        if (e < cp * cp) Gs[i * LDA + j] = gv[q];
    }
    __syncthreads();
    const float tau = temp[h];
    const float* sqq = sq + (long)b * 2 * C + h * c;
    const float* sqk = sqq + C;
    const int grp = tid >> 4, gl = tid & 15;
    float kn[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) kn[t] = gl + 16 * t < c ? clamp_norm(sqk[gl + 16 * t]) : 1.f;
    float part = 0.f;
#pragma unroll 1
    for (int i = grp; i < cp; i += NT / 16) {
        float pv[CT], dvv[CT], acc = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int j = gl + 16 * t;
            pv[t] = As[i * LDA + j];                                         // zero in the padding
            dvv[t] = (i < c && j < c) ? dAs[i * LDA + j] : 0.f;
            acc += pv[t] * dvv[t];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        const float nq = i < c ? clamp_norm(sqq[i]) : 1.f;
        float rs_ = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int j = gl + 16 * t;
            const bool ok = i < c && j < c;
            const float sv = pv[t] * (dvv[t] - acc);
            const float sg = ok ? sv * Gs[i * LDA + j] : 0.f;
            rs_ += sg;
            const float e = tau * sv / (nq * kn[t]);
            dAs[i * LDA + j] = ok ? sg : 0.f;                               // dS.Gn (for the column sums)
            As[i * LDA + j] = ok ? e : 0.f;                                 // Eq (for the transposed copy)
            if (ok) Eq[off + i * c + j] = e;
        }
        part += rs_;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) rs_ += __shfl_xor(rs_, o, 64);
        if (gl == 0 && i < c) {
            const float q2 = sqq[i];
            Dq[(long)b * C + h * c + i] = q2 >= 1e-24f ? -tau * rs_ / q2 : 0.f;
        }
    }
    part = block_sum<256>(part, red);                                       // (its barriers also publish As / dAs)
    if (tid == 0) dtemp_part[(long)b * heads + h] = part;
#pragma unroll 1
    for (int i = grp; i < c; i += NT / 16) {                                // row i of Eq^T and column sum i of dS.Gn
        float cs_ = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int j = gl + 16 * t;
            if (j < c) {
                cs_ += dAs[j * LDA + i];
                EqT[off + (long)i * c + j] = As[j * LDA + i];
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) cs_ += __shfl_xor(cs_, o, 64);
        if (gl == 0) {
            const float k2 = sqk[i];
            Dk[(long)b * C + h * c + i] = k2 >= 1e-24f ? -tau * cs_ / k2 : 0.f;
        }
    }
}

template <int CT>
int launch_core_bwd(const float* dM, int S, long ldd, long sDs, long sDb, const float* Wo, const float* A, const float* Gn,
                    const float* sq, const float* temp, float* Mf, float* dWo_part, float* dtemp_part, float* Eq, float* EqT, float* Dq,
                    float* Dk, int B, int heads, hipStream_t st) {
    using K = AB<CT>;
    static bool once = (hipFuncSetAttribute((const void*)attn_bwd_core_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024) == hipSuccess);
    (void)once;
    RCOT_LAUNCH(attn_bwd_core_kernel<CT>, dim3(heads, B), dim3(K::NT), K::smem, st, dM, S < 1 ? 1 : S, ldd, sDs, sDb, Wo, A, Gn, sq,
                       temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk, heads);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // namespace

extern "C" {

int rcot_attn_core_fwd(const float* u, long sUb, const float* temp, const float* WoT, long ldwt, float* sq, float* Gn, float* A,
                       float* MfT, long ldm, long sMb, int B, int heads, int c, int N, float* ws, size_t ws_bytes, void* stream) {
    if (!u || !temp || !WoT || !sq || !Gn || !A || !MfT || B <= 0 || heads <= 0 || c <= 0 || N <= 0 || B > 65535 || heads > 65535)
        return RCOT_EINVAL;
    const int C = heads * c;
    if ((reinterpret_cast<uintptr_t>(u) & 15) || (sUb & 3) || ldwt < C || ldm < C) return RCOT_EINVAL;
    // Planes above 64x64: the kernels take them (round 5: N = 16384 as 32 pixel ranges of 512 per head and image, 256x256 patches as
    // ranges of 2048; tests/test_kernels_gpu.py), two launches instead of four — and MEASURED SLOWER there than rcot_row_sumsq +
    // rcot_bmm_nt_slabs + rcot_attn_softmax + rcot_bmm_nn: a 128x128 block forward 545 -> 579 us (fp32), 447 -> 470 us (bf16x3),
    // profiles/r05_ab_small_levels.txt: qk_stats_kernel loads a chunk, stores it to LDS and multiplies it one after the other, which the
    // LDS-DMA Gram kernel overlaps.  The host wrapper (HipBackend.attn_core_fwd, attn_core_maxn) therefore stops at 4096 pixels.
    if ((C % 16) || (N % 256) || N > 65536) return RCOT_EUNSUPPORTED;
    if (N > 4096 && (N % 512)) return RCOT_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (c == 48) return launch_core_fwd<3>(u, sUb, temp, WoT, ldwt, sq, Gn, A, MfT, ldm, sMb, B, heads, c, N, ws, ws_bytes, st);
    if (c == 96) return launch_core_fwd<6>(u, sUb, temp, WoT, ldwt, sq, Gn, A, MfT, ldm, sMb, B, heads, c, N, ws, ws_bytes, st);
    if (c == 24) return launch_core_fwd<2>(u, sUb, temp, WoT, ldwt, sq, Gn, A, MfT, ldm, sMb, B, heads, c, N, ws, ws_bytes, st);
    return RCOT_EUNSUPPORTED;
}

int rcot_attn_core_bwd(const float* dM, int S, int ldd, const float* Wo, const float* A, const float* Gn, const float* sq,
                       const float* temp, float* Mf, float* dWo_part, float* dtemp_part, float* Eq, float* EqT, float* Dq, float* Dk, int B,
                       int heads, int c, void* stream) {
    if (!dM || !Wo || !A || !Gn || !sq || !temp || !Mf || !dWo_part || !dtemp_part || !Eq || !EqT || !Dq || !Dk || B <= 0 ||
        heads <= 0 || B > 65535 || heads > 65535 || S < 0 || S > 8)
        return RCOT_EINVAL;
    const int C = heads * c;
    if ((C % 16) || (c != 24 && c != 48 && c != 96)) return RCOT_EUNSUPPORTED;
    if (S == 0) ldd = C;
    if (ldd < C || (ldd & 3) || (reinterpret_cast<uintptr_t>(dM) & 15) || (reinterpret_cast<uintptr_t>(Wo) & 15)) return RCOT_EINVAL;
    const long sDs = (long)C * ldd, sDb = (long)(S < 1 ? 1 : S) * sDs;
    hipStream_t st = (hipStream_t)stream;
    if (c == 48) return launch_core_bwd<3>(dM, S, ldd, sDs, sDb, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk, B, heads, st);
    if (c == 96) return launch_core_bwd<6>(dM, S, ldd, sDs, sDb, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk, B, heads, st);
    return launch_core_bwd<2>(dM, S, ldd, sDs, sDb, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk, B, heads, st);
}

}  // extern "C"
