// Second-generation fp32 MFMA GEMM for the hot "projection" products of the transport map:
//
//     C[z] (M x N) = A[z] (M x K) * B[z] (K x N)      with M, K <= a few hundred and N = pixels (huge)
//
// Both operands are consumed K-MAJOR ( At[k][m] with m contiguous, B[k][n] with n contiguous ) so that every
// 16-row K-slab of either operand is a set of full 512-byte rows.  Slabs are moved HBM/L2 -> LDS by the
// LDS-DMA path (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a 4-stage ring; three slabs are in
// flight while one is being multiplied, tracked with counted s_waitcnt vmcnt and ONE raw s_barrier per slab.
// The LDS image is exactly the lane-linear DMA image ( [16][128] floats per operand and stage ), which is also
// conflict-free for the 32x32x2 operand fetch (a half-wave reads 32 consecutive floats of one k-row).
// The WithBias-LayerNorm prologue is applied when a B fragment is read from LDS (per-lane mu/rstd in registers,
// per-channel weight/bias in LDS), so normalised activations never exist in memory.
//
// Contract (checked by the entry point, otherwise RCOT_EINVAL and the caller uses the general engine):
//   N % 128 == 0; lda, ldb, batch strides % 4 == 0; 16-byte aligned bases;
//   At must be readable for rows [0, ceil16(K)) and hold ZEROS in rows >= K (weights come from the padded
//   pack made by rcot_pack_weight; per-image matrices have K % 16 == 0).
//   Columns m >= M of At may hold anything (rows of C are independent; they are never stored).
#include <cstdlib>
#include "gemm_core.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int XX_NST = 3;             // LDS ring stages of the 96- / 128-row tiles (16 KiB each; several workgroups per CU hide the rest)
// 64 x 64 tiles (the 32x32 / 16x16 planes): ring depth as a build parameter.  MEASURED (round 5, scripts/ab_variant.sh, profiles/r05_ab_ring.txt,
// r05_ab_dual.txt, r05_lds_mfma_modes.txt): six stages instead of three change no product (384 <- 2042 at 8 x 16x16: 51.1 vs 50.2 us) and cost
// 0.6 ms per iteration — these kernels do not wait for memory; two alternating accumulators per wavefront (no MFMA behind the one it
// depends on) change nothing either (45.8 vs 45.6 us, 77.6 vs 77.6 ms per iteration: removed again).  The LDS -> MFMA loop of this tile
// shape ALONE runs at 100 TF/s at one workgroup per CU, whatever the read schedule, with or without the barrier (128 TF/s for four 32 x 32
// tiles per wavefront, 125 TF/s at two workgroups per CU): one tile per wavefront needs two ds_read_b32 per MFMA, four tiles one.
#ifndef XX_NST_SMALL
#define XX_NST_SMALL 3
#endif
template <int BM, int BN> constexpr int xx_nst() { return (BM <= 64 && BN <= 64) ? XX_NST_SMALL : XX_NST; }

// wait until at most y slabs (OPS vm operations each) of this wave are outstanding, 0 <= y <= YMAX (wave-uniform y)
template <int OPS, int YMAX> __device__ __forceinline__ void wait_vm_slabs(int y) {
    static_assert(OPS * YMAX <= 63, "vmcnt is a 6-bit field");
    if constexpr (YMAX == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else {
        if (y >= YMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS * YMAX) : "memory");
        else wait_vm_slabs<OPS, YMAX - 1>(y);
    }
}

struct XXP {
    int M, N, K, Zi, tilesM, tilesN;
    const float* At; long lda, sAo, sAi;
    const float* B;  long ldb, sBo, sBi;
    const float* mu; const float* rs; long sLN;
    const float* lnw; const float* lnb;
    // ln_comp: (mu, rs) are OUTPUTS — every workgroup makes the statistics of its own BN pixel columns before its slab loop (the
    // formula and the summation order of ln_stats_kernel, pointwise.hip: bit-identical values) and row tile 0 writes them out for the
    // backward pass.  The B panel [K][BN] it walks for that is the one its slab loop reads anyway (L2-resident for the other row
    // tiles of the same columns): no rcot_ln_stats launch in front of an exact-fp32 LayerNorm projection.
    float* mu_out; float* rs_out; int ln_comp;
    // EPI == 2 (xx_body): the WithBias-LayerNorm statistics of the OUTPUT (over its M <= 96 rows = channels, per pixel) are made by the
    // epilogue and written to st_mu / st_rs [Zo][N] (batch stride sST): the LayerNorm that follows needs no pass over the tensor
    float* st_mu; float* st_rs; long sST;
    EpiP ep;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// epilogue_vec (gemm_core.h) for a tile that holds EVERY row (channel) of its pixels (one row tile: M <= BM, 1 x 4 wavefronts: each
// wavefront all rows of 32 pixels), plus the per-pixel WithBias-LayerNorm statistics of the values it stores (Net_Restormer.py:186-189:
// mean and biased variance over the channels) — what rcot_ln_stats would compute from the stored tensor in a pass of its own.
// ln_stats_kernel's formula (sums shifted by the pixel's channel-0 value, var = max(E[d^2] - E[d]^2, 0), rs = 1/sqrt(var + 1e-5)); the
// summation order differs (a lane's twelve rows, then the eight row groups of the wavefront): equal to fp32 rounding, not to the bit.
template <int TM>
__device__ __forceinline__ void epilogue_vec_stats(f32x16 (&acc)[TM][1], float* scr, float* Cb, long ldc, const float* Rb, long ldr,
                                                   const float* Sb, float alpha, float beta, int nbase, int M, int lane, bool nts,
                                                   float* mu_out, float* rs_out) {
    const int lm = lane & 31, lk = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    const float* Ab = Rb ? Rb : (beta != 0.f ? Cb : nullptr);
    const long lda_ = Rb ? ldr : ldc;
    const float amul = Rb ? 1.f : beta;
    float4 q[TM][4];
    float sc[TM][4];
    const int n = nbase + c4;
    if (Ab) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int m = i * 32 + ps * 8 + rr;
                sc[i][ps] = (Rb && Sb && m < M) ? Sb[m] : amul;
                q[i][ps] = m < M ? *reinterpret_cast<const float4*>(Ab + (long)m * lda_ + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    const bool both = Rb && beta != 0.f;
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), s = sh, ss = sh;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + lm] = acc[i][0][r];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + rr;
            const int m = i * 32 + row;
            float4 v = *reinterpret_cast<const float4*>(scr + row * 32 + c4);
            v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
            if (Ab) {
                const float4 a = q[i][ps];
                const float s_ = sc[i][ps];
                v.x += s_ * a.x; v.y += s_ * a.y; v.z += s_ * a.z; v.w += s_ * a.w;
            }
            float* dst = Cb + (long)m * ldc + n;
            if (both && m < M) {
                const float4 o = *reinterpret_cast<const float4*>(dst);
                v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
            }
            if (i == 0 && ps == 0) {
                // the shift: channel 0 of these four pixels, held by the lanes of row group 0 (lanes 0..7): lane (lane & 7) has it
                sh.x = __shfl(v.x, lane & 7, 64); sh.y = __shfl(v.y, lane & 7, 64); sh.z = __shfl(v.z, lane & 7, 64); sh.w = __shfl(v.w, lane & 7, 64);
            }
            if (m < M) {
                const float dx_ = v.x - sh.x, dy_ = v.y - sh.y, dz_ = v.z - sh.z, dw_ = v.w - sh.w;
                s.x += dx_; s.y += dy_; s.z += dz_; s.w += dw_;
                ss.x += dx_ * dx_; ss.y += dy_ * dy_; ss.z += dz_ * dz_; ss.w += dw_ * dw_;
                if (nts) {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    nt_f4 w_ = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(w_, reinterpret_cast<nt_f4*>(dst));
                } else {
                    *reinterpret_cast<float4*>(dst) = v;
                }
            }
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        ss.x += __shfl_xor(ss.x, o, 64); ss.y += __shfl_xor(ss.y, o, 64); ss.z += __shfl_xor(ss.z, o, 64); ss.w += __shfl_xor(ss.w, o, 64);
    }
    if (rr == 0) {
        const float inv = 1.0f / (float)M;
        float4 m4, r4;
#define RCOT_XX_ST_FIN(c)                                           \
    {                                                               \
        const float e = s.c * inv;                                  \
        const float var = fmaxf(ss.c * inv - e * e, 0.f);           \
        m4.c = sh.c + e;                                            \
        r4.c = 1.0f / sqrtf(var + 1e-5f);                           \
    }
        RCOT_XX_ST_FIN(x) RCOT_XX_ST_FIN(y) RCOT_XX_ST_FIN(z) RCOT_XX_ST_FIN(w)
#undef RCOT_XX_ST_FIN
        *reinterpret_cast<float4*>(mu_out + n) = m4;
        *reinterpret_cast<float4*>(rs_out + n) = r4;
    }
}

// BM x BN output tile; the A slab image is AW = 64 or 128 columns wide (BM = 96 rides in a 128-wide image), the B
// image BN (64 or 128) wide.  64x64 tiles keep the small-N levels (32x32 / 16x16 pixels per image) on this kernel.
template <int BM, int BN, int WM, int WN, bool LNP, int EPI = 0>
__device__ __forceinline__ void xx_body(const XXP& p, const int bx, const int bz) {
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int AW = BM <= 64 ? 64 : 128, BW = BN;
    constexpr int XX_STAGE = BK * (AW + BW);
    constexpr int NST_ = xx_nst<BM, BN>();
    constexpr int PA = AW / 64, PB = BW / 64;                         // 1-KiB DMA pieces per wave per slab
    static_assert(WM * WN == 4 && TM * 32 * WM == BM && TN * 32 * WN == BN && (BN == 64 || BN == 128), "tile/wave grid mismatch");
    extern __shared__ __attribute__((aligned(16))) float lds[];     // ring, then (LNP) lnw[Kp], lnb[Kp]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = p.tilesM * p.tilesN;
    const int bid = xcd_remap(bx, nblk);
    const int tm = bid % p.tilesM, tn = bid / p.tilesM;
    const int z = bz, zo = z / p.Zi, zi = z - zo * p.Zi;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = (p.K + BK - 1) / BK;

    // ---- DMA addressing: this wave issues ring pieces q = wave and wave+4 of A and of B (1 KiB = 2 k-rows each)
    const int arow_l = lane / (AW / 4), acol = (lane % (AW / 4)) * 4;  // a piece = 256/AW rows of the A image
    const int brow_l = lane / (BW / 4), bcol = (lane % (BW / 4)) * 4;
    int mcol = m0 + acol;
    if (mcol > (int)p.lda - 4) mcol = (int)p.lda - 4;               // stay inside the row (columns >= M are don't-care)
    const float* Ab = p.At + zo * p.sAo + zi * p.sAi + mcol;
    const float* Bb = p.B + zo * p.sBo + zi * p.sBi + n0 + bcol;

    if (LNP) {
        float* lw = lds + NST_ * XX_STAGE;
        const int Kp = nk * BK;
        for (int k = tid; k < Kp; k += GEMM_NT) {
            lw[k] = k < p.K ? p.lnw[k] : 0.f;
            lw[Kp + k] = k < p.K ? p.lnb[k] : 0.f;
        }
    }
    float mu_[TN], rs_[TN];
    if (LNP && p.ln_comp) {
        // 16 lanes x float4 cover 64 pixels, 16 thread rows stride over the channels; cross-row sums through the (still unused) ring
        float4* red = reinterpret_cast<float4*>(lds);
        float* smu = lds + 1024;
        float* srs = smu + BN;
        const int tx = tid & 15, ty = tid >> 4;
        const float inv = 1.0f / (float)p.K;
        const float* Bz = p.B + zo * p.sBo + zi * p.sBi;
#pragma unroll 1
        for (int h = 0; h < BN / 64; ++h) {
            const int n = n0 + h * 64 + tx * 4;
            const float* px = Bz + n;
            const float4 sh = *reinterpret_cast<const float4*>(px);     // shifted sums (shift = channel 0), as ln_stats_kernel
            float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0);
#pragma unroll 4
            for (int c = ty; c < p.K; c += 16) {
                const float4 v = *reinterpret_cast<const float4*>(px + (long)c * p.ldb);
                const float dx_ = v.x - sh.x, dy_ = v.y - sh.y, dz_ = v.z - sh.z, dw_ = v.w - sh.w;
                s.x += dx_; s.y += dy_; s.z += dz_; s.w += dw_;
                ss.x += dx_ * dx_; ss.y += dy_ * dy_; ss.z += dz_ * dz_; ss.w += dw_ * dw_;
            }
            __syncthreads();
            red[ty * 16 + tx] = s;
            __syncthreads();
            float4 t = red[tx];
#pragma unroll
            for (int i = 1; i < 16; ++i) { const float4 q = red[i * 16 + tx]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
            s = t;
            __syncthreads();
            red[ty * 16 + tx] = ss;
            __syncthreads();
            t = red[tx];
#pragma unroll
            for (int i = 1; i < 16; ++i) { const float4 q = red[i * 16 + tx]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
            ss = t;
            if (ty == 0) {
                float4 m, r;
#define RCOT_XX_LN_FIN(q)                                           \
    {                                                               \
        const float e = s.q * inv;                                  \
        const float var = fmaxf(ss.q * inv - e * e, 0.f);           \
        m.q = sh.q + e;                                             \
        r.q = 1.0f / sqrtf(var + 1e-5f);                            \
    }
                RCOT_XX_LN_FIN(x) RCOT_XX_LN_FIN(y) RCOT_XX_LN_FIN(z) RCOT_XX_LN_FIN(w)
#undef RCOT_XX_LN_FIN
                *reinterpret_cast<float4*>(smu + h * 64 + tx * 4) = m;
                *reinterpret_cast<float4*>(srs + h * 64 + tx * 4) = r;
                if (tm == 0) {
                    *reinterpret_cast<float4*>(p.mu_out + zo * p.sLN + n) = m;
                    *reinterpret_cast<float4*>(p.rs_out + zo * p.sLN + n) = r;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 32 + (lane & 31);
            mu_[j] = smu[nl];
            rs_[j] = srs[nl];
            asm volatile("" ::"v"(mu_[j]), "v"(rs_[j]));            // retire these reads before any DMA lands in the ring
        }
    } else if (LNP) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            mu_[j] = p.mu[zo * p.sLN + n];
            rs_[j] = p.rs[zo * p.sLN + n];
            asm volatile("" ::"v"(mu_[j]), "v"(rs_[j]));            // retire these ordinary loads before any DMA is issued
        }
    }
    __syncthreads();

    auto issue = [&](int kt) {
        float* st = lds + (kt % NST_) * XX_STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < PA; ++h) {
            const int q = wave + 4 * h;
            const int kr = k0 + (256 / AW) * q + arow_l;             // A rows < ceil16(K) are readable by contract
            __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (long)kr * p.lda), (lptr_t)(st + q * 256), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < PB; ++h) {
            const int q = wave + 4 * h;
            int kr = k0 + (256 / BW) * q + brow_l;
            if (kr >= p.K) kr = p.K - 1;                             // finite filler; the matching A rows are zero
            __builtin_amdgcn_global_load_lds((gptr_t)(Bb + (long)kr * p.ldb), (lptr_t)(st + BK * AW + q * 256), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: NST_ - 1 slabs in flight
#pragma unroll
    for (int i = 0; i < NST_ - 1; ++i)
        if (i < nk) issue(i);

    const int lm = lane & 31, lk = lane >> 5;
    const float* lw = lds + NST_ * XX_STAGE;
    const int Kp = nk * BK;
    for (int kt = 0; kt < nk; ++kt) {
        // slab kt has landed when at most the younger slabs of this wave (NST_ - 2 of them, fewer at the end) are outstanding
        wait_vm_slabs<PA + PB, NST_ - 2>(nk - 1 - kt);
        __builtin_amdgcn_s_barrier();          // every wave's pieces of slab kt are in LDS; slab kt-1 is no longer read
        if (kt + NST_ - 1 < nk) issue(kt + NST_ - 1);   // refill the stage that slab kt-1 occupied
        const float* As = lds + (kt % NST_) * XX_STAGE;
        const float* Bs = As + BK * AW;
        // all fragment reads of the slab first (32..48 VGPRs), then the MFMAs back-to-back behind counted lgkmcnt waits
        float a[BK / 2][TM], b[BK / 2][TN];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[ks][i] = As[(2 * ks + lk) * AW + (wm * TM + i) * 32 + lm];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[ks][j] = Bs[(2 * ks + lk) * BW + (wn * TN + j) * 32 + lm];
        }
        if (LNP) {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const float w = lw[kt * BK + 2 * ks + lk], bb = lw[Kp + kt * BK + 2 * ks + lk];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[ks][j] = (b[ks][j] - mu_[j]) * rs_[j] * w + bb;
            }
        }
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: 16-byte row-contiguous stores through a per-wave LDS transpose (gemm_core.h)
    const EpiP& ep = p.ep;
    float* Cb = ep.C + zo * ep.sCo + zi * ep.sCi;
    const float* Rb = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
    const float* Sb = ep.rowscale ? ep.rowscale + zo * ep.sSo + zi * ep.sSi : nullptr;
    __syncthreads();                     // every wave is done with the ring
    if constexpr (EPI == 2) {
        static_assert(EPI != 2 || (WM == 1 && TN == 1), "the statistics epilogue needs every row of a pixel in one wavefront");
        epilogue_vec_stats<TM>(acc, lds + wave * 1024, Cb, ep.ldc, Rb, ep.ldr, Sb, ep.alpha, ep.beta, n0 + wn * 32, p.M, lane, ep.nts != 0,
                               p.st_mu + zo * p.sST, p.st_rs + zo * p.sST);
        return;
    }
    epilogue_vec<TM, TN>(acc, lds + wave * 1024, Cb, ep.ldc, Rb, ep.ldr, Sb, ep.alpha, ep.beta, m0 + wm * TM * 32,
                         n0 + wn * TN * 32, p.M, p.N, lane, ep.nts != 0);
}

// 64 x 64 tile on EIGHT wavefronts (round 5): two k-groups of 2 x 2 wavefronts; group g multiplies the slabs kt = g (mod 2) into its own
// accumulators, the groups' sums meet in LDS at the end.  The long reductions of the small planes (data gradients with K = 510 ... 2042
// on 8 x 256 or 8 x 1024 pixels) are 24-192 workgroups of ONE 32 x 32 tile per wavefront: one workgroup per CU, one wavefront per SIMD,
// and every slab's barrier + fragment reads are exposed (384 <- 2042 at 8 x 16x16: 49 us for 20 us of MFMA work).  Two wavefronts per SIMD
// hide one group's barrier and LDS latency behind the other's MFMAs without a split-K slab in memory.  One s_barrier per slab PAIR; ring
// of 2 KG_NPAIR 8-KiB stages (KG_NPAIR - 1 slab pairs in flight behind the one being multiplied); wave w moves piece (w & 3) of A and of B of slab
// 2 j + (w >> 2).  Summation order differs from gemm_xx_kernel's (two partial chains, then one add): exact fp32, not bit-identical.
#ifndef KG_NPAIR
#define KG_NPAIR 3
#endif
constexpr int KG_NST = 2 * KG_NPAIR;      // stages: KG_NPAIR slab pairs (KG_NPAIR - 1 in flight behind the one being multiplied)

__global__ __launch_bounds__(512, 1) void gemm_xx_kg_kernel(XXP p) {
    constexpr int AW = 64, BW = 64, STAGE = BK * (AW + BW);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int nblk = p.tilesM * p.tilesN;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int tm = bid % p.tilesM, tn = bid / p.tilesM;
    const int z = blockIdx.z, zo = z / p.Zi, zi = z - zo * p.Zi;
    const int m0 = tm * 64, n0 = tn * 64;
    const int nk = (p.K + BK - 1) / BK, nint = (nk + 1) / 2;
    // a 1-KiB piece = 4 rows of a 64-wide image
    const int row_l = lane >> 4, col = (lane & 15) * 4;
    int mcol = m0 + col;
    if (mcol > (int)p.lda - 4) mcol = (int)p.lda - 4;               // stay inside the row (columns >= M are don't-care)
    const float* Ab = p.At + zo * p.sAo + zi * p.sAi + mcol;
    const float* Bb = p.B + zo * p.sBo + zi * p.sBi + n0 + col;
    auto issue = [&](int j) {                                         // this wave's two pieces of slab pair j (none when its slab is past K)
        const int kt = 2 * j + grp;
        if (kt >= nk) return;
        float* st = lds + (kt % KG_NST) * STAGE;
        const int kr = kt * BK + 4 * w4 + row_l;
        __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (long)kr * p.lda), (lptr_t)(st + w4 * 256), 16, 0, 0);
        const int krb = kr < p.K ? kr : p.K - 1;                      // finite filler; the matching A rows are zero
        __builtin_amdgcn_global_load_lds((gptr_t)(Bb + (long)krb * p.ldb), (lptr_t)(st + BK * AW + w4 * 256), 16, 0, 0);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < KG_NPAIR - 1; ++i) issue(i);
    const int lm = lane & 31, lk = lane >> 5;
    const int mine = (nk - grp + 1) / 2;       // slabs of this wave's group: kt = grp, grp + 2, ...
    for (int j = 0; j < nint; ++j) {
        // this wave's pieces of pair j have landed when only its pieces of the younger pairs (two operations each) are outstanding
        wait_vm_slabs<2, KG_NPAIR - 2>(mine - 1 - j);
        __builtin_amdgcn_s_barrier();          // every wave's pieces of pair j are in LDS; pair j - 1 is no longer read
        issue(j + KG_NPAIR - 1);               // into the stages pair j - 1 occupied
        const int kt = 2 * j + grp;
        if (kt < nk) {
            const float* As = lds + (kt % KG_NST) * STAGE;
            const float* Bs = As + BK * AW;
            float a[BK / 2], b[BK / 2];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                a[ks] = As[(2 * ks + lk) * AW + wm * 32 + lm];
                b[ks] = Bs[(2 * ks + lk) * BW + wn * 32 + lm];
            }
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[ks], acc, 0, 0, 0);
        }
    }
    __syncthreads();                           // every wave is done with the ring
    float* scr = lds + w4 * 1024;
    if (grp == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += scr[r * 64 + lane];
    const EpiP& ep = p.ep;
    float* Cb = ep.C + zo * ep.sCo + zi * ep.sCi;
    const float* Rb = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
    const float* Sb = ep.rowscale ? ep.rowscale + zo * ep.sSo + zi * ep.sSi : nullptr;
    f32x16 accs[1][1] = {{acc}};
    epilogue_vec<1, 1>(accs, scr, Cb, ep.ldc, Rb, ep.ldr, Sb, ep.alpha, ep.beta, m0 + wm * 32, n0 + wn * 32, p.M, p.N, lane);
}

template <int BM, int BN, int WM, int WN, bool LNP>
__global__ __launch_bounds__(GEMM_NT, (BM == 128 ? 2 : 1)) void gemm_xx_kernel(XXP p) {
    xx_body<BM, BN, WM, WN, LNP>(p, blockIdx.x, blockIdx.z);
}

// a plain product whose single row tile holds every channel of its pixels, with the LayerNorm statistics of its OUTPUT made by the epilogue
template <int BM>
__global__ __launch_bounds__(GEMM_NT, 1) void gemm_xx_stats_kernel(XXP p) { xx_body<BM, 128, 1, 4, false, 2>(p, blockIdx.x, blockIdx.z); }

// A PERSISTENT form of this kernel (one slab stream per workgroup over all its tiles, next tile's slabs in flight during this tile's
// last slab, epilogue stores counted into the vmcnt waits and draining under the next tile's MFMAs; bit-identical results) was built
// and measured in round 5 — NOTES.md item 13, profiles/r05_persistent_fp32.txt: 288 <- 96 at 8 x 128x128 103-109 us against 98-101 us
// (73 us with the stores compiled out, 88 us with every store hitting L2), staggered starts change nothing.  Removed.

// Up to three INDEPENDENT products in one grid (blockIdx.y names the product): the data gradients dV, dQ, dK of one MDTA block
// (rcot_gemm_kmajor_multi) are three launches of 8-50 workgroups each on the small levels — 10 us apiece of which most is the
// launch — and nothing orders them among each other.  A flat grid: the workgroups of product 0, then 1, then 2 (no idle workgroups).
struct XXP3 {
    XXP q[3];
    int first[4];        // workgroups [first[i], first[i + 1]) of the flat grid belong to product i (tiles x batch entries each)
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(GEMM_NT, (BM == 128 ? 2 : 1)) void gemm_xx_multi_kernel(XXP3 P) {
    const int g = blockIdx.x;
    const int y = g >= P.first[2] ? 2 : (g >= P.first[1] ? 1 : 0);
    const XXP& p = P.q[y];
    const int r = g - P.first[y], nt = p.tilesM * p.tilesN;
    const int bz = r / nt;
    xx_body<BM, BN, WM, WN, false>(p, r - bz * nt, bz);
}

// W [Co][Ci] (leading dim ldw) -> WT [ceil16(Ci)][ceil4(Co)] = W^T zero padded  (A^T operand of the forward product)
//                               and WP [ceil16(Co)][ceil4(Ci)] = W   zero padded  (A^T operand of the data gradient)
// and, for a projection that follows a LayerNorm (lnw/lnb given), the LN-folded forward operand of the bf16x3 kernel
// (gemm_x3.hip):  WTf = (W diag(lnw))^T  with  c12 = [ c1 = W lnw | c2 = W lnb ]  (2 x ceil4(Co) floats).
struct PackD {
    const float* W; long ldw; int Co, Ci;
    float* WT; float* WP;
    const float* lnw; const float* lnb; float* WTf; float* c12;
    unsigned char* WTs; unsigned char* WPs; unsigned char* WTfs;    // pre-split fragment packs of WT / WP / WTf (gemm_x3w.hip), optional
    unsigned char* WTs6; unsigned char* WPs6; unsigned char* WTfs6; // the same in THREE bf16 terms (3 KiB records; bf16x6 arithmetic), optional
};

// records (one per slab, 32-row tile and lane) of the pre-split packs: [ WTs | WPs | WTfs ]
__device__ __forceinline__ long pack_recs_t(const PackD& d) { return (long)((d.Ci + 15) / 16) * ((d.Co + 31) / 32) * 64; }
__device__ __forceinline__ long pack_recs_p(const PackD& d) { return (long)((d.Co + 15) / 16) * ((d.Ci + 31) / 32) * 64; }

__device__ __forceinline__ long pack_elems(const PackD& d) {
    const long nt = (long)((d.Ci + 15) & ~15) * ((d.Co + 3) & ~3), np = (long)((d.Co + 15) & ~15) * ((d.Ci + 3) & ~3);
    return nt + np + (d.WTf ? nt : 0) + (d.WTs ? pack_recs_t(d) : 0) + (d.WPs ? pack_recs_p(d) : 0) + (d.WTfs ? pack_recs_t(d) : 0) +
           (d.WTs6 ? pack_recs_t(d) : 0) + (d.WPs6 ? pack_recs_p(d) : 0) + (d.WTfs6 ? pack_recs_t(d) : 0);
}

// fp32 -> (rne bf16 hi, rne bf16 of the exact residual) for eight consecutive k of one A row: the MFMA operands of
// gemm_x3w.hip; lane (lm, kg) of row tile mt holds A[32 mt + lm][16 slab + 8 kg + (0..7)]
// NT = 3 (the *6 packs): a third term rne bf16 of what the first two left over; 3 KiB per (slab, row tile): [t0 | t1 | t2]
template <int NT>
__device__ __forceinline__ void pack_record(const PackD& d, long rec, int which) {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const int M = which == 1 ? d.Ci : d.Co, K = which == 1 ? d.Co : d.Ci;
    const int MT = (M + 31) / 32;
    const unsigned rq = (unsigned)(rec >> 6);
    const int lane = (int)(rec & 63), sl = (int)(rq / (unsigned)MT), mt = (int)(rq - (unsigned)sl * (unsigned)MT);
    const int m = mt * 32 + (lane & 31), k0 = sl * 16 + 8 * (lane >> 5);
    u32x4_ hi, lo, l2 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
        float x[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + 2 * kp + e;
            float v = 0.f;
            if (m < M && k < K) {
                v = which == 1 ? d.W[(long)k * d.ldw + m] : d.W[(long)m * d.ldw + k];
                if (which == 2) v *= d.lnw[k];
            }
            x[e] = v;
        }
        const f32x2_ a = {x[0], x[1]};
        const bf16x2_ h = __builtin_convertvector(a, bf16x2_);
        const f32x2_ r = a - __builtin_convertvector(h, f32x2_);
        hi[kp] = __builtin_bit_cast(unsigned, h);
        const bf16x2_ m = __builtin_convertvector(r, bf16x2_);
        lo[kp] = __builtin_bit_cast(unsigned, m);
        if (NT == 3) {
            const f32x2_ r2 = r - __builtin_convertvector(m, f32x2_);
            l2[kp] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_));
        }
    }
    unsigned char* base = NT == 3 ? (which == 0 ? d.WTs6 : which == 1 ? d.WPs6 : d.WTfs6) : (which == 0 ? d.WTs : which == 1 ? d.WPs : d.WTfs);
    unsigned char* out = base + ((long)(sl * MT + mt) * NT) * 1024 + lane * 16;
    *reinterpret_cast<u32x4_*>(out) = hi;
    *reinterpret_cast<u32x4_*>(out + 1024) = lo;
    if (NT == 3) *reinterpret_cast<u32x4_*>(out + 2048) = l2;
}

// element e of the pack space of one weight: [ WT | WP | WTf ]
__device__ __forceinline__ void pack_elem(const PackD& d, long e) {
    const int ldt = (d.Co + 3) & ~3, rt = (d.Ci + 15) & ~15;
    const int ldp = (d.Ci + 3) & ~3, rp = (d.Co + 15) & ~15;
    const long nt = (long)rt * ldt, np = (long)rp * ldp;
    // (the pack space of one weight is far below 2^31 elements: 32-bit divisions — the 64-bit ones were most of this kernel's time)
    if (e < nt) {
        const int k = (int)((unsigned)e / (unsigned)ldt), m = (int)(e - (long)k * ldt);
        d.WT[e] = (k < d.Ci && m < d.Co) ? d.W[(long)m * d.ldw + k] : 0.f;
    } else if (e < nt + np) {
        const long j = e - nt;
        const int k = (int)((unsigned)j / (unsigned)ldp), m = (int)(j - (long)k * ldp);
        d.WP[j] = (k < d.Co && m < d.Ci) ? d.W[(long)k * d.ldw + m] : 0.f;
    } else if (d.WTf && e < 2 * nt + np) {
        const long j = e - nt - np;
        const int k = (int)((unsigned)j / (unsigned)ldt), m = (int)(j - (long)k * ldt);
        d.WTf[j] = (k < d.Ci && m < d.Co) ? d.W[(long)m * d.ldw + k] * d.lnw[k] : 0.f;
    } else {
        long j = e - nt - np - (d.WTf ? nt : 0);
        if (d.WTs) {
            if (j < pack_recs_t(d)) return pack_record<2>(d, j, 0);
            j -= pack_recs_t(d);
        }
        if (d.WPs) {
            if (j < pack_recs_p(d)) return pack_record<2>(d, j, 1);
            j -= pack_recs_p(d);
        }
        if (d.WTfs) {
            if (j < pack_recs_t(d)) return pack_record<2>(d, j, 2);
            j -= pack_recs_t(d);
        }
        if (d.WTs6) {
            if (j < pack_recs_t(d)) return pack_record<3>(d, j, 0);
            j -= pack_recs_t(d);
        }
        if (d.WPs6) {
            if (j < pack_recs_p(d)) return pack_record<3>(d, j, 1);
            j -= pack_recs_p(d);
        }
        if (d.WTfs6 && j < pack_recs_t(d)) pack_record<3>(d, j, 2);
    }
}

// c1[m] = sum_k W[m][k] lnw[k], c2[m] = sum_k W[m][k] lnb[k] for rows [m0, m0 + 64): one wavefront per row, lanes over k
__device__ __forceinline__ void pack_rowsums(const PackD& d, int m0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ldc = (d.Co + 3) & ~3;
    for (int m = m0 + wave; m < m0 + 64 && m < ldc; m += 4) {
        float s1 = 0.f, s2 = 0.f;
        if (m < d.Co)
            for (int k = lane; k < d.Ci; k += 64) {
                const float w = d.W[(long)m * d.ldw + k];
                s1 += w * d.lnw[k];
                s2 += w * d.lnb[k];
            }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            d.c12[m] = s1;
            d.c12[ldc + m] = s2;
        }
    }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(PackD d) {
    const long n = pack_elems(d);
    const int nel = (int)((n + 1023) / 1024);
    if ((int)blockIdx.x < nel) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pack_elem(d, (long)blockIdx.x * 1024 + u * 256 + threadIdx.x);
    } else {
        pack_rowsums(d, ((int)blockIdx.x - nel) * 64);
    }
}

// Every repack of a network in ONE launch: tab[d] = { W, ldw, Co, Ci, WT, WP, first chunk, lnw, lnb, WTf, c12, WTs, WPs, WTfs, WTs6, WPs6,
// WTfs6, -, -, - } (20 x int64 per weight); the pack space of every weight is cut into 1024-element chunks followed (LN-folded weights)
// by 64-row chunks for c1/c2; chunk2desc[chunk] names the weight: one workgroup per chunk, no search.
__global__ __launch_bounds__(256) void pack_weights_kernel(const long long* __restrict__ tab,
                                                           const int* __restrict__ chunk2desc) {
    const long long* t = tab + (long)chunk2desc[blockIdx.x] * 20;
    PackD d;
    d.W = reinterpret_cast<const float*>(t[0]);
    d.ldw = t[1];
    d.Co = (int)t[2]; d.Ci = (int)t[3];
    d.WT = reinterpret_cast<float*>(t[4]);
    d.WP = reinterpret_cast<float*>(t[5]);
    d.lnw = reinterpret_cast<const float*>(t[7]);
    d.lnb = reinterpret_cast<const float*>(t[8]);
    d.WTf = reinterpret_cast<float*>(t[9]);
    d.c12 = reinterpret_cast<float*>(t[10]);
    d.WTs = reinterpret_cast<unsigned char*>(t[11]);
    d.WPs = reinterpret_cast<unsigned char*>(t[12]);
    d.WTfs = reinterpret_cast<unsigned char*>(t[13]);
    d.WTs6 = reinterpret_cast<unsigned char*>(t[14]);
    d.WPs6 = reinterpret_cast<unsigned char*>(t[15]);
    d.WTfs6 = reinterpret_cast<unsigned char*>(t[16]);
    const int cl = (int)((long)blockIdx.x - t[6]);
    const int nel = (int)((pack_elems(d) + 1023) / 1024);
    if (cl < nel) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pack_elem(d, (long)cl * 1024 + u * 256 + threadIdx.x);
    } else {
        pack_rowsums(d, (cl - nel) * 64);
    }
}

template <int BM, int BN, int WM, int WN>
int launch_xx(XXP p, bool ln, int Z, hipStream_t st) {
    constexpr int AW = BM <= 64 ? 64 : 128;
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = p.N / BN;
    const int nk = cdiv(p.K, BK);
    const size_t smem = sizeof(float) * ((size_t)xx_nst<BM, BN>() * BK * (AW + BN) + (ln ? 2 * (size_t)nk * BK : 0));
    dim3 grid(p.tilesM * p.tilesN, 1, Z);
    if (ln) {
        static bool once = (hipFuncSetAttribute((const void*)gemm_xx_kernel<BM, BN, WM, WN, true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        note_kernel("gemm_xx_kernel<%d, %d, %d, %d, true>", BM, BN, WM, WN);
        RCOT_LAUNCH((gemm_xx_kernel<BM, BN, WM, WN, true>), grid, dim3(GEMM_NT), smem, st, p);
    } else {
        static bool once = (hipFuncSetAttribute((const void*)gemm_xx_kernel<BM, BN, WM, WN, false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        note_kernel("gemm_xx_kernel<%d, %d, %d, %d, false>", BM, BN, WM, WN);
        RCOT_LAUNCH((gemm_xx_kernel<BM, BN, WM, WN, false>), grid, dim3(GEMM_NT), smem, st, p);
    }
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

inline EpiP p_ep_probe(float* C, long ldc, long sCo, long sCi, const float* R, long ldr, long sRo, long sRi) {
    EpiP e{};
    e.C = C; e.ldc = ldc; e.sCo = sCo; e.sCi = sCi; e.R = R; e.ldr = ldr; e.sRo = sRo; e.sRi = sRi;
    return e;
}

}  // namespace

namespace rcot {
int try_gemm_kmajor_x3(const float* At, long lda, long sAo, long sAi, const float* Bm, long ldb, long sBo, long sBi,
                       const EpiP& ep, const float* ln_mu, const float* ln_rs, long sLN, const float* ln_c1,
                       const float* ln_c2, int Zo, int Zi, int M, int N, int K, float* ws, size_t ws_bytes, hipStream_t st);
int try_gemm_kmajor_x3w(const float* At, long lda, long sAo, long sAi, const void* Apk, const float* Bm, long ldb, long sBo,
                        long sBi, const EpiP& ep, const float* ln_mu, const float* ln_rs, long sLN, const float* ln_c1,
                        const float* ln_c2, int Zo, int Zi, int M, int N, int K, float* ws, size_t ws_bytes, hipStream_t st,
                        bool ln_compute, int nterms);
}

extern "C" {

int rcot_gemm_kmajor(const float* At, long lda, long sAo, long sAi, int a_rows, const float* Bm, long ldb, long sBo,
                     long sBi, float* C, long ldc, long sCo, long sCi, const float* R, long ldr, long sRo, long sRi,
                     const float* rowscale, long sSo, long sSi, float* ln_mu, float* ln_rs, long sLN, int ln_compute,
                     const float* ln_w, const float* ln_b, const float* AtF, const float* ln_c12, const void* Asplit, int Zo,
                     int Zi, int M, int N, int K, float beta, float* ws, size_t ws_bytes, int prec, void* stream) {
    if (!At || !Bm || !C || Zo <= 0 || Zi <= 0 || M <= 0 || N <= 0 || K <= 0) return RCOT_EINVAL;
    if ((N % 64) || (lda & 3) || (ldb & 3) || (sAo & 3) || (sAi & 3) || (sBo & 3) || (sBi & 3) || lda < 4 ||
        !al16(At) || !al16(Bm))
        return RCOT_EINVAL;
    if (a_rows < cdiv(K, BK) * BK) return RCOT_EINVAL;              // zero rows up to ceil16(K) must exist
    if (!epi_vec_ok(p_ep_probe(C, ldc, sCo, sCi, R, ldr, sRo, sRi), N)) return RCOT_EINVAL;
    if ((long)Zo * Zi > 65535) return RCOT_EINVAL;
    const bool ln = ln_mu != nullptr;
    if (ln && (!ln_rs || !ln_w || !ln_b || K > 4096)) return RCOT_EINVAL;
    XXP p{};
    p.M = M; p.N = N; p.K = K; p.Zi = Zi;
    p.At = At; p.lda = lda; p.sAo = sAo; p.sAi = sAi;
    p.B = Bm; p.ldb = ldb; p.sBo = sBo; p.sBi = sBi;
    p.mu = ln_mu; p.rs = ln_rs; p.sLN = sLN; p.lnw = ln_w; p.lnb = ln_b;
    p.ep.C = C; p.ep.ldc = ldc; p.ep.sCo = sCo; p.ep.sCi = sCi;
    p.ep.R = R; p.ep.ldr = ldr; p.ep.sRo = sRo; p.ep.sRi = sRi;
    p.ep.rowscale = rowscale; p.ep.sSo = sSo; p.ep.sSi = sSi;
    p.ep.alpha = 1.f; p.ep.beta = beta; p.ep.lrelu = 1.f;
    const int Z = Zo * Zi;
    {   // streaming stores when the output is larger than the L2s can hold for its consumer (and nothing is accumulated into it)
        static const long mb = getenv("RCOT_XX_NTS_MB") ? atol(getenv("RCOT_XX_NTS_MB")) : 32;     // (read once, like every other switch here)
        p.ep.nts = (mb > 0 && beta == 0.f && 4L * M * N * Z >= (mb << 20)) ? 1 : 0;
    }
    // RCOT_PREC_BF16X1: the split-bf16 kernels and packs of RCOT_PREC_BF16X3 with the hi * hi product alone (EpiP::one)
    const bool x1 = prec == RCOT_PREC_BF16X1;
    if (x1) {
        prec = RCOT_PREC_BF16X3;
        p.ep.one = 1;
    }
    static const bool xx_ln_comp = !(getenv("RCOT_XX_LN_COMP") && atoi(getenv("RCOT_XX_LN_COMP")) == 0);
    if (ln_compute && prec == RCOT_PREC_FP32) {
        // exact fp32: gemm_xx_kernel makes the statistics of its pixel columns itself (XXP::ln_comp), in ln_stats_kernel's arithmetic
        // planes above 64x64 keep the rcot_ln_stats launch: there the workgroups' own pass over their column panel costs more than
        // the launch it replaces (a 128x128 block forward 553 -> 579 us, profiles/r05_ab_small_levels.txt); below, the two are equal
        // in time (145.8 vs 145.8 us at 32x32) and the projection is one launch less to enqueue
        static const int xx_ln_maxn = getenv("RCOT_XX_LN_MAXN") ? atoi(getenv("RCOT_XX_LN_MAXN")) : 4096;
        if (!ln || !xx_ln_comp || N > xx_ln_maxn || (sLN & 3) || !al16(ln_mu) || !al16(ln_rs) || (ldb & 3)) return RCOT_EUNSUPPORTED;
        p.ln_comp = 1;
        p.mu_out = ln_mu; p.rs_out = ln_rs;
    } else if (ln_compute) {
        // the statistics are made by the kernel that stages X: the producer / consumer kernel does that for the split arithmetics
        if (!ln || !AtF || !ln_c12 || !Asplit) return RCOT_EUNSUPPORTED;
        const int rcw = try_gemm_kmajor_x3w(AtF, lda, sAo, sAi, Asplit, Bm, ldb, sBo, sBi, p.ep, ln_mu, ln_rs, sLN, ln_c12,
                                            ln_c12 + ((M + 3) & ~3), Zo, Zi, M, N, K, ws, ws_bytes, (hipStream_t)stream, true,
                                            prec == RCOT_PREC_BF16X6 ? 3 : (prec == RCOT_PREC_BF16X3 ? 2 : 1));
        return rcw == -100 ? RCOT_EUNSUPPORTED : rcw;
    }
    if (prec == RCOT_PREC_BF16X6 && Asplit && (!ln || (AtF && ln_c12))) {
        // fp32-class arithmetic on the bf16 pipe: three-term split, six products — the producer / consumer kernel with the THREE-term
        // pack.  Products without a pack (two activations: the attention apply and its gradients) take the exact-fp32 kernels below:
        // a six-product form of gemm_x3_kernel (both operands split in three when the fragments leave LDS) passed its tests at the
        // fp32 bar and measured SLOWER than gemm_xx_kernel in the block (sum over the unit's blocks 75.2 vs 71.2 ms) — removed.
        const int rcw = try_gemm_kmajor_x3w(ln ? AtF : At, lda, sAo, sAi, Asplit, Bm, ldb, sBo, sBi, p.ep, ln_mu, ln_rs, sLN,
                                            ln ? ln_c12 : nullptr, ln ? ln_c12 + ((M + 3) & ~3) : nullptr, Zo, Zi, M, N, K, ws, ws_bytes,
                                            (hipStream_t)stream, false, 3);
        if (rcw != -100) return rcw;
    }
    if (prec == RCOT_PREC_BF16X3 && (!ln || (AtF && ln_c12))) {
        // with a LayerNorm prologue the split kernel multiplies the LN-FOLDED operand and applies mu/rstd in its epilogue
        const float* c1 = ln ? ln_c12 : nullptr;
        if (Asplit) {
            // a weight projection with its pre-split pack: the producer / consumer kernel (gemm_x3w.hip)
            const int rcw = try_gemm_kmajor_x3w(ln ? AtF : At, lda, sAo, sAi, Asplit, Bm, ldb, sBo, sBi, p.ep, ln_mu, ln_rs, sLN, c1,
                                                ln ? ln_c12 + ((M + 3) & ~3) : nullptr, Zo, Zi, M, N, K, ws, ws_bytes,
                                                (hipStream_t)stream, false, 2);
            if (rcw != -100) return rcw;
        }
        const int rc = try_gemm_kmajor_x3(ln ? AtF : At, lda, sAo, sAi, Bm, ldb, sBo, sBi, p.ep, ln_mu, ln_rs, sLN, c1,
                                          ln ? ln_c12 + ((M + 3) & ~3) : nullptr, Zo, Zi, M, N, K, ws, ws_bytes,
                                          (hipStream_t)stream);
        if (rc != -100) return rc;
    }
    // exact fp32 on the producer / consumer data path (x3p_kernel<.., NT = 1>: v_mfma_f32_32x32x2_f32 fed from the raw rings, LN fold
    // and epilogue statistics as the split kernels) — OPT-IN (RCOT_F32_PC=1): measured equal to gemm_xx_kernel on average (510 <- 96
    // + LN at 8 x 128x128: 160 vs 170 us, 1020 <- 192 at 32x32: 36 vs 41; 96 <- 510: 162 vs 135 — a single 128-row tile wastes a
    // quarter of the MFMA work on 96 rows, and with the consumers MFMA-bound their epilogue stores no longer hide behind anything)
    static const bool pc_f32 = getenv("RCOT_F32_PC") && atoi(getenv("RCOT_F32_PC")) == 1;
    static const int pc_f32_maxn = getenv("RCOT_F32_PC_MAXN") ? atoi(getenv("RCOT_F32_PC_MAXN")) : 1 << 30;
    if (pc_f32 && !p.ln_comp && N <= pc_f32_maxn && (!ln || (AtF && ln_c12))) {
        const int rcw = try_gemm_kmajor_x3w(ln ? AtF : At, lda, sAo, sAi, nullptr, Bm, ldb, sBo, sBi, p.ep, ln_mu, ln_rs, sLN,
                                            ln ? ln_c12 : nullptr, ln ? ln_c12 + ((M + 3) & ~3) : nullptr, Zo, Zi, M, N, K, ws, ws_bytes,
                                            (hipStream_t)stream, false, 1);
        if (rcw != -100) return rcw;
    }
    const long pad96 = (long)cdiv(M, 96) * 96, pad128 = (long)cdiv(M, 128) * 128;
    const long big_tiles = (long)cdiv(M, 128) * (N / 128) * Z;
    static const int force_tile = getenv("RCOT_XX_TILE") ? atoi(getenv("RCOT_XX_TILE")) : 0;     // tuning: 64 / 96 / 128 forces the tile
    if (force_tile == 64 || (force_tile && (N % 128))) return launch_xx<64, 64, 2, 2>(p, ln, Z, (hipStream_t)stream);
    if (force_tile == 96) return launch_xx<96, 128, 1, 4>(p, ln, Z, (hipStream_t)stream);
    if (force_tile == 128) return launch_xx<128, 128, 2, 2>(p, ln, Z, (hipStream_t)stream);
    if ((N % 128) == 0 && big_tiles >= 192) {
        // M <= 64 (the 48-channel level: 48 <- 48 / 127 / 144 / 254): a 64-row tile on 1 x 4 wavefronts — a third less MFMA work than the 96-row
        // tile these products rode in through round 5 (half of whose rows they left empty); same bits (round 6, RCOT_XX_M64=0 for the A/B)
        static const bool m64 = !(getenv("RCOT_XX_M64") && atoi(getenv("RCOT_XX_M64")) == 0);
        if (m64 && M <= 64) return launch_xx<64, 128, 1, 4>(p, ln, Z, (hipStream_t)stream);
        const bool w96 = pad96 < pad128;
        if (w96) return launch_xx<96, 128, 1, 4>(p, ln, Z, (hipStream_t)stream);
        return launch_xx<128, 128, 2, 2>(p, ln, Z, (hipStream_t)stream);
    }
    // long reductions on few workgroups (the data gradients of the 32x32 / 16x16 planes): the eight-wavefront k-group form
    static const bool kg_on = !(getenv("RCOT_XX_KG") && atoi(getenv("RCOT_XX_KG")) == 0);
    static const int kg_mink = getenv("RCOT_XX_KG_MINK") ? atoi(getenv("RCOT_XX_KG_MINK")) : 512;
    static const int kg_maxwg = getenv("RCOT_XX_KG_MAXWG") ? atoi(getenv("RCOT_XX_KG_MAXWG")) : 512;
    const long wgs64 = (long)cdiv(M, 64) * (N / 64) * Z;
    if (kg_on && !ln && K >= kg_mink && wgs64 <= kg_maxwg) {
        p.tilesM = cdiv(M, 64);
        p.tilesN = N / 64;
        const size_t smem = sizeof(float) * (size_t)KG_NST * BK * 128;
        static bool once = (hipFuncSetAttribute((const void*)gemm_xx_kg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        note_kernel("gemm_xx_kg_kernel");
        RCOT_LAUNCH(gemm_xx_kg_kernel, dim3(p.tilesM * p.tilesN, 1, Z), dim3(512), smem, (hipStream_t)stream, p);
        RCOT_LAUNCH_CHECK();
        return RCOT_OK;
    }
    return launch_xx<64, 64, 2, 2>(p, ln, Z, (hipStream_t)stream);   // small-N levels: 4x more workgroups
}

int rcot_gemm_kmajor_stats(const float* At, long lda, long sAo, long sAi, int a_rows, const float* Bm, long ldb, long sBo, long sBi,
                           float* C, long ldc, long sCo, long sCi, const float* R, long ldr, long sRo, long sRi, float* st_mu,
                           float* st_rs, long sST, int Zo, int Zi, int M, int N, int K, void* stream) {
    if (!At || !Bm || !C || !st_mu || !st_rs || Zo <= 0 || Zi != 1 || M <= 0 || N <= 0 || K <= 0) return RCOT_EINVAL;
    if (M > 96 || (N % 128)) return RCOT_EUNSUPPORTED;                // one row tile must hold every channel of its pixels
    if ((lda & 3) || (ldb & 3) || (sAo & 3) || (sAi & 3) || (sBo & 3) || (sBi & 3) || lda < 4 || !al16(At) || !al16(Bm) || (sST & 3) ||
        !al16(st_mu) || !al16(st_rs))
        return RCOT_EINVAL;
    if (a_rows < cdiv(K, BK) * BK) return RCOT_EINVAL;
    if (!epi_vec_ok(p_ep_probe(C, ldc, sCo, sCi, R, ldr, sRo, sRi), N)) return RCOT_EINVAL;
    if ((long)Zo > 65535) return RCOT_EINVAL;
    XXP p{};
    p.M = M; p.N = N; p.K = K; p.Zi = 1;
    p.At = At; p.lda = lda; p.sAo = sAo; p.sAi = sAi;
    p.B = Bm; p.ldb = ldb; p.sBo = sBo; p.sBi = sBi;
    p.ep.C = C; p.ep.ldc = ldc; p.ep.sCo = sCo; p.ep.sCi = sCi;
    p.ep.R = R; p.ep.ldr = ldr; p.ep.sRo = sRo; p.ep.sRi = sRi;
    p.ep.alpha = 1.f; p.ep.beta = 0.f; p.ep.lrelu = 1.f;
    p.st_mu = st_mu; p.st_rs = st_rs; p.sST = sST;
    static const long mb = getenv("RCOT_XX_NTS_MB") ? atol(getenv("RCOT_XX_NTS_MB")) : 32;
    p.ep.nts = (mb > 0 && 4L * M * N * Zo >= (mb << 20)) ? 1 : 0;
    p.tilesM = 1;
    p.tilesN = N / 128;
    hipStream_t st = (hipStream_t)stream;
    if (M <= 64) {
        const size_t smem = sizeof(float) * (size_t)XX_NST * BK * (64 + 128);
        static bool once = (hipFuncSetAttribute((const void*)gemm_xx_stats_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        note_kernel("gemm_xx_stats_kernel<64>");
        RCOT_LAUNCH(gemm_xx_stats_kernel<64>, dim3(p.tilesN, 1, Zo), dim3(GEMM_NT), smem, st, p);
    } else {
        const size_t smem = sizeof(float) * (size_t)XX_NST * BK * (128 + 128);
        static bool once = (hipFuncSetAttribute((const void*)gemm_xx_stats_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        note_kernel("gemm_xx_stats_kernel<96>");
        RCOT_LAUNCH(gemm_xx_stats_kernel<96>, dim3(p.tilesN, 1, Zo), dim3(GEMM_NT), smem, st, p);
    }
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_kmajor_desc_size(void) { return (int)sizeof(rcot_kmajor_desc); }

int rcot_gemm_kmajor_multi(const rcot_kmajor_desc* d, int n, int N, int prec, void* stream) {
    if (!d || n < 1 || n > 3 || N <= 0) return RCOT_EINVAL;
    // the split-bf16 arithmetic runs these products on gemm_x3_kernel (one launch each): not merged
    if (prec != RCOT_PREC_FP32 && prec != RCOT_PREC_BF16X6) return RCOT_EUNSUPPORTED;
    if (N % 64) return RCOT_EINVAL;
    XXP3 P{};
    for (int i = 0; i < 3; ++i) {
        const rcot_kmajor_desc& q = d[i < n ? i : 0];
        if (!q.At || !q.Bm || !q.C || q.Zo <= 0 || q.Zi <= 0 || q.M <= 0 || q.K <= 0) return RCOT_EINVAL;
        if ((q.lda & 3) || (q.ldb & 3) || (q.sAo & 3) || (q.sAi & 3) || (q.sBo & 3) || (q.sBi & 3) || q.lda < 4 || !al16(q.At) ||
            !al16(q.Bm))
            return RCOT_EINVAL;
        if (q.a_rows < cdiv(q.K, BK) * BK) return RCOT_EINVAL;
        if (!epi_vec_ok(p_ep_probe(q.C, q.ldc, q.sCo, q.sCi, q.R, q.ldr, q.sRo, q.sRi), N)) return RCOT_EINVAL;
        if ((long)q.Zo * q.Zi > 65535) return RCOT_EINVAL;
        XXP& p = P.q[i];
        p.M = q.M; p.N = N; p.K = q.K; p.Zi = q.Zi;
        p.At = q.At; p.lda = q.lda; p.sAo = q.sAo; p.sAi = q.sAi;
        p.B = q.Bm; p.ldb = q.ldb; p.sBo = q.sBo; p.sBi = q.sBi;
        p.ep.C = q.C; p.ep.ldc = q.ldc; p.ep.sCo = q.sCo; p.ep.sCi = q.sCi;
        p.ep.R = q.R; p.ep.ldr = q.ldr; p.ep.sRo = q.sRo; p.ep.sRi = q.sRi;
        p.ep.rowscale = q.rowscale; p.ep.sSo = q.sSo; p.ep.sSi = q.sSi;
        p.ep.alpha = 1.f; p.ep.beta = 0.f; p.ep.lrelu = 1.f;
    }
    int nz[3];
    for (int i = 0; i < 3; ++i) nz[i] = i < n ? d[i].Zo * d[i].Zi : 0;
    // one tile shape for the grid, chosen as rcot_gemm_kmajor chooses for the FIRST product (the full-channel one: dV); the per-element
    // summation order does not depend on the tile, so every product equals its own single launch bit for bit
    int total = 0;
    auto plan = [&](int BM, int BN) {
        total = 0;
        for (int i = 0; i < 3; ++i) {
            P.q[i].tilesM = cdiv(P.q[i].M, BM);
            P.q[i].tilesN = N / BN;
            P.first[i] = total;
            total += P.q[i].tilesM * P.q[i].tilesN * nz[i];
        }
        P.first[3] = total;
    };
    const int M0 = P.q[0].M;
    const long pad96 = (long)cdiv(M0, 96) * 96, pad128 = (long)cdiv(M0, 128) * 128;
    const long big_tiles = (long)cdiv(M0, 128) * (N / 128) * nz[0];
    hipStream_t st = (hipStream_t)stream;
#define RCOT_XX_MULTI(BM, BN, WM, WN)                                                                                        \
    do {                                                                                                                     \
        plan(BM, BN);                                                                                                        \
        constexpr int AW = BM <= 64 ? 64 : 128;                                                                              \
        const size_t smem = sizeof(float) * ((size_t)xx_nst<BM, BN>() * BK * (AW + BN));                                     \
        static bool once = (hipFuncSetAttribute((const void*)gemm_xx_multi_kernel<BM, BN, WM, WN>,                           \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);      \
        (void)once;                                                                                                          \
        note_kernel("gemm_xx_multi_kernel<%d, %d, %d, %d>", BM, BN, WM, WN);                                                 \
        RCOT_LAUNCH((gemm_xx_multi_kernel<BM, BN, WM, WN>), dim3(total), dim3(GEMM_NT), smem, st, P);                 \
    } while (0)
    if ((N % 128) == 0 && big_tiles >= 192) {
        static const bool m64 = !(getenv("RCOT_XX_M64") && atoi(getenv("RCOT_XX_M64")) == 0);
        int mmax = 0;
        for (int i = 0; i < n; ++i) mmax = P.q[i].M > mmax ? P.q[i].M : mmax;
        if (m64 && mmax <= 64) RCOT_XX_MULTI(64, 128, 1, 4);
        else if (pad96 < pad128) RCOT_XX_MULTI(96, 128, 1, 4);
        else RCOT_XX_MULTI(128, 128, 2, 2);
    } else {
        RCOT_XX_MULTI(64, 64, 2, 2);
    }
#undef RCOT_XX_MULTI
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_pack_weight(const float* W, long ldw, int Co, int Ci, float* WT, float* WP, const float* ln_w, const float* ln_b,
                     float* WTf, float* c12, void* WTs, void* WPs, void* WTfs, void* WTs6, void* WPs6, void* WTfs6, void* stream) {
    if (!W || !WT || !WP || Co <= 0 || Ci <= 0) return RCOT_EINVAL;
    if (WTf && (!ln_w || !ln_b || !c12)) return RCOT_EINVAL;
    if ((WTfs || WTfs6) && !WTf) return RCOT_EINVAL;
    PackD d{W, ldw, Co, Ci, WT, WP, ln_w, ln_b, WTf, c12, (unsigned char*)WTs, (unsigned char*)WPs, (unsigned char*)WTfs,
            (unsigned char*)WTs6, (unsigned char*)WPs6, (unsigned char*)WTfs6};
    const long nt = (long)((Ci + 15) & ~15) * ((Co + 3) & ~3), np = (long)((Co + 15) & ~15) * ((Ci + 3) & ~3);
    const long rt = (long)((Ci + 15) / 16) * ((Co + 31) / 32) * 64, rp = (long)((Co + 15) / 16) * ((Ci + 31) / 32) * 64;
    const long n = nt + np + (WTf ? nt : 0) + (WTs ? rt : 0) + (WPs ? rp : 0) + (WTfs ? rt : 0) + (WTs6 ? rt : 0) + (WPs6 ? rp : 0) +
                   (WTfs6 ? rt : 0);
    const int grid = (int)((n + 1023) / 1024) + (WTf ? (((Co + 3) & ~3) + 63) / 64 : 0);
    RCOT_LAUNCH(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_pack_weights(const long long* table, const int* chunk2desc, int nchunks, void* stream) {
    if (!table || !chunk2desc || nchunks <= 0) return RCOT_EINVAL;
    RCOT_LAUNCH(pack_weights_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunk2desc);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // extern "C"
