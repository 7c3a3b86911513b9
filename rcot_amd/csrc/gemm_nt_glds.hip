// LDS-DMA pipelined fp32 MFMA kernel for the PIXEL-REDUCTION products of the transport map:
//
//     C[z] (M x N) = sum_k A[z][m][k] * LN?(B[z])[n][k]        K = pixels (x batch), both operands K-contiguous
//
// i.e. the 1x1 weight gradients  dW = sum_b dY_b X_b^T  (batch folded into K, LayerNorm of X applied on the fly)
// and MDTA's Gram matrices  q k^T  /  dM = dY V^T.  M, N are channel counts (<= ~2000), K is 10^4..10^5, so the
// reduction is split over workgroups into slabs that a second kernel sums deterministically (+ beta*C).
//
// Data movement: 16-pixel K-slabs of 128 rows per operand are DMA'd (global_load_lds_dwordx4) into an NST-stage LDS
// ring.  The DMA image is lane-linear, so each lane places the 16-byte chunk  (row, kq)  at physical chunk
// kq ^ ((row>>2)&3)  by choosing its SOURCE address; fragment reads apply the same XOR and are bank-conflict
// free.  One 8-byte read per lane feeds two 32x32x2 MFMA k-steps (lanes 0-31 take elements 0,1 of the quad, lanes
// 32-63 elements 2,3).  LayerNorm statistics of the slab's 16 pixels travel in the same ring (one 4-byte DMA op
// per wave) and the affine normalisation is applied to the B fragment in registers.
#include <cstdlib>
#include "gemm_core.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

#include "gemm_nt_body.h"

namespace rcot_nt {

template <int TM, int TN, int WM, int WN, bool LNP, bool X3, bool X6 = false, int COOP = 0>
__global__ __launch_bounds__(GEMM_NT, 2) void gemm_nt_kernel(NTP p) {
    nt_body<TM, TN, WM, WN, LNP, X3, X6, COOP>(p, blockIdx.x, blockIdx.z);
}

// C = beta*C + sum_s slab_s   (full epilogue options of EpiP).  64 outputs per workgroup (one per lane, coalesced
// along n); the 4 wavefronts take interleaved quarters of the S slabs with 4 loads in flight each, then combine
// through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void nt_reduce_kernel(const float* __restrict__ ws, int ldws, int S, int M, int N, int Z,
                                                        int Zi, EpiP ep) {
    __shared__ float part[4][64];
    const long mn = (long)M * N, total = mn * Z;
    const long slab = (long)M * ldws;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long idx = base + lane;
        float a = 0.f;
        int z = 0, m = 0, n = 0;
        if (idx < total) {
            z = (int)(idx / mn);
            const long r = idx - (long)z * mn;
            m = (int)(r / N);
            n = (int)(r - (long)m * N);
            const float* w = ws + (long)z * S * slab + (long)m * ldws + n;
            // 8 loads in flight per wavefront (the kernel is latency-bound: up to 64 slabs per wavefront), fixed order
            float acc8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc8[q] = 0.f;
            int s = wave;
            for (; s + 28 < S; s += 32) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc8[q] += w[(long)(s + 4 * q) * slab];
            }
            for (; s < S; s += 4) acc8[0] += w[(long)s * slab];
            a = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
        }
        __syncthreads();
        part[wave][lane] = a;
        __syncthreads();
        if (wave == 0 && idx < total)
            epi_store(ep, z / Zi, z % Zi, m, n, (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
    }
}

// the same for S <= 8: one output per thread, all S loads in flight
__global__ __launch_bounds__(256) void nt_reduce_few_kernel(const float* __restrict__ ws, int ldws, int S, int M, int N, int Z, int Zi,
                                                            EpiP ep) {
    const long mn = (long)M * N, total = mn * Z;
    const long slab = (long)M * ldws;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int z = (int)(idx / mn);
        const long r = idx - (long)z * mn;
        const int m = (int)(r / N), n = (int)(r - (long)m * N);
        const float* w = ws + (long)z * S * slab + (long)m * ldws + n;
        float v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = s < S ? w[(long)s * slab] : 0.f;
        epi_store(ep, z / Zi, z % Zi, m, n, ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
    }
}

// reduce = false: leave the S split-K slabs [z][s][M][ldws] in p.ws for the caller (rcot_conv1x1_wgrad_slabs)
// COOP: the operand shared by the four wavefronts is normalised / split once per workgroup (gemm_nt_body.h); two fragment buffers behind the ring
template <int TM, int TN, int WM, int WN, bool X3, bool X6 = false, int COOP = 0>
int launch_nt(NTP p, const EpiP& ep, int Z, hipStream_t st, bool reduce) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = cdiv(p.N, BN);
    const size_t smem = sizeof(float) * (size_t)NST * STAGE + (COOP ? 2 * (size_t)(X6 ? 3 : 2) * 2 * 128 * 16 : 0);
    dim3 grid(p.tilesM * p.tilesN * p.S, 1, Z);
    // (the name rocprofv3 and the library's own per-launch profile list: bench.py matches its rows by it)
    if (X6) note_kernel("gemm_nt_kernel<%d, %d, %d, %d, %s, true, true, %d>", TM, TN, WM, WN, tf(p.mu != nullptr), COOP);
    else note_kernel("gemm_nt_kernel<%d, %d, %d, %d, %s, %s, false, %d>", TM, TN, WM, WN, tf(p.mu != nullptr), tf(X3), COOP);
    if (p.mu) {
        static bool once = (hipFuncSetAttribute((const void*)gemm_nt_kernel<TM, TN, WM, WN, true, X3, X6, COOP>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        RCOT_LAUNCH((gemm_nt_kernel<TM, TN, WM, WN, true, X3, X6, COOP>), grid, dim3(GEMM_NT), smem, st, p);
    } else {
        static bool once = (hipFuncSetAttribute((const void*)gemm_nt_kernel<TM, TN, WM, WN, false, X3, X6, COOP>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
        (void)once;
        RCOT_LAUNCH((gemm_nt_kernel<TM, TN, WM, WN, false, X3, X6, COOP>), grid, dim3(GEMM_NT), smem, st, p);
    }
    RCOT_LAUNCH_CHECK();
    if (!reduce) return RCOT_OK;
    const long total = (long)p.M * p.N * Z;
    if (p.S <= 8) {                                           // few slabs: one output per THREAD (the wave-per-slab-quarter form
        long nb = (total + 255) / 256;                        // leaves 3 of 4 wavefronts idle and needs 4x the workgroups)
        if (nb > 4096) nb = 4096;
        RCOT_LAUNCH(nt_reduce_few_kernel, dim3((int)nb), dim3(256), 0, st, p.ws, p.ldws, p.S, p.M, p.N, Z, p.Zi, ep);
        RCOT_LAUNCH_CHECK();
        return RCOT_OK;
    }
    long nb = (total + 63) / 64;
    if (nb > 8192) nb = 8192;
    RCOT_LAUNCH(nt_reduce_kernel, dim3((int)nb), dim3(256), 0, st, p.ws, p.ldws, p.S, p.M, p.N, Z, p.Zi, ep);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // namespace rcot_nt

namespace rcot {
static int g_nt_coop_override = -1;      // rcot_debug_nt_coop(): -1 = the environment's RCOT_NT_COOP (read once), else the mask to use

// Parameter block, tile shape (cfg 0..5: 128x128, 128x96, 96x128, 128x64, 64x128, 64x64) and split factor of one product.
// Returns RCOT_OK, or -100 when the product is not eligible for this kernel family.  Used by try_gemm_nt_glds below and by the
// paired data-gradient + weight-gradient launch of gemm_x3w.hip.
int nt_configure(int M, int N, int K, int Zo, int Zi, const float* A, long lda, long sAo, long sAi, const float* B, long ldb,
                 long sBo, long sBi, int Kb, long sAk, long sBk, const float* mu, const float* rs, long sLNb, const float* lnw,
                 const float* lnb, float* ws, size_t ws_bytes, int prec, int conv_wp, rcot_nt::NTP* out, int* out_cfg, int slots) {
    using namespace rcot_nt;
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int Z = Zo * Zi;
    if (!ws || (K & 15) || (lda & 3) || (ldb & 3) || (sAo & 3) || (sAi & 3) || (sBo & 3) || (sBi & 3) || (sAk & 3) ||
        (sBk & 3) || !a16(A) || !a16(B) || (Kb && (Kb & 15)) || Z > 16384)
        return -100;
    if (mu && ((sLNb & 3) || !a16(mu) || !a16(rs))) return -100;
    static const bool old_gate = getenv("RCOT_NT_OLD") != nullptr;     // debugging: the round-1 gate
    if (old_gate && (M < 96 || N < 96)) return -100;
    if (M < 33 || N < 33) return -100;                 // 64- or 128-row DMA images: tiny channel counts stay on the 64x64 engine
    NTP p{};
    p.M = M; p.N = N; p.K = K; p.Zi = Zi;
    p.A = A; p.lda = lda; p.sAo = sAo; p.sAi = sAi;
    p.B = B; p.ldb = ldb; p.sBo = sBo; p.sBi = sBi;
    p.Kb = Kb; p.sAk = sAk; p.sBk = sBk;
    p.mu = mu; p.rs = rs; p.sLNb = sLNb; p.lnw = lnw; p.lnb = lnb;
    p.ws = ws;
    p.conv_taps = conv_wp ? 9 : 0; p.conv_wp = conv_wp;
    p.one = prec == RCOT_PREC_BF16X1 ? 1 : 0;
    p.ldws = (N + 3) & ~3;
    // tile shape: least padded area among 128x128, 128x96, 96x128
    // workgroup tile: the (bm, bn) of {128, 96, 64} x {128, 96, 64} (not 96 x 96, 96 x 64, 64 x 96) with the least padded area
    static const int cand_bm[6] = {128, 128, 96, 128, 64, 64}, cand_bn[6] = {128, 96, 128, 64, 128, 64};
    int cfg = 0;
    long best = -1;
    for (int c = 0; c < 6; ++c) {
        const long area = (long)cdiv(M, cand_bm[c]) * cand_bm[c] * cdiv(N, cand_bn[c]) * cand_bn[c];
        if (best < 0 || area < best) { best = area; cfg = c; }
    }
    // short reductions (the 16x16 / 32x32 levels: K = batch * pixels <= 8192): the product is a few microseconds of work and the
    // slabs dominate (written once, read once by the reduce: 8 M N S bytes).  64 x 64 tiles need a quarter of the split factor
    // of 128 x 128 tiles for the same number of workgroups: 4x fewer slab bytes for the kernel AND for the launch that sums them
    // (rcot_block_param_reduce).  MEASURED (round 3): the 64 x 64 form runs the weight gradients of the 16x16 level in 20 / 45 / 26 us against
    // 15 / 24 / 18 us for the 128-wide tiles with their larger split: NOT adopted, RCOT_NT_SMALLK=1 turns it on for A/B runs.
    static const bool smallk = getenv("RCOT_NT_SMALLK") && atoi(getenv("RCOT_NT_SMALLK")) == 1;
    const bool small = smallk && Z == 1 && K <= 8192 && (long)cdiv(M, 64) * cdiv(N, 64) >= 24;
    if (small) cfg = 5;                                       // (off by default below: measured slower, see DESIGN)
    const int bm = cand_bm[cfg], bn = cand_bn[cfg];
    const long tiles = (long)cdiv(M, bm) * cdiv(N, bn) * Z;
    const int nslab = K / BK;
    // split factor from a two-term cost model: MFMA time at the parallel efficiency the grid reaches, plus
    // operand + slab (write once, read once) traffic
    const double flops = 2.0 * M * N * (double)K * Z;
    const double in_bytes = 4.0 * ((double)M * cdiv(N, bn) + (double)N) * K * Z;
    static const int slots_env = getenv("RCOT_NT_SLOTS") ? atoi(getenv("RCOT_NT_SLOTS")) : 0;     // tuning: workgroup slots the split factor aims at
    if (slots_env > 0 && slots == 640) slots = slots_env;
    long S = 1;
    double best_t = 1e30;
    for (long cand = 1; cand <= nslab / 4 && cand * Z <= 65535; cand *= 2) {
        const double eff = fmin(1.0, (double)(tiles * cand) / (double)slots);   // (slots: 640, half of it in a paired launch)
        const double t = flops / ((prec == 2 ? 2.0e14 : (prec ? 3.0e14 : 9.0e13)) * eff) + (in_bytes + 8.0 * M * N * (double)cand * Z) / 3.5e12;
        if (t < best_t) { best_t = t; S = cand; }
    }
    if (small) {                                              // fill ~640 workgroup slots, at least 8 slabs per piece
        S = 1;
        while (S * 2 * tiles <= slots && S * 2 <= nslab / 8) S *= 2;
    }
    const size_t per = (size_t)M * p.ldws * Z * sizeof(float);
    while (S > 1 && per * S > ws_bytes) --S;
    if (per * S > ws_bytes) return -100;
    if ((long)Z * S > 65535) S = 65535 / Z;
    p.kchunk = cdiv(cdiv(nslab, (int)S), 1) * BK;
    p.S = cdiv(K, p.kchunk);
    constexpr int bms[6] = {128, 128, 96, 128, 64, 64}, bns[6] = {128, 96, 128, 64, 128, 64};
    p.tilesM = cdiv(p.M, bms[cfg]);
    p.tilesN = cdiv(p.N, bns[cfg]);
    *out = p;
    *out_cfg = cfg;
    return RCOT_OK;
}

// Returns RCOT_OK after launching, or a negative "not eligible" code (-100) so that the caller can use the general engine.
int try_gemm_nt_glds(int M, int N, int K, int Zo, int Zi, const float* A, long lda, long sAo, long sAi, const float* B,
                     long ldb, long sBo, long sBi, int Kb, long sAk, long sBk, const float* mu, const float* rs,
                     long sLNb, const float* lnw, const float* lnb, const EpiP& ep, float* ws, size_t ws_bytes,
                     hipStream_t st, int prec, int* slabs_S, int* slabs_ld, int conv_wp) {
    using namespace rcot_nt;
    NTP p{};
    int cfg = 0;
    const int rc = nt_configure(M, N, K, Zo, Zi, A, lda, sAo, sAi, B, ldb, sBo, sBi, Kb, sAk, sBk, mu, rs, sLNb, lnw, lnb, ws, ws_bytes,
                                prec, conv_wp, &p, &cfg);
    if (rc != RCOT_OK) return rc;
    const int Z = Zo * Zi;
    const bool reduce = slabs_S == nullptr;
    if (!reduce) {
        *slabs_S = p.S;
        *slabs_ld = p.ldws;
    }
    // RCOT_NT_COOP (A/B switch): bit 0 = bf16x6, bit 1 = bf16x3 products split once per workgroup (tiles whose four waves share an operand)
    // read ONCE (2 700 replayed launches per iteration pass here); rcot_nt_coop_override() below is the test hook that
    // tests/test_x3_gpu.py uses to compare the two forms' bits inside one process
    static const int coop_env = getenv("RCOT_NT_COOP") ? atoi(getenv("RCOT_NT_COOP")) : 3;
    const int coop = g_nt_coop_override >= 0 ? g_nt_coop_override : coop_env;
    // (an exact-fp32 form — LayerNorm of the shared operand once per workgroup, in place, and a slab loop unrolled over the ring stages with
    // immediate offsets: 97 -> 34 VALU instructions per slab — was built and measured: 175.0 / 117.1 / 46.8 us -> 171.6 / 120.7 / 49.0, the
    // iteration 80.7 vs 80.5 ms: nothing, removed; profiles/r05_ab_coop_fp32.txt, NOTES round 5 item 20)
    if (prec == RCOT_PREC_BF16X6 && (coop & 1) && cfg >= 1 && cfg <= 4) {
        if (cfg == 1) return launch_nt<1, 3, 4, 1, true, true, 1>(p, ep, Z, st, reduce);
        if (cfg == 2) return launch_nt<3, 1, 1, 4, true, true, 2>(p, ep, Z, st, reduce);
        if (cfg == 3) return launch_nt<1, 2, 4, 1, true, true, 1>(p, ep, Z, st, reduce);
        return launch_nt<2, 1, 1, 4, true, true, 2>(p, ep, Z, st, reduce);
    }
    if (prec && prec != RCOT_PREC_BF16X6 && (coop & 2) && cfg >= 1 && cfg <= 4) {
        if (cfg == 1) return launch_nt<1, 3, 4, 1, true, false, 1>(p, ep, Z, st, reduce);
        if (cfg == 2) return launch_nt<3, 1, 1, 4, true, false, 2>(p, ep, Z, st, reduce);
        if (cfg == 3) return launch_nt<1, 2, 4, 1, true, false, 1>(p, ep, Z, st, reduce);
        return launch_nt<2, 1, 1, 4, true, false, 2>(p, ep, Z, st, reduce);
    }
    if (prec == RCOT_PREC_BF16X6) {
        if (cfg == 1) return launch_nt<1, 3, 4, 1, true, true>(p, ep, Z, st, reduce);
        if (cfg == 2) return launch_nt<3, 1, 1, 4, true, true>(p, ep, Z, st, reduce);
        if (cfg == 3) return launch_nt<1, 2, 4, 1, true, true>(p, ep, Z, st, reduce);
        if (cfg == 4) return launch_nt<2, 1, 1, 4, true, true>(p, ep, Z, st, reduce);
        if (cfg == 5) return launch_nt<1, 1, 2, 2, true, true>(p, ep, Z, st, reduce);
        return launch_nt<2, 2, 2, 2, true, true>(p, ep, Z, st, reduce);
    }
    if (prec) {
        if (cfg == 1) return launch_nt<1, 3, 4, 1, true>(p, ep, Z, st, reduce);
        if (cfg == 2) return launch_nt<3, 1, 1, 4, true>(p, ep, Z, st, reduce);
        if (cfg == 3) return launch_nt<1, 2, 4, 1, true>(p, ep, Z, st, reduce);
        if (cfg == 4) return launch_nt<2, 1, 1, 4, true>(p, ep, Z, st, reduce);
        if (cfg == 5) return launch_nt<1, 1, 2, 2, true>(p, ep, Z, st, reduce);
        return launch_nt<2, 2, 2, 2, true>(p, ep, Z, st, reduce);
    }
    if (cfg == 1) return launch_nt<1, 3, 4, 1, false>(p, ep, Z, st, reduce);
    if (cfg == 2) return launch_nt<3, 1, 1, 4, false>(p, ep, Z, st, reduce);
    if (cfg == 3) return launch_nt<1, 2, 4, 1, false>(p, ep, Z, st, reduce);
    if (cfg == 4) return launch_nt<2, 1, 1, 4, false>(p, ep, Z, st, reduce);
    if (cfg == 5) return launch_nt<1, 1, 2, 2, false>(p, ep, Z, st, reduce);
    return launch_nt<2, 2, 2, 2, false>(p, ep, Z, st, reduce);
}

}  // namespace rcot

// Test hook (tests/test_x3_gpu.py::test_cooperative_split_changes_no_bit): which arithmetics split the operand a tile's four wavefronts
// share once per workgroup — bit 0 bf16x6, bit 1 bf16x3; -1 gives the choice back to RCOT_NT_COOP (read once per process, default 3).
extern "C" int rcot_debug_nt_coop(int mask) {
    if (mask < -1 || mask > 3) return RCOT_EINVAL;
    rcot::g_nt_coop_override = mask;
    return RCOT_OK;
}
