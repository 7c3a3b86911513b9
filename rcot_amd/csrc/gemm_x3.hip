// bf16x3 split-MFMA variant of the K-major projection GEMM (see gemm_glds.hip for the exact-fp32 form):
//
//     C[z] (M x N) = A[z] (M x K) * LN?(B[z]) (K x N)        A given K-major (At[k][m]), B[k][n], fp32 in HBM
//
// fp32 operands are split on chip, when a fragment is read from LDS, into two bfloat16 terms x = hi + lo
// (hi = rne_bf16(x), lo = rne_bf16(x - hi); 16 mantissa bits together) and every fp32 product block becomes THREE
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation:  lo*hi + hi*lo + hi*hi  (the lo*lo term, 2^-18 relative, is
// dropped).  Per-product relative error <= ~1e-5, measured 5e-6 of max|C| on the transport map's projection shapes
// (scripts/micro/bf16x3_error.py); 3 x 32 cycles per 32x32x16 block against 8 x 64 cycles of v_mfma_f32_32x32x2_f32:
// the MFMA floor drops 5.3x and the projections become HBM-bound.  HBM layouts, the pack contract and the
// epilogue options are those of rcot_gemm_kmajor; precision is selected per call (prec = RCOT_PREC_BF16X3).
//
// Tiling.  A wavefront owns 32*TM rows x 128 columns.  The 128 columns are FOUR INTERLEAVED 32-column MFMA tiles:
// tile t holds columns {4j + t}.  A lane (j = lane & 31, kg = lane >> 5) therefore fetches its B operands as eight
// ds_read_b128 (k = 8kg..8kg+7, columns 4j..4j+3) straight from the lane-linear K-major DMA image, conflict-free
// (vs 32 ds_read_b32 for four plain tiles), and in the epilogue it holds, for every row of its accumulators, the
// four CONSECUTIVE columns 4j..4j+3: results leave as 16-byte row-contiguous stores with no LDS transpose.
// WM x WN wavefronts (1..4) form a (32*TM*WM) x (128*WN) workgroup tile; operands arrive through the same 3-stage
// LDS-DMA ring (global_load_lds_dwordx4, counted vmcnt, one s_barrier per 16-row slab).
//
// Split-K.  The small levels (16x16 / 32x32 pixels per image) have few output tiles and long reductions (C = 384 outputs
// from K = 2042: 48 tiles of 128 slabs on 256 CUs).  There the K range of a tile is cut into S pieces that run as
// separate tiles, each storing its partial result to a slab of the caller's workspace; x3_reduce_kernel sums the slabs in a
// fixed order and applies the epilogue (the linear part of the LN fold is applied per piece, its constants by piece 0).
//
// LayerNorm prologue.  W * LN(X) with LN(X)[k][n] = (X[k][n] - mu[n]) rs[n] w[k] + b[k] is evaluated as
//     rs[n] * ( (W diag(w)) X )[m][n]  -  rs[n] mu[n] c1[m]  +  c2[m],      c1 = W w,  c2 = W b,
// i.e. the main loop multiplies the RAW activations by the LN-folded weight pack (At = (W diag(w))^T, made together with
// c1, c2 by rcot_pack_weights) and the per-pixel statistics enter once, in the epilogue.  The four VALU operations per
// B element that the in-loop normalisation costs (as many as the bf16 split itself) disappear from the slab loop.
#include "gemm_core.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace rcot_x3 {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef X3_NST
#define X3_NST 3
#endif
constexpr int NST_STREAM = X3_NST;   // LDS ring stages of the streaming form (2-3 workgroups per CU share the LDS)

struct X3P {
    int M, N, K, Zi, tilesM, tilesN, ntiles;
    int S, kchunk;                                    // split-K: S K-ranges of kchunk slabs, each written to its own slab of ws
    float* ws;
    const float* At; long lda, sAo, sAi;
    const float* B;  long ldb, sBo, sBi;
    const float* mu; const float* rs; long sLN;      // LN statistics per pixel (LNP)
    const float* c1; const float* c2;                 // LN fold constants per output row (LNP)
    EpiP ep;
#ifdef X3_TRACE
    unsigned long long* trace;                        // debug build: 64 time stamps (100 MHz) per workgroup, first tile only
#endif
};

#ifdef X3_TRACE
static unsigned long long* g_x3_trace = nullptr;
extern "C" int rcot_x3_set_trace(void* q) { g_x3_trace = (unsigned long long*)q; return 0; }
#define X3_STAMP(i) do { if (p.trace && tid == 0 && t == vb) p.trace[(long)blockIdx.x * 64 + (i)] = wall_clock64(); } while (0)
#else
#define X3_STAMP(i) do {} while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// wait until at most y slabs (PW vm operations each) of this wave are still in flight; y is wave-uniform, 0 <= y < NST
template <int PW, int NST>
__device__ __forceinline__ void wait_slabs(int y) {
    static_assert((NST - 1) * PW <= 63, "vmcnt is a 6-bit field");
    if (y <= 0) wait_vm<0>();
    else if (y == 1) wait_vm<PW>();
    else if (NST > 2 && y == 2) wait_vm<(NST > 2 ? 2 : 0) * PW>();
    else if (NST > 3 && y == 3) wait_vm<(NST > 3 ? 3 : 0) * PW>();
    else if (NST > 4 && y == 4) wait_vm<(NST > 4 ? 4 : 0) * PW>();
    else if (NST > 5 && y == 5) wait_vm<(NST > 5 ? 5 : 0) * PW>();
    else if (NST > 6 && y == 6) wait_vm<(NST > 6 ? 6 : 0) * PW>();
    else wait_vm<(NST > 7 ? 7 : 0) * PW>();
}

// eight fp32 values (consecutive k of one row / column) -> the hi and lo bf16x8 MFMA operands
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2 v = {x[q], x[q + 1]};
        const bf16x2 h = __builtin_convertvector(v, bf16x2);                 // v_cvt_pk_bf16_f32 (rne)
        const f32x2 r = v - __builtin_convertvector(h, f32x2);               // exact in fp32
        const bf16x2 l = __builtin_convertvector(r, bf16x2);
        hi[q] = h[0]; hi[q + 1] = h[1];
        lo[q] = l[0]; lo[q + 1] = l[1];
    }
}

// second launch-bound = waves per SIMD the register allocation must leave room for: 3 with one accumulator row
// (64 AGPRs), 2 with two (128); the compiler otherwise spends ~350 registers on a one-wave schedule.
//
// PERSISTENT: the grid is sized to what is resident (host side) and every workgroup walks tiles t = vb, vb + G, ...
// with ONE slab ring running across tile boundaries: the first two slabs (and the LayerNorm statistics) of the next
// tile are requested while the last two slabs of the current tile are multiplied, so neither the load latency at the
// head of a tile nor the store burst at its tail leaves the memory pipe idle.  With K = 96..510 a tile is only 6..32
// slabs long; the non-persistent form of this kernel spent most of its time in those two bubbles (2.5-3 TB/s).
template <int TM, int WM, int WN, bool LNP, int NST>
__global__ __launch_bounds__(64 * WM * WN, (TM == 1 ? 3 : (TM == 2 ? 2 : 1))) void gemm_x3_kernel(X3P p) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 128 * WN;
    constexpr int AW = BM <= 64 ? 64 : 128, BW = BN;
    constexpr int PA = BK * AW / 256, PB = BK * BW / 256, PT = PA + PB;   // 1-KiB DMA pieces per slab
    constexpr int PW = (PT + NW - 1) / NW;                                // DMA ops per wave per slab (padded with dummies)
    constexpr int STAGE = BK * (AW + BW);
    static_assert(NW >= 1 && NW <= 4 && BM <= 128 && WN <= NW, "tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];          // ring | dummy piece
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int G = gridDim.x;
    const int vb = xcd_remap(blockIdx.x, G);                            // consecutive virtual ids share an XCD (and its L2)
    const int ntiles = p.ntiles;
    const int nk_all = (p.K + BK - 1) / BK;                             // slabs of the whole reduction; every piece has >= 2 (host)
    float* dummy = lds + NST * STAGE;
    const int lm = lane & 31, kg = lane >> 5;

    // ---- DMA addressing (piece q of an image = 256/W rows; lane -> (row in piece, 4 columns))
    const int arow_l = lane / (AW / 4), acol = (lane % (AW / 4)) * 4;
    const int brow_l = lane / (BW / 4), bcol = (lane % (BW / 4)) * 4;

    // ---- issue cursor: two slabs ahead of the consumer, across tiles
    // tile id -> (tm fastest, then the K piece, then the column tile, then z)
    int it = vb, ikt = 0, ik0 = 0, ink = 0, gi = 0;
    const float* iAb = nullptr;
    const float* iBb = nullptr;
    auto icursor = [&]() {
        const int tm = it % p.tilesM, r0 = it / p.tilesM;
        const int ks = r0 % p.S, r = r0 / p.S;
        ik0 = ks * p.kchunk;
        ink = min(p.kchunk, nk_all - ik0);
        const int tn = r % p.tilesN, z = r / p.tilesN;
        const int zo = z / p.Zi, zi = z - zo * p.Zi;
        int mcol = tm * BM + acol;
        if (mcol > (int)p.lda - 4) mcol = (int)p.lda - 4;             // stay inside the row (columns >= M are don't-care)
        iAb = p.At + zo * p.sAo + zi * p.sAi + mcol;
        iBb = p.B + zo * p.sBo + zi * p.sBi + tn * BN + bcol;
    };
    auto issue_next = [&]() {
        float* st = lds + (gi % NST) * STAGE;
        const int k0 = (ik0 + ikt) * BK;
#pragma unroll
        for (int h = 0; h < PW; ++h) {
            const int q = wave + NW * h;                             // wave-uniform
            const float* src;
            float* dst;
            if (q < PA) {
                const int kr = k0 + (256 / AW) * q + arow_l;         // A rows < ceil16(K) are readable by contract
                src = iAb + (long)kr * p.lda;
                dst = st + q * 256;
            } else if (q < PT) {
                const int qb = q - PA;
                int kr = k0 + (256 / BW) * qb + brow_l;
                if (kr >= p.K) kr = p.K - 1;                         // finite filler; the matching A rows are zero
                src = iBb + (long)kr * p.ldb;
                dst = st + BK * AW + qb * 256;
            } else {
                src = iAb;                                           // keeps every wave at PW ops per slab (uniform vmcnt)
                dst = dummy;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        }
        ++gi;
        if (++ikt == ink) {
            ikt = 0;
            it += G;
            if (it < ntiles) icursor();
        }
    };

#ifdef X3_TRACE
    if (p.trace && tid == 0) p.trace[(long)blockIdx.x * 64 + 0] = wall_clock64();
#endif
    icursor();
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (it < ntiles) issue_next();
#ifdef X3_TRACE
    if (p.trace && tid == 0) p.trace[(long)blockIdx.x * 64 + 1] = wall_clock64();
#endif

    const EpiP& ep = p.ep;
    int gc = 0, landed = 0;            // slabs consumed; slabs known to be in LDS (everything issued before the last vmcnt(0))
    for (int t = vb; t < ntiles; t += G) {
        const int tm = t % p.tilesM, r0_ = t / p.tilesM;
        const int ks = r0_ % p.S, r_ = r0_ / p.S;
        const int nk = min(p.kchunk, nk_all - ks * p.kchunk);
        const int tn = r_ % p.tilesN, z = r_ / p.tilesN;
        const int zo = z / p.Zi, zi = z - zo * p.Zi;
        const int m0 = tm * BM, n0 = tn * BN;
        const bool has_next = t + G < ntiles;
        f32x16 acc[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            // slab gc has landed when only the gi - gc - 1 younger slabs of this wave are still in flight (the first slabs of a
            // tile that follows another one were waited for before that tile's epilogue)
            if (gc >= landed) wait_slabs<PW, NST>(gi - gc - 1);
            __builtin_amdgcn_s_barrier();      // every wave's pieces of this slab are in LDS; the previous slab is no longer read
#if defined(X3_TRACE) && X3_TRACE >= 2
            if (kt < 48) X3_STAMP(8 + kt);
#endif
            if (it < ntiles) issue_next();     // refill the stage the previous slab occupied
            const float* As = lds + (gc % NST) * STAGE;
            const float* Bs = As + BK * AW;
            ++gc;
#ifndef X3_NO_COMPUTE
            // ---- raw fp32 fragments: lane (lm, kg) takes k = 8kg..8kg+7
            float ar[TM][8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i) ar[i][kk] = As[(8 * kg + kk) * AW + (wm * TM + i) * 32 + lm];
            bf16x8 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) split8(ar[i], ah[i], al[i]);
            // B: columns 4lm..4lm+3 in two 8-byte halves (interleaved tiles 0,1 then 2,3): half the live registers of one
            // 16-byte read per k, same LDS bytes and rate
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                f32x2 br[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    br[kk] = *reinterpret_cast<const f32x2*>(Bs + (8 * kg + kk) * BW + wn * 128 + 4 * lm + 2 * hq);
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * hq + qq;
                    float bt[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) bt[kk] = br[kk][qq];
                    bf16x8 bh, bl;
                    split8(bt, bh, bl);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if (!p.ep.one) {                // (RCOT_PREC_BF16X1: the hi * hi product alone)
                            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][q], 0, 0, 0);
                            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][q], 0, 0, 0);
                        }
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][q], 0, 0, 0);
                    }
                }
            }
#else
            (void)As; (void)Bs;
#endif
        }
        // The next tile's first slabs were requested one and two slabs ago: they have to be in
        // LDS before this tile's stores join the queue (stores and loads share the vm counter; after this wait the
        // counted waits of the next tile only ever have to cover loads that are older than its own).
        X3_STAMP(2);
        if (has_next) {
            wait_vm<0>();
            landed = gi;
        }
        X3_STAMP(3);

        // ---- epilogue: out = alpha*acc + rowscale[m]*R + beta*C_old, 16-byte stores straight from the accumulators
        // (lane holds columns 4lm..4lm+3 of rows (r&3) + 8(r>>2) + 4kg of every 32-row tile)
        if (p.S > 1) {
            // a piece of a split reduction: raw partial sums (times the per-pixel LN scale; the LN constants ride on piece 0)
            // to slab (z, ks) of the workspace [z][ks][M][N]; x3_reduce_kernel finishes the epilogue
            float* Sb_ = p.ws + ((long)z * p.S + ks) * p.M * p.N;
            const int ncol_ = n0 + wn * 128 + 4 * lm;
            f32x4 rs4_ = {1.f, 1.f, 1.f, 1.f}, murs4_ = {0.f, 0.f, 0.f, 0.f};
            if (LNP) {
                const long n = (long)zo * p.sLN + ncol_;
                rs4_ = *reinterpret_cast<const f32x4*>(p.rs + n);
                murs4_ = *reinterpret_cast<const f32x4*>(p.mu + n) * rs4_;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + (wm * TM + i) * 32 + 4 * kg;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m >= p.M) continue;
                    f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if (LNP) {
                        v = v * rs4_;
                        if (ks == 0) v = v - murs4_ * p.c1[m] + p.c2[m];
                    }
                    *reinterpret_cast<f32x4*>(Sb_ + (long)m * p.N + ncol_) = v;
                }
            }
            continue;
        }
        float* Cb = ep.C + zo * ep.sCo + zi * ep.sCi;
        const float* Rb = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
        const float* Sb = ep.rowscale ? ep.rowscale + zo * ep.sSo + zi * ep.sSi : nullptr;
        const float* Ad = Rb ? Rb : (ep.beta != 0.f ? Cb : nullptr);     // the addend requested up front
        const long ldad = Rb ? ep.ldr : ep.ldc;
        const bool both = Rb && ep.beta != 0.f;
        const int ncol = n0 + wn * 128 + 4 * lm;
        f32x4 rs4 = {1.f, 1.f, 1.f, 1.f}, murs4 = {0.f, 0.f, 0.f, 0.f};
        if (LNP) {                                                        // per-pixel statistics of this lane's 4 columns
            const long n = (long)zo * p.sLN + ncol;
            rs4 = *reinterpret_cast<const f32x4*>(p.rs + n);
            murs4 = *reinterpret_cast<const f32x4*>(p.mu + n) * rs4;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * kg;
#pragma unroll
            for (int hf = 0; hf < 4; ++hf) {                              // four rows at a time (register budget)
                f32x4 q[4];
                float sc[4];
                if (Ad) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = hf * 4 + r4, m = mb + (r & 3) + 8 * (r >> 2);
                        sc[r4] = Rb ? ((Sb && m < p.M) ? Sb[m] : 1.f) : ep.beta;
                        q[r4] = m < p.M ? *reinterpret_cast<const f32x4*>(Ad + (long)m * ldad + ncol) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = hf * 4 + r4, m = mb + (r & 3) + 8 * (r >> 2);
                    if (m >= p.M) continue;
                    f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if (LNP) v = v * rs4 - murs4 * p.c1[m] + p.c2[m];    // LN fold (file header)
                    v *= ep.alpha;
                    if (Ad) v += q[r4] * sc[r4];
                    float* dst = Cb + (long)m * ep.ldc + ncol;
                    if (both) v += *reinterpret_cast<const f32x4*>(dst) * ep.beta;
#ifndef X3_NO_STORE
                    *reinterpret_cast<f32x4*>(dst) = v;
#else
                    if (v[0] == 1.2345f) *reinterpret_cast<f32x4*>(dst) = v;
#endif
                }
            }
        }
        X3_STAMP(4);
#ifdef X3_TRACE
        if (t == vb) { wait_vm<0>(); X3_STAMP(5); }
#endif
    }
}

// C[z] = alpha * sum_ks slab[z][ks] + rowscale * R + beta * C   (fixed summation order; 16 bytes per thread)
__global__ __launch_bounds__(256) void x3_reduce_kernel(const float* __restrict__ ws, int S, int M, int N4, int Zi, EpiP ep) {
    const long per = (long)M * N4;
    const int z = blockIdx.y, zo = z / Zi, zi = z - zo * Zi;
    const float* w = ws + (long)z * S * per * 4;
    float* Cz = ep.C + zo * ep.sCo + zi * ep.sCi;
    const float* Rz = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
    const float* Sz = ep.rowscale ? ep.rowscale + zo * ep.sSo + zi * ep.sSi : nullptr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N4), n4 = (int)(i - (long)m * N4);
        f32x4 a = *reinterpret_cast<const f32x4*>(w + i * 4);
        for (int s = 1; s < S; ++s) a += *reinterpret_cast<const f32x4*>(w + ((long)s * per + i) * 4);
        a *= ep.alpha;
        if (Rz) a += *reinterpret_cast<const f32x4*>(Rz + (long)m * ep.ldr + 4 * n4) * (Sz ? Sz[m] : 1.f);
        float* dst = Cz + (long)m * ep.ldc + 4 * n4;
        if (ep.beta != 0.f) a += *reinterpret_cast<const f32x4*>(dst) * ep.beta;
        *reinterpret_cast<f32x4*>(dst) = a;
    }
}

template <int TM, int WM, int WN, bool LNP, int NST>
int launch_x3_k(const X3P& p, int grid, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 128 * WN, AW = BM <= 64 ? 64 : 128;
    const size_t smem = sizeof(float) * ((size_t)NST * BK * (AW + BN) + 256);
    static bool once = (hipFuncSetAttribute((const void*)gemm_x3_kernel<TM, WM, WN, LNP, NST>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);
    (void)once;
    note_kernel("gemm_x3_kernel<%d, %d, %d, %s, %d>", TM, WM, WN, tf(LNP), NST);
    RCOT_LAUNCH((gemm_x3_kernel<TM, WM, WN, LNP, NST>), dim3(grid), dim3(64 * WM * WN), smem, st, p);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

template <int TM, int WM, int WN>
int launch_x3(X3P p, bool ln, int Z, hipStream_t st, size_t ws_bytes) {
    constexpr int BM = 32 * TM * WM, BN = 128 * WN;
    // lone workgroup per CU: most of the LDS as ring, within what a 6-bit vmcnt can count ((NST - 1) * ops-per-slab <= 63)
    constexpr int AW_ = BM <= 64 ? 64 : 128, PW_ = (BK * (AW_ + BN) / 256 + WM * WN - 1) / (WM * WN);
    constexpr int NST_LDS = BN == 256 ? 6 : 8, NST_VM = 1 + 63 / PW_;
    constexpr int NST_DEEP = NST_LDS < NST_VM ? NST_LDS : NST_VM;
#ifdef X3_PER_CU
    constexpr int PER_CU = X3_PER_CU;
#else
    constexpr int PER_CU = TM == 1 ? 3 : 2;                            // resident workgroups per CU (registers / LDS)
#endif
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = p.N / BN;
    // split-K when the output tiles alone cannot fill the chip and the reduction is long
    const int nk = cdiv(p.K, BK);
    const int base = p.tilesM * p.tilesN * Z;
    int S = 1;
    if (p.ws && base < 256 && nk >= 16) {
        // two workgroups per CU: a lone wavefront per SIMD runs its slab chain (LDS reads -> split -> MFMA -> barrier) at
        // ~0.8 us per slab whatever the ring depth, a second workgroup fills those gaps (measured: 512 pieces beat 256
        // despite the extra slab traffic: 31.6 vs 36.7 us on C=384 <- 2042 at 16x16)
        S = cdiv(512, base);
        if (S > nk / 8) S = nk / 8;                                   // >= 8 slabs per piece
        while (S > 1 && (size_t)S * Z * p.M * p.N * sizeof(float) > ws_bytes) --S;
        if (S < 1) S = 1;
    }
    p.kchunk = cdiv(nk, S);
    p.S = cdiv(nk, p.kchunk);
    if (p.S > 1 && nk - (p.S - 1) * p.kchunk < 2) {                   // the slab ring needs two slabs in every piece
        p.kchunk = cdiv(nk, p.S - 1);
        p.S = cdiv(nk, p.kchunk);
    }
    p.ntiles = base * p.S;
    // every workgroup gets the same number of tiles (+-1): grid = tiles / rounds
    const int rounds = cdiv(p.ntiles, PER_CU * 256);
    const int grid = cdiv(p.ntiles, rounds);
    int rc;
    const bool deep = grid <= 256;
    if (ln) rc = deep ? launch_x3_k<TM, WM, WN, true, NST_DEEP>(p, grid, st) : launch_x3_k<TM, WM, WN, true, NST_STREAM>(p, grid, st);
    else rc = deep ? launch_x3_k<TM, WM, WN, false, NST_DEEP>(p, grid, st) : launch_x3_k<TM, WM, WN, false, NST_STREAM>(p, grid, st);
    if (rc != RCOT_OK) return rc;
    if (p.S > 1) {
        const long per = (long)p.M * (p.N / 4);
        long nb = (per + 255) / 256;
        if (nb > 2048) nb = 2048;
        RCOT_LAUNCH(x3_reduce_kernel, dim3((int)nb, Z), dim3(256), 0, st, p.ws, p.S, p.M, p.N / 4, p.Zi, p.ep);
        RCOT_LAUNCH_CHECK();
    }
    return RCOT_OK;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace rcot_x3

namespace rcot {

// Returns RCOT_OK after launching, or -100 when the shape is not eligible (the caller then uses the fp32 kernels).
int try_gemm_kmajor_x3(const float* At, long lda, long sAo, long sAi, const float* Bm, long ldb, long sBo, long sBi,
                       const EpiP& ep, const float* ln_mu, const float* ln_rs, long sLN, const float* ln_c1,
                       const float* ln_c2, int Zo, int Zi, int M, int N, int K, float* ws, size_t ws_bytes, hipStream_t st) {
    using namespace rcot_x3;
    if ((N % 128) || K < 17) return -100;          // the slab ring needs at least two slabs per tile
    const bool ln = ln_mu != nullptr;
    if (ln && ((sLN & 3) || !al16(ln_mu) || !al16(ln_rs) || !ln_c1 || !ln_c2)) return -100;
    X3P p{};
    p.M = M; p.N = N; p.K = K; p.Zi = Zi;
    p.At = At; p.lda = lda; p.sAo = sAo; p.sAi = sAi;
    p.B = Bm; p.ldb = ldb; p.sBo = sBo; p.sBi = sBi;
    p.mu = ln_mu; p.rs = ln_rs; p.sLN = sLN; p.c1 = ln_c1; p.c2 = ln_c2;
    p.ep = ep;
    p.ws = ws;
#ifdef X3_TRACE
    p.trace = g_x3_trace;
#endif
    const int Z = Zo * Zi;
    const bool wide = (N % 256) == 0;
    // row tiling with the least padding: 128-row (2 x 64), 96-row (3 x 32) or 64-row workgroup tiles
    const long pad128 = (long)cdiv(M, 128) * 128, pad96 = (long)cdiv(M, 96) * 96, pad64 = (long)cdiv(M, 64) * 64;
    const long cols = (long)(N / 128) * Z;
    if (M <= 64 || (pad64 < pad96 && pad64 < pad128)) {
        if (wide && (long)cdiv(M, 64) * cols / 2 >= 512) return launch_x3<2, 1, 2>(p, ln, Z, st, ws_bytes);
        return launch_x3<2, 1, 1>(p, ln, Z, st, ws_bytes);
    }
    if (pad96 < pad128) {
        return launch_x3<1, 3, 1>(p, ln, Z, st, ws_bytes);
    }
    if (wide && (long)cdiv(M, 128) * cols / 2 >= 512) return launch_x3<2, 2, 2>(p, ln, Z, st, ws_bytes);
    return launch_x3<1, 4, 1>(p, ln, Z, st, ws_bytes);
}

}  // namespace rcot
