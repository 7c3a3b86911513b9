// Device side of the LDS-DMA pixel-reduction GEMM (gemm_nt_glds.hip, file header there): parameter block, fragment helpers and
// the workgroup body.  A header because two launch forms use the body: gemm_nt_kernel (gemm_nt_glds.hip) and the paired
// data-gradient + weight-gradient launch of gemm_x3w.hip (x3p_nt_pair_kernel), where workgroups of BOTH products of one
// incoming gradient share a grid.
#pragma once
#include "gemm_core.h"

using namespace rcot;

namespace rcot_nt {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef NT_NST
#define NT_NST 3
#endif
constexpr int NST = NT_NST;            // ring stages = slabs requested ahead (3: two workgroups per CU; 4 measured the same, 5 = one workgroup per CU 10-40 % slower)
constexpr int IMG = 128 * BK;                 // floats per operand image (128 rows x 16 k)
constexpr int STAGE = 2 * IMG + 4 * 64;       // + per-wave LN stats (mu16 | rs16 | dup)

struct NTP {
    int M, N, K, Zi, S, kchunk, tilesM, tilesN, ldws;
    const float* A; long lda, sAo, sAi;
    const float* B; long ldb, sBo, sBi;
    int Kb; long sAk, sBk;                    // batch folded into K (Kb = per-image K, 0 = off)
    const float* mu; const float* rs; long sLNb;
    const float* lnw; const float* lnb;
    float* ws;
    // conv_taps > 0 (rcot_conv_pcm_wgrad: the 3x3 weight gradient over padded channel-major planes, csrc/conv_pcm.hip): B row
    // n = (channel n / 9, tap n % 9) starts at  B + channel * ldb + (ky - 1) * conv_wp + (kx - 1)  — the same row shifted by
    // the tap (any 4-byte alignment is fine for LDS-DMA) — so C[m][n] is the OIHW weight gradient itself
    int conv_taps, conv_wp;
    int one;                                  // X3 kernels: the single product hi * hi (RCOT_PREC_BF16X1)
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most y slabs (NPW vm operations each) of this wave are outstanding, 0 <= y <= 3
template <int NPW> __device__ __forceinline__ void wait_slabs(int y) {
    if (y <= 0) wait_vm<0>();
    else if (y == 1) wait_vm<NPW>();
    else if (y == 2) wait_vm<2 * NPW>();
    else wait_vm<3 * NPW>();
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// Fragment reads are issued as inline asm: the compiler then neither places an "LDS-DMA may alias" s_waitcnt vmcnt(0)
// in front of them (which would serialise the DMA ring) nor sinks each read next to its consumer.  The value of a
// read may only be used after the matching wait_lgkm<>() + pin(): pin() is the data-dependence fence.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lds_read64(uint32_t byte_addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 lds_read128(uint32_t byte_addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }
// (inline asm for the same reason as the reads: no compiler-placed vmcnt(0) in front of it)
__device__ __forceinline__ void lds_write128(uint32_t byte_addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}
// bf16x3 split (see gemm_x3.hip): eight consecutive-k fp32 values -> hi / lo bf16x8 MFMA operands
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2 v = {x[q], x[q + 1]};
        const bf16x2 h = __builtin_convertvector(v, bf16x2);
        const f32x2 r = v - __builtin_convertvector(h, f32x2);
        const bf16x2 l = __builtin_convertvector(r, bf16x2);
        hi[q] = h[0]; hi[q + 1] = h[1];
        lo[q] = l[0]; lo[q + 1] = l[1];
    }
}

// three-term split (bf16x6, fp32-class: 24 mantissa bits together): hi = rne(x), mid = rne(x - hi), lo = rne(x - hi - mid)
__device__ __forceinline__ void split8_3(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2 v = {x[q], x[q + 1]};
        const bf16x2 h = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        hi[q] = h[0]; hi[q + 1] = h[1];
        mid[q] = m[0]; mid[q + 1] = m[1];
        lo[q] = l[0]; lo[q + 1] = l[1];
    }
}

// X3 = false: exact fp32 (v_mfma_f32_32x32x2_f32).  X3 = true: bf16x3 split products (three v_mfma_f32_32x32x16_bf16
// per 16-pixel slab and tile pair, operands split when the fragment leaves LDS; same ring, swizzle and epilogue).  X3 && X6: bf16x6 —
// both operands split into THREE terms, the six products of order <= 2^-16 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; smallest
// first), as accurate as the exact-fp32 form at 6 x 32 instead of 8 x 64 MFMA cycles per 32x32x16 block.
// bx / bz: the workgroup's position in a (tiles * splits, 1, Z) grid of this product (the kernel passes blockIdx; the paired launch
// of gemm_x3w.hip passes its own numbering)
//
// COOP (X3 kernels, round 5 session 4): the operand that ALL FOUR wavefronts of the workgroup consume — B when the waves are stacked
// along m (COOP = 1: WN == 1, the 128 x 96 / 128 x 64 weight-gradient and Gram tiles), A when they are stacked along n (COOP = 2:
// WM == 1) — is normalised (B) and split ONCE per workgroup instead of once per wavefront: every wave transforms the rows of the DMA
// pieces it requested itself (its own vmcnt wait covers them: no extra barrier), leaves the bf16 terms as ready MFMA fragments
// [buffer k & 1][term][k group][row] (16 bytes per lane, lane-linear: conflict-free to write and to read) behind the ring, and the
// slab barrier that publishes the raw slab publishes them too.  Same operations on the same values as the per-wave split: same bits.
// Per slab and wave the 128 x 96 bf16x6 tile did 32 elements x (split + LayerNorm) = ~230 VALU instructions next to 18 MFMAs
// (the MFMAs of a SIMD wait for its VALU instructions); now 8 + 8 elements.
template <int TM, int TN, int WM, int WN, bool LNP, bool X3, bool X6 = false, int COOP = 0>
__device__ __forceinline__ void nt_body(const NTP& p, const int bx, const int bz) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr bool CA = COOP == 2, CB = COOP == 1;
    constexpr int NTERM = X6 ? 3 : 2;
    constexpr uint32_t FBSZ = NTERM * 2 * 128 * 16;                // bytes of one fragment buffer: [term][k group][128 rows] x 16
    static_assert(!COOP || X3, "cooperative split: split kernels only");
    static_assert((!CB || (WN == 1 && TM == 1)) && (!CA || (WM == 1 && TN == 1)), "cooperative operand = the one all four waves share");
    constexpr int PA = BM <= 64 ? 1 : 2, PB = BN <= 64 ? 1 : 2;   // 16-row DMA pieces per wave and operand (64 or 128 image rows)
    constexpr int NPW = PA + PB + (LNP ? 1 : 0);                   // DMA ops per wave per slab
    constexpr int NRD = TM + TN + (LNP ? 2 : 0);   // LDS reads per k-quad
    static_assert(WM * WN == 4 && BM <= 128 && BN <= 128, "tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // grid.x = tiles * splits, remapped so that one XCD works on consecutive (split, tile) pairs: the M/N tiles of ONE
    // K-range then share that XCD's L2 (with the splits in grid.z the tiles of a K-range sat on different XCDs and the
    // shared operand was fetched once per XCD: 514 MiB vs 303 MiB of operands on the 510x96 weight gradient)
    const int nblk = p.tilesM * p.tilesN;
    const int q = xcd_remap(bx, nblk * p.S);
    const int s = q / nblk, bid = q - s * nblk;
    const int tm = bid % p.tilesM, tn = bid / p.tilesM;
    const int z = bz;
    const int zs = z * p.S + s;
    const int zo = z / p.Zi, zi = z - zo * p.Zi;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = s * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg) / BK;        // K, kchunk are multiples of 16

    // ---- DMA addressing.  Piece q (1 KiB) of an image = rows 16q..16q+15; lane -> (row, physical chunk)
    const int prow = lane >> 2, pc = lane & 3;
    const float* arow[2];
    const float* brow[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 16 * (wave + 4 * h) + prow;                 // row of the 128-row image
        const int kq = pc ^ ((row >> 2) & 3);                       // logical chunk this lane fetches
        const int gm = min(m0 + row, p.M - 1), gn = min(n0 + row, p.N - 1);   // rows past the edge: any finite data
        arow[h] = p.A + zo * p.sAo + zi * p.sAi + (long)gm * p.lda + kq * 4;
        brow[h] = p.B + zo * p.sBo + zi * p.sBi + (long)gn * p.ldb + kq * 4;
        if (p.conv_taps) {
            const int ch = gn / p.conv_taps, tp = gn - ch * p.conv_taps, ky = tp / 3, kx = tp - 3 * ky;
            brow[h] = p.B + (long)ch * p.ldb + ((ky - 1) * p.conv_wp + (kx - 1)) + kq * 4;
        }
    }
    float lw_[TN], lb_[TN];
    if (LNP) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n0 + (wn * TN + j) * 32 + (lane & 31), p.N - 1);
            lw_[j] = p.lnw[n];
            lb_[j] = p.lnb[n];
            asm volatile("" ::"v"(lw_[j]), "v"(lb_[j]));           // retire ordinary loads before the DMA pipeline starts
        }
    }

    auto issue = [&](int kt) {
        float* st = lds + (kt % NST) * STAGE;
        int k0 = kbeg + kt * BK;
        long ka = k0, kb = k0, kl = k0;
        if (p.Kb) {
            const int b = k0 / p.Kb, kk = k0 - b * p.Kb;
            ka = (long)b * p.sAk + kk;
            kb = (long)b * p.sBk + kk;
            kl = (long)b * p.sLNb + kk;
        }
#pragma unroll
        for (int h = 0; h < PA; ++h)
            __builtin_amdgcn_global_load_lds((gptr_t)(arow[h] + ka), (lptr_t)(st + (wave + 4 * h) * 256), 16, 0, 0);
#pragma unroll
        for (int h = 0; h < PB; ++h)
            __builtin_amdgcn_global_load_lds((gptr_t)(brow[h] + kb), (lptr_t)(st + IMG + (wave + 4 * h) * 256), 16, 0, 0);
        if (LNP) {
            const float* src = ((lane & 16) ? p.rs : p.mu) + kl + (lane & 15);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + 2 * IMG + wave * 64), 4, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (nk > i) issue(i);

    // ---- fragment addressing (LDS byte addresses of stage 0; the swizzle is an XOR of address bits 4-5).
    // Of every 16-byte k-quad the lower half-wave consumes k = 0,1 and the upper half-wave k = 2,3 (one 8-byte read
    // each, no selects): MFMA step s of the quad multiplies k = s (lanes 0-31) and k = 2+s (lanes 32-63), for A and B alike.
    const int lm = lane & 31;
    const uint32_t hi8 = lane >= 32 ? 8u : 0u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)lds;
    uint32_t aad[TM], bad[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lm;
        aad[i] = lds0 + row * (BK * 4) + (((row >> 2) & 3) << 4) + hi8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + lm;
        bad[j] = lds0 + IMG * 4 + row * (BK * 4) + (((row >> 2) & 3) << 4) + hi8;
    }
    const uint32_t lad = lds0 + (2 * IMG + wave * 64) * 4 + hi8;

    if constexpr (X3) {
        // Lane (row = lane & 31 of a 32-row tile, kg = lane >> 5) consumes k = 8kg..8kg+7 of the slab: logical 16-byte chunks
        // 2kg and 2kg+1 of its row (two ds_read_b128 through the same XOR swizzle: conflict-free in the b128 lane groups).
        const uint32_t kgo = lane >= 32 ? 32u : 0u;                // chunk 2kg -> address bit 5
        uint32_t aad3[TM], bad3[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 32 + lm;
            aad3[i] = lds0 + row * (BK * 4) + ((((row >> 2) & 3) << 4) ^ kgo);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = (wn * TN + j) * 32 + lm;
            bad3[j] = lds0 + IMG * 4 + row * (BK * 4) + ((((row >> 2) & 3) << 4) ^ kgo);
        }
        const uint32_t lad3 = lds0 + (2 * IMG + wave * 64) * 4 + kgo;     // mu[8kg..], rs at +64 bytes
        // ---- cooperative operand (COOP): this lane's task = (row of one of the wave's OWN DMA pieces, k group lane >> 5)
        const int trow = 16 * (wave + 4 * ((lane >> 4) & 1)) + (lane & 15);         // row of the 128-row image
        const uint32_t fb0 = lds0 + (uint32_t)(NST * STAGE * 4);
        const uint32_t xraw = lds0 + (CB ? IMG * 4 : 0) + trow * (BK * 4) + ((((trow >> 2) & 3) << 4) ^ kgo);
        const uint32_t xfb = fb0 + (uint32_t)(((lane >> 5) * 128 + trow) * 16);
        uint32_t afb[TM], bfb[TN];                                                  // fragment addresses of the consumer side
#pragma unroll
        for (int i = 0; i < TM; ++i) afb[i] = fb0 + (uint32_t)(((lane >> 5) * 128 + (wm * TM + i) * 32 + lm) * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfb[j] = fb0 + (uint32_t)(((lane >> 5) * 128 + (wn * TN + j) * 32 + lm) * 16);
        float lwx = 0.f, lbx = 0.f;
        if (LNP && CB) {
            const int n = min(n0 + trow, p.N - 1);
            lwx = p.lnw[n];
            lbx = p.lnb[n];
            asm volatile("" ::"v"(lwx), "v"(lbx));
        }
        f32x4 xr[2], xm[2], xs[2];
        auto xf_read = [&](int kt) {                               // raw rows of the wave's own pieces of slab kt (+ its LN statistics)
            const uint32_t so = (uint32_t)((kt % NST) * (STAGE * 4));
            xr[0] = lds_read128(xraw + so);
            xr[1] = lds_read128((xraw + so) ^ 16u);
            if (LNP && CB) {
                xm[0] = lds_read128(lad3 + so);
                xm[1] = lds_read128(lad3 + so + 16);
                xs[0] = lds_read128(lad3 + so + 64);
                xs[1] = lds_read128(lad3 + so + 80);
            }
        };
        auto xf_write = [&](int buf) {                             // normalise, split, leave the terms as fragments (buf compile-time)
            pin(xr[0]); pin(xr[1]);
            f32x4 b0 = xr[0], b1 = xr[1];
            if (LNP && CB) {
                pin(xm[0]); pin(xm[1]); pin(xs[0]); pin(xs[1]);
                b0 = (b0 - xm[0]) * xs[0] * lwx + lbx;
                b1 = (b1 - xm[1]) * xs[1] * lwx + lbx;
            }
            bf16x8 h, l;
            const uint32_t o = xfb + (uint32_t)buf * FBSZ;
            if constexpr (X6) {
                bf16x8 m;
                split8_3(b0, b1, h, m, l);
                lds_write128(o, __builtin_bit_cast(f32x4, h));
                lds_write128(o + 4096u, __builtin_bit_cast(f32x4, m));
                lds_write128(o + 8192u, __builtin_bit_cast(f32x4, l));
            } else {
                split8(b0, b1, h, l);
                lds_write128(o, __builtin_bit_cast(f32x4, h));
                lds_write128(o + 4096u, __builtin_bit_cast(f32x4, l));
            }
        };
        f32x4 ra[2][TM][2], rb[2][TN][2], rm[2][2], rr[2][2];
        f32x4 raf[2][CA ? TM : 1][NTERM], rbf[2][CB ? TN : 1][NTERM];             // COOP: ready fragments (hi | mid | lo)
        auto rd3 = [&](int kt, int buf) {                          // buf is compile-time at every call site (= kt & 1)
            const uint32_t so = (uint32_t)((kt % NST) * (STAGE * 4));
            const uint32_t fo = (uint32_t)buf * FBSZ;
            if constexpr (CA) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < NTERM; ++t) raf[buf][i][t] = lds_read128(afb[i] + fo + 4096u * t);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ra[buf][i][0] = lds_read128(aad3[i] + so);
                    ra[buf][i][1] = lds_read128((aad3[i] + so) ^ 16u);
                }
            }
            if constexpr (CB) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t = 0; t < NTERM; ++t) rbf[buf][j][t] = lds_read128(bfb[j] + fo + 4096u * t);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    rb[buf][j][0] = lds_read128(bad3[j] + so);
                    rb[buf][j][1] = lds_read128((bad3[j] + so) ^ 16u);
                }
                if (LNP) {
                    rm[buf][0] = lds_read128(lad3 + so);
                    rm[buf][1] = lds_read128(lad3 + so + 16);
                    rr[buf][0] = lds_read128(lad3 + so + 64);
                    rr[buf][1] = lds_read128(lad3 + so + 80);
                }
            }
        };
        auto mm3 = [&](int buf) {
#ifdef NT_NO_SPLIT        // tuning build (-DNT_NO_SPLIT, results are garbage): MFMAs on unsplit, bit-cast fragments = the ring, the barriers
                          // and the MFMAs without the split / LayerNorm VALU work: 90 / 65.5 / 49.5 us against 122.6 / 93.3 / 52.7
            if constexpr (!COOP) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    pin(rb[buf][j][0]); pin(rb[buf][j][1]);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, rb[buf][j][0]), bl = __builtin_bit_cast(bf16x8, rb[buf][j][1]);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        pin(ra[buf][i][0]); pin(ra[buf][i][1]);
                        const bf16x8 ah_ = __builtin_bit_cast(bf16x8, ra[buf][i][0]), al_ = __builtin_bit_cast(bf16x8, ra[buf][i][1]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh, acc[i][j], 0, 0, 0);
                    }
                }
                return;
            }
#endif
            bf16x8 ah[TM], al[TM], am[X6 ? TM : 1];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (CA) {
#pragma unroll
                    for (int t = 0; t < NTERM; ++t) pin(raf[buf][i][t]);
                    ah[i] = __builtin_bit_cast(bf16x8, raf[buf][i][0]);
                    if constexpr (X6) am[i] = __builtin_bit_cast(bf16x8, raf[buf][i][1]);
                    al[i] = __builtin_bit_cast(bf16x8, raf[buf][i][NTERM - 1]);
                } else {
                    pin(ra[buf][i][0]);
                    pin(ra[buf][i][1]);
                    if constexpr (X6) split8_3(ra[buf][i][0], ra[buf][i][1], ah[i], am[i], al[i]);
                    else split8(ra[buf][i][0], ra[buf][i][1], ah[i], al[i]);
                }
            }
            if (LNP && !CB) {
                pin(rm[buf][0]); pin(rm[buf][1]); pin(rr[buf][0]); pin(rr[buf][1]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 bh, bl, bm;
                if constexpr (CB) {
#pragma unroll
                    for (int t = 0; t < NTERM; ++t) pin(rbf[buf][j][t]);
                    bh = __builtin_bit_cast(bf16x8, rbf[buf][j][0]);
                    bm = __builtin_bit_cast(bf16x8, rbf[buf][j][X6 ? 1 : 0]);
                    bl = __builtin_bit_cast(bf16x8, rbf[buf][j][NTERM - 1]);
                } else {
                    pin(rb[buf][j][0]);
                    pin(rb[buf][j][1]);
                    f32x4 b0 = rb[buf][j][0], b1 = rb[buf][j][1];
                    if (LNP) {
                        b0 = (b0 - rm[buf][0]) * rr[buf][0] * lw_[j] + lb_[j];
                        b1 = (b1 - rm[buf][1]) * rr[buf][1] * lw_[j] + lb_[j];
                    }
                    if constexpr (X6) split8_3(b0, b1, bh, bm, bl);
                    else { split8(b0, b1, bh, bl); bm = bh; }
                }
                if constexpr (X6) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                    }
                    continue;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (!p.one) {                       // (RCOT_PREC_BF16X1: the hi * hi product alone)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        };
        wait_slabs<NPW>(min(nk, NST) - 1);      // slab 0 landed
        if constexpr (COOP) {
            if (nk > 0) {
                xf_read(0);
                wait_lgkm<0>();
                xf_write(0);
                wait_lgkm<0>();
            }
        }
        __builtin_amdgcn_s_barrier();
        if (nk > 0) rd3(0, 0);
        // two slabs per trip so that the raw-fragment buffer index stays compile-time
        for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = kt + u;
                if (k >= nk) break;
                wait_lgkm<0>();                        // slab k's fragments are in registers (buffer u)
                if (k + 1 < nk) {
                    wait_slabs<NPW>(min(k + NST - 1, nk - 1) - (k + 1));   // slab k+1 landed: only the younger ones are outstanding
                    if constexpr (COOP) {              // this wave's share of slab k+1's cooperative operand, before the barrier publishes it
                        xf_read(k + 1);
                        wait_lgkm<0>();
                        xf_write(u ^ 1);
                        wait_lgkm<0>();
                    }
                    __builtin_amdgcn_s_barrier();      // all waves: slab k+1 visible, slab k's stage free
                    if (k + NST < nk) issue(k + NST);
                    rd3(k + 1, u ^ 1);                 // next slab's reads fly while this slab is split and multiplied
                }
                mm3(u);
            }
        }
    } else {
    f32x2 fa[2][TM], fb[2][TN], fm[2], fr[2];
    auto rd = [&](int kt, int kq, int buf) {                       // kq, buf are compile-time at every call site
        const uint32_t so = (uint32_t)((kt % NST) * (STAGE * 4));
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[buf][i] = lds_read64((aad[i] + so) ^ (uint32_t)(kq << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[buf][j] = lds_read64((bad[j] + so) ^ (uint32_t)(kq << 4));
        if (LNP) {
            fm[buf] = lds_read64(lad + so + kq * 16);
            fr[buf] = lds_read64(lad + so + 64 + kq * 16);
        }
    };
    auto mm = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) pin(fa[buf][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) pin(fb[buf][j]);
        float b0[TN], b1[TN];
        if (LNP) {
            pin(fm[buf]);
            pin(fr[buf]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x2 bb = fb[buf][j];
            if (LNP) {
                // the k pair as ONE packed operation per step (v_pk_add / v_pk_mul: same operations in the same order per element as the
                // scalar form, half the VALU instructions — which the MFMAs of this SIMD wait for)
                const f32x2 w2 = {lw_[j], lw_[j]}, c2 = {lb_[j], lb_[j]};
                bb = (bb - fm[buf]) * fr[buf] * w2 + c2;
            }
            b0[j] = bb.x;
            b1[j] = bb.y;
        }
        // the two k-steps of the quad as two sweeps over the accumulators: consecutive MFMAs never share one
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].x, b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].y, b1[j], acc[i][j], 0, 0, 0);
    };

    // ---- main loop: the reads of k-quad q+1 are in flight while the MFMAs of quad q execute; the slab barrier sits
    // in front of the LAST quad's MFMAs (all of this wave's reads of the slab are complete by then), and the DMA that
    // refills the stage is issued three slabs ahead.
    wait_slabs<NPW>(min(nk, NST) - 1);          // slab 0 landed
    __builtin_amdgcn_s_barrier();
    if (nk > 0) rd(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        rd(kt, 1, 1);
        wait_lgkm<NRD>();
        mm(0);
        rd(kt, 2, 0);
        wait_lgkm<NRD>();
        mm(1);
        rd(kt, 3, 1);
        wait_lgkm<NRD>();
        mm(0);
        __builtin_amdgcn_sched_barrier(0);     // keep the MFMAs of this quad above the wait
        wait_lgkm<0>();                        // every read of slab kt by this wave has completed
        if (kt + 1 < nk) {
            wait_slabs<NPW>(min(kt + NST - 1, nk - 1) - (kt + 1));   // slab kt+1 landed: only the younger ones are outstanding
            __builtin_amdgcn_s_barrier();      // all waves: slab kt+1 visible, slab kt's stage free
            if (kt + NST < nk) issue(kt + NST);
            rd(kt + 1, 0, 0);
        }
        mm(1);
    }

    }

    // ---- every split writes its slab (16-byte stores through the per-wave LDS transpose)
    __syncthreads();
    float* wsb = p.ws + (long)zs * p.M * p.ldws;
    epilogue_vec<TM, TN>(acc, lds + wave * 1024, wsb, p.ldws, nullptr, 0, nullptr, 1.f, 0.f, m0 + wm * TM * 32,
                         n0 + wn * TN * 32, p.M, p.ldws, lane);
}

}  // namespace rcot_nt

namespace rcot {
// (gemm_nt_glds.hip) parameter block, tile shape and split factor of one pixel-reduction product; -100: not eligible
int nt_configure(int M, int N, int K, int Zo, int Zi, const float* A, long lda, long sAo, long sAi, const float* B, long ldb,
                 long sBo, long sBi, int Kb, long sAk, long sBk, const float* mu, const float* rs, long sLNb, const float* lnw,
                 const float* lnb, float* ws, size_t ws_bytes, int prec, int conv_wp, rcot_nt::NTP* out, int* out_cfg, int slots = 640);
}  // namespace rcot
