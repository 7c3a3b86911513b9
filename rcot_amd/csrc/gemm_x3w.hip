// bf16x3 split-MFMA K-major projection GEMM for WEIGHT projections, wave-specialised ("x3p"):
//
//     C[z] (M x N) = A (M x K) * LN?(B[z]) (K x N)      A = a pre-split weight pack (rcot_pack_weight), B[k][n] fp32 in HBM
//
// Arithmetic, LayerNorm fold, split-K and epilogue contract are those of gemm_x3.hip (file header there), which stays the
// general kernel (batch-dependent A, N % 256 != 0, per-row scales).  This file exists because of what time stamps and
// counter-free A/B builds of that kernel showed (scripts/dbg/x3_trace.py, round 2):
//   * a 3-4 wave workgroup advanced one 16-row slab per 1.0-1.6 us while the MFMA work of a slab is 0.15-0.3 us: every
//     wavefront issued ~430 instructions per slab — ~150 address arithmetic for its DMA pieces, ~150 the bf16 split of
//     fragments that 2-4 waves of the workgroup each split again — at ~5 cycles per instruction;
//   * loads 40 us + compute 40 us + stores 40 us of a 510 <- 96 projection at 128x128 pixels ran BACK TO BACK (117 us):
//     loads and stores share one counter (vmcnt) on gfx9, so a wave that prefetches and stores drains at every tile
//     boundary, and an epilogue that fetches its row constants when it needs them pays an L2 round trip per row group.
// Hence: dedicated producer wavefronts (DMA + one split of B per workgroup tile), consumer wavefronts whose slab loop is
// 12 ds_read_b128 + 24 MFMAs, pre-split weights, three-instruction DMA pieces (s_add m0 / s_nop / global_load_lds with an
// SGPR base that advances per slab and a per-lane 32-bit offset that is constant per tile), and an epilogue whose
// constants are requested before the slab loop.  Measured against gemm_x3.hip (graph-replayed, cold operands, us):
// 510<-96 +LN 126 -> 107, 288<-96 +LN 87 -> 68, 144<-48 +LN 43 -> 31, 96<-288 62 -> 46 at 8 x 128x128 pixels;
// 576<-192 +LN at 8 x 32x32: 26 -> 21.
#include <cstdlib>
#include <type_traits>
#include "gemm_core.h"
#include "gemm_nt_body.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace rcot_x3w {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct P {
    int M, N, K, Zi, tilesM, tilesN, ntiles;
    int S, kchunk;                                    // split-K: S K-ranges of kchunk slabs, each written to its own slab of ws
    float* ws;
    const float* At; long lda, sAo, sAi;              // fp32 K-major A (APRE = false)
    const unsigned char* Apk; int MT;                 // pre-split A: [slab][MT][hi|lo][64 lanes][8 bf16]  (APRE = true)
    const float* B;  long ldb, sBo, sBi;
    const float* mu; const float* rs; long sLN;      // LN statistics per pixel (LNP)
    float* mu_out; float* rs_out;                     // STAT: the kernel computes them from the B rows it splits and writes them here
    const float* c1; const float* c2;                 // LN fold constants per output row (LNP)
    EpiP ep;
    // CONV (rcot_conv_pcm): the reduction runs over (tap, 16-channel group) slabs of a padded, channel-major input: the B rows of a
    // slab start at  B + group * 16 * ldb + tapoff[tap]  (any 4-byte alignment: LDS-DMA takes it at full rate,
    // scripts/micro/glds_unaligned.hip); the epilogue adds a per-row bias, applies LeakyReLU and stores 4-column groups at the
    // dense NCHW offsets colmap[] names (-1: padding positions, skipped).
    int conv_cslabs;                                  // slabs per tap (Ci / 16); 0 = plain projection
    int tapoff[16];
    const int* colmap;
    const float* cbias;
    float clrelu;
#ifdef X3_TRACE
    unsigned long long* trace;                        // debug build: 64 time stamps (100 MHz) per workgroup, first tile only
#endif
};

#ifdef X3_TRACE
static unsigned long long* g_x3w_trace = nullptr;
extern "C" int rcot_x3w_set_trace(void* q) { g_x3w_trace = (unsigned long long*)q; return 0; }
#define X3_STAMP(i) do { if (p.trace && threadIdx.x == 0 && t == vb) p.trace[(long)blockIdx.x * 64 + (i)] = wall_clock64(); } while (0)
#else
#define X3_STAMP(i) do {} while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// wait until at most y slabs (PW vm operations each) are still in flight; y is wave-uniform, 0 <= y < NST
template <int PW, int NST>
__device__ __forceinline__ void wait_slabs(int y) {
    static_assert((NST - 1) * PW <= 63, "vmcnt is a 6-bit field");
    if (y >= NST - 1) wait_vm<(NST - 1) * PW>();
    else if (y <= 0) wait_vm<0>();
    else if (y == 1) wait_vm<PW>();
    else if (NST > 3 && y == 2) wait_vm<(NST > 3 ? 2 : 0) * PW>();
    else wait_vm<(NST > 4 ? 3 : 0) * PW>();
}

// one 1-KiB DMA piece: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS byte address stage + OFF (+ 16 lane)
template <int OFF>
__device__ __forceinline__ void dma_piece(unsigned stage, const void* sbase, unsigned voff) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                 :: "s"(stage), "n"(OFF), "v"(voff), "s"(sbase) : "memory", "scc");
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {                      // v_cvt_pk_bf16_f32 (rne): a -> low half
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// Split of the B fragment.  x[kk] = columns 4lm..4lm+3 of row 8kg+kk.  For a row pair (kk, kk+1) and a column pair the
// residuals are formed with PACKED fp32 subtractions on registers that are adjacent as read (columns), while the bf16
// packing pairs the two rows of one column: no register moves.  hi[c], lo[c]: the MFMA operands of column tile c.
__device__ __forceinline__ void split_b(const f32x4 (&x)[8], u32x4 (&hi)[4], u32x4 (&lo)[4]) {
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
        const f32x4 x0 = x[2 * kp], x1 = x[2 * kp + 1];
        unsigned h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) h[c] = pk_bf16(x0[c], x1[c]);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const f32x2 a0 = {x0[2 * cp], x0[2 * cp + 1]}, a1 = {x1[2 * cp], x1[2 * cp + 1]};
            const u32x2 t0 = {h[2 * cp] << 16, h[2 * cp + 1] << 16};
            const u32x2 t1 = {h[2 * cp] & 0xffff0000u, h[2 * cp + 1] & 0xffff0000u};
            const f32x2 r0 = a0 - __builtin_bit_cast(f32x2, t0);                      // exact in fp32
            const f32x2 r1 = a1 - __builtin_bit_cast(f32x2, t1);
            lo[2 * cp][kp] = pk_bf16(r0[0], r1[0]);
            lo[2 * cp + 1][kp] = pk_bf16(r0[1], r1[1]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) hi[c][kp] = h[c];
    }
}

// pieces H..HN-1 of one operand: LDS offsets BASE + H*STEP (+ the loader's own 1-KiB slot, already in st)
template <int H, int HN, int STEP, int BASE>
__device__ __forceinline__ void issue_run(unsigned st, const void* sb, const unsigned (&vo)[HN]) {
    if constexpr (H < HN) {
        dma_piece<BASE + H * STEP>(st, sb, vo[H]);
        issue_run<H + 1, HN, STEP, BASE>(st, sb, vo);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// PRODUCER / CONSUMER form for weight projections (pre-split A), 128 x 256 tiles, one workgroup of eight wavefronts per CU:
//
//   producers (4): keep three rings in LDS full, D slabs ahead and across tile boundaries — pre-split A fragments (DMA),
//                  raw fp32 B rows (DMA) — and SPLIT the raw B slab of the next step into the bf16 hi/lo fragment image the
//                  consumers read (each producer owns 64 columns: 4 ds_read_b128, ~45 VALU, 8 ds_write_b64);
//   consumers (4): a 64 x 128 sub-tile each: 12 ds_read_b128 and 24 MFMAs per slab, nothing else in the loop; epilogue.
//
// The B operand is therefore split ONCE per workgroup tile (not once per wavefront row), the consumers' slab time is the
// MFMA time (24 x 32 cycles), and the vm counters never mix loads and stores.  One s_barrier per slab:
// at barrier g the producers guarantee "A slab g landed, B slab g split" and learn "slab g-1 was read".
//   LDS: A ring (D+1) x 8 KiB | raw B ring D x 16 KiB | split B 2 x 16 KiB  = 136 KiB at D = 4.
// WN = 2: 128 x 256 tiles, 4 + 4 wavefronts, one workgroup per CU (D = 4: 136 KiB of LDS).
// WN = 1: 128 x 128 tiles, 2 + 2 wavefronts, TWO workgroups per CU (D = 3: 72 KiB each): one workgroup's stores and epilogue
//         meet the other's slab loop in the CU's (in-order) memory pipeline.
// STAT (with LNP, unsplit reductions, K % 16 == 0): the per-pixel LayerNorm statistics are not read but MADE here — the producers
// see every fp32 row of the B tile when they split it, accumulate shifted sums per column (shift = row 0, as rcot_ln_stats) and
// leave (mu, rstd) of the tile's columns in LDS before the barrier of its last slab; the consumers pick them up for the fold
// epilogue and row tile 0 writes them to HBM for the backward pass.  No separate pass over x, no extra launch.
// x3p_body: the workgroup (bidx of G workgroups of THIS product; x3p_kernel passes blockIdx.x / gridDim.x, the paired launch at the
// end of this file its own numbering).
// NT = 2: bf16x3 (x = hi + lo, products hi*hi + hi*lo + lo*hi).  NT = 3: bf16x6, the fp32-CLASS arithmetic (RCOT_PREC_BF16X6):
// x = t0 + t1 + t2 with three bfloat16 terms (8 + 8 + 8 significand bits: fp32's 24) and the SIX products of order <= 2
// (t0 t0', t0 t1', t1 t0', t0 t2', t2 t0', t1 t1'; the dropped ones are 2^-24 relative and below): the term-dropping error is
// 6e-9 of max|C|, under the rounding of the fp32 accumulation every fp32 GEMM has (scripts/micro/bf16x6_error.py) — fp32
// results from the bf16 MFMA pipe at 6 x 32 cycles per 32x32x16 block where v_mfma_f32_32x32x2_f32 needs 8 x 64.
template <bool LNP, bool ADD, int WN, int D, bool CONV = false, bool STAT = false, int NT = 2>
__device__ __forceinline__ void x3p_body(const P& p, const int bidx, const int G) {
    // NT = 1: EXACT fp32 (RCOT_PREC_FP32) on the same data path — the producers only move data (the fp32 K-major operand p.At, batch-
    // dependent or not, and the raw B rows), the consumers multiply straight from the raw rings with v_mfma_f32_32x32x2_f32: a lane's
    // B operands of one k step for its four interleaved column tiles are ONE ds_read_b128 (row 2s + kg, columns 4 lm .. + 3), its A
    // operand one ds_read_b32 (row 2s + kg of the [16][128] K-major A stage).  64 MFMAs of 64 cycles per slab and consumer: the four
    // SIMDs of a CU are busy with nothing but MFMA while the producers keep D slabs in flight — what gemm_xx_kernel, whose
    // wavefronts load, wait, multiply and store in turn, reaches half of (510 <- 96 + LN at 8 x 128x128: 164 us, 78 TF/s).
    constexpr bool F32 = NT == 1;
    constexpr int TM = 2, BM = 128, BN = 128 * WN, RA = D + 1, RB = F32 ? D + 1 : D;     // (F32: the consumers read the raw B ring itself)
    constexpr int NC = 2 * WN, NP = 2 * WN;                             // consumer / producer wavefronts
    constexpr unsigned A_ST = F32 ? 8192 : 4096 * NT, B_ST = 8192 * WN; // A stage: 4 row tiles x NT terms x 1 KiB (F32: [16][128] floats); raw B stage
    constexpr unsigned S_ST = F32 ? 0 : 4096 * NT * WN;                 // split-B stage: per 128 columns 4 column tiles x NT terms x 1 KiB
    constexpr unsigned AREC = 1024 * NT;                                // bytes of one (slab, row tile) record of the pre-split pack
    constexpr unsigned RAW0 = RA * A_ST, SPL0 = RAW0 + RB * B_ST;
    constexpr unsigned STAT0 = SPL0 + 2 * S_ST, STAT_ST = 2 * BN * 4;   // STAT: [tile parity][mu | rstd][BN] floats
    static_assert(!STAT || LNP, "statistics are made for the LayerNorm fold only");
    static_assert(NT >= 1 && NT <= 3, "exact fp32, or two / three bf16 terms");
    static_assert(!(F32 && CONV), "the padded-plane convolutions have no exact-fp32 form here");
    constexpr int PLA = (F32 ? 8 : 4 * NT) / NP, PLW = PLA + 4;         // DMA ops per producer per slab (A pieces + 4 B)
    static_assert((D - 1) * PLW <= 63, "vmcnt is a 6-bit field");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 31, kg = lane >> 5;
    const int vb = xcd_remap(bidx, G);
    const int ntiles = p.ntiles;
    const int nk_all = (p.K + BK - 1) / BK;
    const char* ldsc = (const char*)lds;

    if (wave >= NC) {
        // ================================ producer j: columns [64 j, 64 j + 64) ================================
        // lane (lq = lane & 15, kq = lane >> 4): columns 64 j + 4 lq .. + 3, slab rows 4 kq .. 4 kq + 3
        const int j = wave - NC;
        const int lq = lane & 15, kq = lane >> 4;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
        const unsigned ldb4 = (unsigned)p.ldb * 4u;
        unsigned voffB[4], voffA[PLA];
#pragma unroll
        for (int q = 0; q < 4; ++q) voffB[q] = (unsigned)(4 * q + kq) * ldb4 + (unsigned)(j * 64 + 4 * lq) * 4u;   // piece q = rows 4q..4q+3
        int it = vb, ikt = 0, ik0 = 0, ink = 0, gi = 0;
        const char* iA = nullptr;
        const char* iB = nullptr;
        const char* iB0 = nullptr;                                      // CONV: tile base; (ctap, cgrp) = the slab iB points at
        int ctap = 0, cgrp = 0;
        const long strideA = F32 ? (long)BK * p.lda * 4 : (long)p.MT * AREC, strideB = (long)BK * p.ldb * 4;
        auto icursor = [&]() {
            const int tm = it % p.tilesM, r0 = it / p.tilesM;
            const int ks = r0 % p.S, r = r0 / p.S;
            ik0 = ks * p.kchunk;
            ink = min(p.kchunk, nk_all - ik0);
            const int tn = r % p.tilesN, z = r / p.tilesN;
            const int zo = z / p.Zi, zi = z - zo * p.Zi;
            if (F32) {
                // piece qa = rows 2 qa, 2 qa + 1 of the slab's [16][128] block of the K-major operand; columns past the operand's
                // row length (the last row tile of M = 510: 512 columns exist) re-read column 0 (finite; rows >= M are never stored)
                iA = (const char*)(p.At + zo * p.sAo + zi * p.sAi) + (long)ik0 * strideA;
                const int col = tm * 128 + (lane & 31) * 4;
#pragma unroll
                for (int h = 0; h < PLA; ++h) {
                    const int qa = j + NP * h;
                    voffA[h] = (unsigned)(2 * qa + (lane >> 5)) * (unsigned)p.lda * 4u + (unsigned)(col + 4 <= p.lda ? col : 0) * 4u;
                }
            } else {
                iA = (const char*)p.Apk + (long)ik0 * strideA;
#pragma unroll
                for (int h = 0; h < PLA; ++h) {
                    const int qa = j + NP * h;                            // piece = (row tile qa / NT, term qa % NT)
                    const int mt = min(tm * 4 + qa / NT, p.MT - 1);       // row tiles beyond the pack repeat its last one (never stored)
                    voffA[h] = (unsigned)(mt * AREC + (qa % NT) * 1024 + lane * 16);
                }
            }
            if (CONV) {
                iB0 = (const char*)(p.B + zo * p.sBo + zi * p.sBi + tn * BN);
                ctap = ik0 / p.conv_cslabs;
                cgrp = ik0 - ctap * p.conv_cslabs;
                iB = iB0 + (long)cgrp * strideB + (long)p.tapoff[min(ctap, 15)] * 4;
            } else {
                iB = (const char*)(p.B + zo * p.sBo + zi * p.sBi + tn * BN) + (long)ik0 * strideB;
            }
        };
        auto issue_next = [&]() {
            const unsigned sa = lds0 + (unsigned)(gi % RA) * A_ST + (unsigned)j * 1024u;
            const unsigned sb = lds0 + RAW0 + (unsigned)(gi % RB) * B_ST + (unsigned)j * 4096u;
            issue_run<0, PLA, NP * 1024, 0>(sa, iA, voffA);
            if ((ik0 + ikt + 1) * BK > p.K) {
                // last slab of a reduction that is not a multiple of 16: rows >= K repeat row K-1 (finite; the matching A rows are 0)
                const int kbase = (ik0 + ikt) * BK + kq - (p.K - 1);
                unsigned vo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) vo[q] = voffB[q] - (unsigned)max(kbase + 4 * q, 0) * ldb4;
                issue_run<0, 4, 1024, 0>(sb, iB, vo);
            } else {
                issue_run<0, 4, 1024, 0>(sb, iB, voffB);
            }
            iA += strideA;
            if (CONV) {
                if (++cgrp == p.conv_cslabs) { cgrp = 0; ++ctap; }
                iB = iB0 + (long)cgrp * strideB + (long)p.tapoff[min(ctap, 15)] * 4;
            } else {
                iB += strideB;
            }
            ++gi;
            if (++ikt == ink) {
                ikt = 0;
                it += G;
                if (it < ntiles) icursor();
            }
        };
        // split this producer's 16 x 64 block of raw slab g into the consumers' fragment image (split stage g & 1):
        // consumer lane (kg = kq >> 1, lm = 16 (j & 1) + lq) of half j >> 1 holds k = 8 kg .. + 7; this lane supplies dwords
        // 2 (kq & 1), 2 (kq & 1) + 1 (two k pairs) of its hi and lo vectors for the four column tiles
        f32x4 x[4];                                                      // raw rows of the slab being split
        auto split_read = [&](int g) {
            const float* raw = (const float*)(ldsc + RAW0 + (unsigned)(g % RB) * B_ST + (unsigned)j * 4096u);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x[kk] = *reinterpret_cast<const f32x4*>(raw + (4 * kq + kk) * 64 + 4 * lq);
        };
        auto split_write = [&](int g) {
            char* dst = (char*)lds + SPL0 + (unsigned)(g & 1) * S_ST + (unsigned)(j >> 1) * (4096u * NT) +
                        (unsigned)(((kq >> 1) * 32 + (j & 1) * 16 + lq) * 16 + (kq & 1) * 8);
            u32x2 hi[4], lo[4], l2[4];
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const f32x4 x0 = x[2 * kp], x1 = x[2 * kp + 1];
                unsigned h[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = pk_bf16(x0[c], x1[c]);
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const f32x2 a0 = {x0[2 * cp], x0[2 * cp + 1]}, a1 = {x1[2 * cp], x1[2 * cp + 1]};
                    const u32x2 t0 = {h[2 * cp] << 16, h[2 * cp + 1] << 16};
                    const u32x2 t1 = {h[2 * cp] & 0xffff0000u, h[2 * cp + 1] & 0xffff0000u};
                    const f32x2 r0 = a0 - __builtin_bit_cast(f32x2, t0);                  // exact in fp32
                    const f32x2 r1 = a1 - __builtin_bit_cast(f32x2, t1);
                    const unsigned m0 = pk_bf16(r0[0], r1[0]), m1 = pk_bf16(r0[1], r1[1]);
                    lo[2 * cp][kp] = m0;
                    lo[2 * cp + 1][kp] = m1;
                    if (NT == 3) {                                                        // third term: what the second left over
                        const u32x2 u0 = {m0 << 16, m1 << 16};
                        const u32x2 u1 = {m0 & 0xffff0000u, m1 & 0xffff0000u};
                        const f32x2 q0 = r0 - __builtin_bit_cast(f32x2, u0);
                        const f32x2 q1 = r1 - __builtin_bit_cast(f32x2, u1);
                        l2[2 * cp][kp] = pk_bf16(q0[0], q1[0]);
                        l2[2 * cp + 1][kp] = pk_bf16(q0[1], q1[1]);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) hi[c][kp] = h[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                *reinterpret_cast<u32x2*>(dst + (NT * c) * 1024) = hi[c];
                *reinterpret_cast<u32x2*>(dst + (NT * c + 1) * 1024) = lo[c];
                if (NT == 3) *reinterpret_cast<u32x2*>(dst + (NT * c + 2) * 1024) = l2[c];
            }
        };
        // STAT: shifted column sums over the rows of the slab in x[] (this lane: 4 columns x 4 rows), closed at the tile's last slab
        // (packed fp32 arithmetic on column pairs: 3 instructions per 2 elements — the producers are the slower role)
        f32x2 st1[2] = {{0.f, 0.f}, {0.f, 0.f}}, st2[2] = {{0.f, 0.f}, {0.f, 0.f}}, sh0[2] = {{0.f, 0.f}, {0.f, 0.f}};
        int tslab = 0;
        unsigned tpar = 0;
        auto stat_acc = [&]() {
            if (tslab == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    sh0[h] = f32x2{__shfl(x[0][2 * h], lq), __shfl(x[0][2 * h + 1], lq)};   // row 0 of the reduction: the kq == 0 lanes hold it
                    st1[h] = f32x2{0.f, 0.f};
                    st2[h] = f32x2{0.f, 0.f};
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 d = f32x2{x[kk][2 * h], x[kk][2 * h + 1]} - sh0[h];
                    st1[h] += d;
                    st2[h] = __builtin_elementwise_fma(d, d, st2[h]);
                }
            if (++tslab == nk_all) {
                tslab = 0;
                f32x4 m4, r4;
                const float inv = 1.0f / (float)p.K;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = st1[c >> 1][c & 1], b = st2[c >> 1][c & 1];
                    a += __shfl_xor(a, 16);
                    b += __shfl_xor(b, 16);
                    a += __shfl_xor(a, 32);
                    b += __shfl_xor(b, 32);
                    const float e = a * inv;
                    const float var = fmaxf(b * inv - e * e, 0.f);
                    m4[c] = sh0[c >> 1][c & 1] + e;
                    r4[c] = 1.0f / sqrtf(var + 1e-5f);
                }
                if (kq == 0) {
                    float* sb = (float*)((char*)lds + STAT0 + tpar * STAT_ST) + j * 64 + 4 * lq;
                    *reinterpret_cast<f32x4*>(sb) = m4;
                    *reinterpret_cast<f32x4*>(sb + BN) = r4;
                }
                tpar ^= 1u;
            }
        };
        int total = 0;                                                   // slabs this workgroup walks
        for (int t = vb; t < ntiles; t += G) total += min(p.kchunk, nk_all - ((t / p.tilesM) % p.S) * p.kchunk);
        icursor();
#pragma unroll
        for (int i = 0; i < D; ++i)
            if (it < ntiles) issue_next();
#if defined(X3_TRACE) && X3_TRACE >= 3     // producer phases of workgroup 0 (perturbs the cadence it measures)
#define PSTAMP(i) do { if (p.trace && j == 0 && lane == 0 && (i) < 56) p.trace[(long)blockIdx.x * 64 + 8 + (i)] = wall_clock64(); } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif
        PSTAMP(0);
        wait_slabs<PLW, D + 1>(gi - 1);                                 // slab 0 landed (gi - 1 younger groups)
        PSTAMP(1);
        if (!F32 || STAT) split_read(0);
        if (!F32) split_write(0);
        if (STAT) stat_acc();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PSTAMP(2);
        __builtin_amdgcn_s_barrier();                                    // barrier 0
        PSTAMP(3);
        for (int g = 0; g + 1 < total; ++g) {
            wait_slabs<PLW, D + 1>(gi - (g + 1) - 1);                    // slab g + 1 landed
            PSTAMP(4 + 4 * g);
            if (!F32 || STAT) split_read(g + 1);                         // its LDS reads fly under the DMA issue below
            if (it < ntiles) issue_next();                               // slab g + D (the stage slab g - 1 .. occupied)
            PSTAMP(5 + 4 * g);
            if (!F32) split_write(g + 1);
            if (STAT) stat_acc();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PSTAMP(6 + 4 * g);
            __builtin_amdgcn_s_barrier();                                // barrier g + 1
            PSTAMP(7 + 4 * g);
        }
        return;
    }

    // ================================ consumer (wm, wn): rows 64 wm.., columns 128 wn.. ================================
    const int wm = wave / WN, wn = wave % WN;
    const EpiP& ep = p.ep;
    int gc = 0;
    unsigned tpar = 0;                                                   // STAT: parity of this workgroup's tile count
#ifdef X3_TRACE
    if (p.trace && threadIdx.x == 0) p.trace[(long)blockIdx.x * 64 + 0] = wall_clock64();
#endif
    for (int t = vb; t < ntiles; t += G) {
        const int tm = t % p.tilesM, r0_ = t / p.tilesM;
        const int ks = r0_ % p.S, r_ = r0_ / p.S;
        const int nk = min(p.kchunk, nk_all - ks * p.kchunk);
        const int tn = r_ % p.tilesN, z = r_ / p.tilesN;
        const int zo = z / p.Zi, zi = z - zo * p.Zi;
        const int mb0 = tm * BM + wm * 64 + 4 * kg, ncol = tn * BN + wn * 128 + 4 * lm;
        // ---- everything the epilogue reads is requested NOW: the loads fly under the slab loop (requested one group at a time
        // in the epilogue, they cost one L2 round trip per row group: 6-7 us per tile against 3.6 us of slab loop at K = 96)
        float* Cb = ep.C + zo * ep.sCo + zi * ep.sCi;
        const float* Rb = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
        const float* Ad = Rb ? Rb : (ep.beta != 0.f ? Cb : nullptr);
        const long ldad = Rb ? ep.ldr : ep.ldc;
        const bool both = Rb && ep.beta != 0.f;
        float* Wb = p.S > 1 ? p.ws + ((long)z * p.S + ks) * p.M * p.N : nullptr;   // split-K: raw partial sums to the workspace
        const bool addend = ADD && Ad && !Wb;     // ADD = false: the caller guarantees there is no residual and beta == 0
        f32x4 rs4 = {1.f, 1.f, 1.f, 1.f}, murs4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 c1v[4], c2v[4];                                           // LN constants of one 32-row tile (four groups of four rows)
        const bool lnc = p.S == 1 || ks == 0;                            // the LN constants ride on piece 0 of a split reduction
        auto load_ln = [&](int i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m4 = mb0 + 32 * i + 8 * g;                     // four consecutive rows; c1/c2 are padded to ceil4(M)
                const bool ok = lnc && m4 < p.M;
                c1v[g] = ok ? *reinterpret_cast<const f32x4*>(p.c1 + m4) : f32x4{0.f, 0.f, 0.f, 0.f};
                c2v[g] = ok ? *reinterpret_cast<const f32x4*>(p.c2 + m4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        if (LNP) {
            if (!STAT) {
                const long n = (long)zo * p.sLN + ncol;
                rs4 = *reinterpret_cast<const f32x4*>(p.rs + n);
                murs4 = *reinterpret_cast<const f32x4*>(p.mu + n);
            }
            load_ln(0);
        }
        int coff = 0;                                                    // CONV: dense offset of this lane's 4-column group
        auto load_bias = [&](int i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m4 = mb0 + 32 * i + 8 * g;                     // cbias is padded to a multiple of 4 rows
                c2v[g] = (p.cbias && m4 < p.M) ? *reinterpret_cast<const f32x4*>(p.cbias + m4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        if (CONV) {
            coff = p.colmap[ncol >> 2];
            load_bias(0);
        }
        // addend rows in batches of eight (batch b = rows 8 (b & 1) .. + 7 of 32-row tile b >> 1), double-buffered
        f32x4 rqA[8], rqB[8];
        const float rsc = Rb ? 1.f : ep.beta;                            // (no per-row scale on this path: host check)
        auto load_addend = [&](int bt, f32x4 (&rq)[8]) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = 8 * (bt & 1) + r8;
                const int m = mb0 + 32 * (bt >> 1) + 8 * (r >> 2) + (r & 3);
                rq[r8] = m < p.M ? *reinterpret_cast<const f32x4*>(Ad + ((unsigned)m * (unsigned)ldad + (unsigned)ncol)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        f32x16 acc[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();              // barrier gc: A slab gc landed, B slab gc split
#if defined(X3_TRACE) && X3_TRACE >= 2
            if (kt < 48) X3_STAMP(8 + kt);
#endif
            if (F32) {
                // exact fp32: k step s multiplies rows 2 s + kg of the raw A / B stages (v_mfma_f32_32x32x2_f32: lane (lm, kg) supplies
                // A[row lm][k = kg] and B[k = kg][column lm]); raw B stage = per producer j a [16][64] block (the DMA image)
                const float* Af = (const float*)(ldsc + (unsigned)(gc % RA) * A_ST) + kg * 128 + 32 * (wm * TM) + lm;
                const float* Bf = (const float*)(ldsc + RAW0 + (unsigned)(gc % RB) * B_ST + (unsigned)(wn * 2 + (lm >> 4)) * 4096u) + kg * 64 + 4 * (lm & 15);
                ++gc;
#ifndef X3W_NO_COMPUTE
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(Bf + s2 * 128);
                    float av[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[i] = Af[s2 * 256 + 32 * i];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[c], acc[i][c], 0, 0, 0);
                }
#endif
                continue;
            }
            const char* As = ldsc + (unsigned)(gc % RA) * A_ST + lane * 16;
            const char* Bs = ldsc + SPL0 + (unsigned)(gc & 1) * S_ST + (unsigned)wn * (4096u * NT) + lane * 16;
            ++gc;
#ifndef X3W_NO_COMPUTE
            bf16x8 ah[TM], al[TM], a2[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(As + (NT * (wm * TM + i)) * 1024);
                al[i] = *reinterpret_cast<const bf16x8*>(As + (NT * (wm * TM + i) + 1) * 1024);
                if (NT == 3) a2[i] = *reinterpret_cast<const bf16x8*>(As + (NT * (wm * TM + i) + 2) * 1024);
            }
            bf16x8 bhv[4], blv[4], b2v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bhv[c] = *reinterpret_cast<const bf16x8*>(Bs + (NT * c) * 1024);
                blv[c] = *reinterpret_cast<const bf16x8*>(Bs + (NT * c + 1) * 1024);
                if (NT == 3) b2v[c] = *reinterpret_cast<const bf16x8*>(Bs + (NT * c + 2) * 1024);
            }
            if (NT == 3) {                              // the three second-order products first (smallest magnitudes)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], blv[c], acc[i][c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], bhv[c], acc[i][c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b2v[c], acc[i][c], 0, 0, 0);
                }
            }
            if (NT != 2 || !ep.one) {                   // RCOT_PREC_BF16X1 (NT = 2, ep.one): the single product hi * hi only
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bhv[c], acc[i][c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], blv[c], acc[i][c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bhv[c], acc[i][c], 0, 0, 0);
            }
#else
            (void)As; (void)Bs;
#endif
        }
#ifdef X3_TRACE
        if (p.trace && threadIdx.x == 0 && (t - vb) / G < 20) p.trace[(long)blockIdx.x * 64 + 23 + 2 * ((t - vb) / G)] = wall_clock64();
#endif
        X3_STAMP(2);
        // ---- epilogue: lane holds columns ncol..ncol+3 of rows mb0 + 32 i + 8 hf + (0..3) in acc[i][0..3][4 hf + (0..3)]
        if (STAT) {
            // left by the producers before the barrier of the last slab
            const float* sb = (const float*)(ldsc + STAT0 + tpar * STAT_ST) + wn * 128 + 4 * lm;
            murs4 = *reinterpret_cast<const f32x4*>(sb);
            rs4 = *reinterpret_cast<const f32x4*>(sb + BN);
            tpar ^= 1u;
            if (tm == 0 && wm == 0 && kg == 0) {
                const long n = (long)zo * p.sLN + ncol;
                *reinterpret_cast<f32x4*>(p.mu_out + n) = murs4;
                *reinterpret_cast<f32x4*>(p.rs_out + n) = rs4;
            }
        }
        if (LNP) murs4 = murs4 * rs4;
        // one destination, 32-bit element offsets (host checks the range), no per-row guards unless this is the last row tile
        float* dstb = Wb ? Wb : Cb;
        const unsigned ldd = Wb ? (unsigned)p.N : (unsigned)ep.ldc;
        const bool inner = tm * BM + wm * 64 + 64 <= p.M;                // wave-uniform: all 64 rows of this wavefront exist
        auto store_batch = [&](int bt, const f32x4 (&rq)[8], auto guard) {
            constexpr bool GUARD = decltype(guard)::value;
            const int i = bt >> 1;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = 8 * (bt & 1) + r8, hf = r >> 2, r4 = r & 3;
                const int m = mb0 + 32 * i + 8 * hf + r4;
                if (GUARD && m >= p.M) continue;
                // acc index must be static: i is a compile-time constant at every call site (bt literal)
                f32x4 v = i == 0 ? f32x4{acc[0][0][r], acc[0][1][r], acc[0][2][r], acc[0][3][r]}
                                 : f32x4{acc[TM - 1][0][r], acc[TM - 1][1][r], acc[TM - 1][2][r], acc[TM - 1][3][r]};
                if (LNP) v = v * rs4 + (c2v[hf][r4] - murs4 * c1v[hf][r4]);   // LN fold (gemm_x3.hip header)
                if (ADD && addend) v += rq[r8] * rsc;
                if (CONV && !Wb) {
                    if (coff < 0) continue;                                   // padding positions of the padded plane
                    v += c2v[hf][r4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.clrelu;
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dstb + ((unsigned)m * ldd + (unsigned)coff)));
                    continue;
                }
                float* dst = dstb + ((unsigned)m * ldd + (unsigned)ncol);
                if (ADD && both) v += *reinterpret_cast<const f32x4*>(dst) * ep.beta;
#ifndef X3W_NO_STORE
                // streaming store: the tile is written once and read by a later launch; without the hint the write-allocated lines
                // push the B rows the neighbouring row tiles are about to re-read out of L2 (510 <- 96: 114 -> 99 us)
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#else
                if (v[0] == 1.2345f) *reinterpret_cast<f32x4*>(dst) = v;
#endif
            }
        };
        auto epilogue = [&](auto guard) {
            if (ADD && addend) {
                load_addend(0, rqA);
                load_addend(1, rqB);
            }
            store_batch(0, rqA, guard);
            if (ADD && addend) load_addend(2, rqA);
            store_batch(1, rqB, guard);
            if (LNP) load_ln(1);
            if (CONV) load_bias(1);
            if (ADD && addend) load_addend(3, rqB);
            store_batch(2, rqA, guard);
            store_batch(3, rqB, guard);
        };
        if (inner) epilogue(std::false_type{});
        else epilogue(std::true_type{});
        X3_STAMP(4);
#ifdef X3_TRACE
        if (p.trace && threadIdx.x == 0 && (t - vb) / G < 20) {
            p.trace[(long)blockIdx.x * 64 + 24 + 2 * ((t - vb) / G)] = wall_clock64();
        }
#endif
    }
}

template <bool LNP, bool ADD, int WN, int D, bool CONV = false, bool STAT = false, int NT = 2>
__global__ __launch_bounds__(256 * WN, 2) void x3p_kernel(P p) {
    x3p_body<LNP, ADD, WN, D, CONV, STAT, NT>(p, blockIdx.x, gridDim.x);
}

// C[z] = alpha * sum_ks slab[z][ks] + rowscale * R + beta * C   (fixed summation order; 16 bytes per thread)
__global__ __launch_bounds__(256) void x3w_reduce_kernel(const float* __restrict__ ws, int S, int M, int N4, int Zi, EpiP ep) {
    const long per = (long)M * N4;
    const int z = blockIdx.y, zo = z / Zi, zi = z - zo * Zi;
    const float* w = ws + (long)z * S * per * 4;
    float* Cz = ep.C + zo * ep.sCo + zi * ep.sCi;
    const float* Rz = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
    const float* Sz = ep.rowscale ? ep.rowscale + zo * ep.sSo + zi * ep.sSi : nullptr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N4), n4 = (int)(i - (long)m * N4);
        // every slab of a group of eight (and the addend) is requested before the first is used: one memory round trip for the
        // S <= 8 of the network's shapes (a serial tail loop cost one per slab); fixed pairwise summation order
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        float* dst = Cz + (long)m * ep.ldc + 4 * n4;
        f32x4 rv = z4, cv = z4;
        if (Rz) rv = *reinterpret_cast<const f32x4*>(Rz + (long)m * ep.ldr + 4 * n4);
        if (ep.beta != 0.f) cv = *reinterpret_cast<const f32x4*>(dst);
        f32x4 a = z4;
        for (int s0 = 0; s0 < S; s0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s0 + u < S ? *reinterpret_cast<const f32x4*>(w + ((long)(s0 + u) * per + i) * 4) : z4;
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        a *= ep.alpha;
        if (Rz) a += rv * (Sz ? Sz[m] : 1.f);
        if (ep.beta != 0.f) a += cv * ep.beta;
        *reinterpret_cast<f32x4*>(dst) = a;
    }
}

// tiles, split-K pieces, grid and LDS bytes of one launch of x3p_kernel<.., WN, D, ..>
// (slots_override: the paired launch gives each of its two products half the chip)
template <int WN, int D, int NT = 2>
int configure_p(P& p, bool ln, int Z, size_t ws_bytes, bool stat, int* grid_out, size_t* smem_out, int slots_override = 0) {
    const int slots = slots_override ? slots_override : (WN == 1 ? 512 : 256);   // resident workgroups on the chip
    p.tilesM = cdiv(p.M, 128);
    p.tilesN = p.N / (128 * WN);
    const int nk = cdiv(p.K, BK);
    const int base = p.tilesM * p.tilesN * Z;
    int S = 1;
    static const int nk_min = getenv("RCOT_X3P_SPLIT_NK") ? atoi(getenv("RCOT_X3P_SPLIT_NK")) : 16;   // tuning: split only reductions of >= nk_min slabs
    static const int s_max = getenv("RCOT_X3P_SMAX") ? atoi(getenv("RCOT_X3P_SMAX")) : 1 << 20;
    // split when the tiles fill less than a third of the chip: at half (128 of 256 slots: the 64x64 level with one row tile) two
    // K pieces + the reduce launch measured 25.4 / 26.9 / 30.9 us against 20.0 / 25.6 / 30.3 us unsplit (K = 288 / 255 / 510)
    if (p.ws && base * 3 <= slots && nk >= nk_min) {
        S = slots / base;
        if (S > s_max) S = s_max;
        if (S > nk / 8) S = nk / 8;
        while (S > 1 && (size_t)S * Z * p.M * p.N * sizeof(float) > ws_bytes) --S;
        if (S < 1) S = 1;
    }
    p.kchunk = cdiv(nk, S);
    p.S = cdiv(nk, p.kchunk);
    if (p.S > 1 && nk - (p.S - 1) * p.kchunk < 2) {
        p.kchunk = cdiv(nk, p.S - 1);
        p.S = cdiv(nk, p.kchunk);
    }
    if (stat && (p.S > 1 || !ln || (p.K % BK))) return RCOT_EUNSUPPORTED;   // a K piece does not see the whole column
    p.ntiles = base * p.S;
    const int rounds = cdiv(p.ntiles, slots);
    *grid_out = cdiv(p.ntiles, rounds);
    *smem_out = (NT == 1 ? (size_t)(D + 1) * 8192 + (size_t)(D + 1) * 8192 * WN
                         : (size_t)(D + 1) * 4096 * NT + (size_t)D * 8192 * WN + (size_t)2 * 4096 * NT * WN) +
                (stat ? 2 * 2 * 128 * WN * sizeof(float) : 0);
    return RCOT_OK;
}

inline void launch_p_reduce(const P& p, int Z, hipStream_t st) {
    const long per = (long)p.M * (p.N / 4);
    long nb = (per + 255) / 256;
    if (nb > 2048) nb = 2048;
    RCOT_LAUNCH(x3w_reduce_kernel, dim3((int)nb, Z), dim3(256), 0, st, p.ws, p.S, p.M, p.N / 4, p.Zi, p.ep);
}

template <int WN, int D, int NT = 2>
int launch_p(P p, bool ln, int Z, hipStream_t st, size_t ws_bytes, bool stat = false) {
    int grid = 0;
    size_t smem = 0;
    const int rcc = configure_p<WN, D, NT>(p, ln, Z, ws_bytes, stat, &grid, &smem);
    if (rcc != RCOT_OK) return rcc;
    const bool add = p.S == 1 && (p.ep.R != nullptr || p.ep.beta != 0.f);
#define X3P_LAUNCH(L, A, T)                                                                                                       \
    do {                                                                                                                           \
        note_kernel("x3p_kernel<%s, %s, %d, %d, false, %s, %d>", tf(L), tf(A), WN, D, tf(T), NT);                                  \
        static bool once = (hipFuncSetAttribute((const void*)x3p_kernel<L, A, WN, D, false, T, NT>,                               \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);            \
        (void)once;                                                                                                                \
        RCOT_LAUNCH((x3p_kernel<L, A, WN, D, false, T, NT>), dim3(grid), dim3(256 * WN), smem, st, p);                      \
    } while (0)
    if (stat && add) X3P_LAUNCH(true, true, true);
    else if (stat) X3P_LAUNCH(true, false, true);
    else if (ln && add) X3P_LAUNCH(true, true, false);
    else if (ln) X3P_LAUNCH(true, false, false);
    else if (add) X3P_LAUNCH(false, true, false);
    else X3P_LAUNCH(false, false, false);
#undef X3P_LAUNCH
    RCOT_LAUNCH_CHECK();
    if (p.S > 1) {
        launch_p_reduce(p, Z, st);
        RCOT_LAUNCH_CHECK();
    }
    return RCOT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// PAIRED LAUNCH: the two products of ONE incoming gradient dY of a 1x1 projection,
//     dX = W^T dY (x3p_body, pre-split pack)   and   dW slabs = sum_pixels dY LN?(X)^T (rcot_nt::nt_body, split-K slabs),
// as workgroups of one grid.  They are independent, read the same dY, and each alone is a launch of one or two tiles per CU whose
// time is memory round trips, not work (profiles/r04_block_trace_*.txt: 14-33 us and 16-28 us on the 16x16 / 32x32 levels): one
// after the other they cost the sum, on two streams the cross-stream hand-off costs what the overlap saves (7-12 us per
// switch), hipExtAnyOrderLaunch is not honoured on gfx9 (scripts/micro/anyorder.hip).  In one grid the dispatcher places
// workgroups of both kinds side by side (two per CU, 72 KiB + 51 KiB of LDS) and the launch lasts as long as the longer of the
// two; on the 128x128 level, where both are bandwidth-bound, the second reader of dY finds it in L2 / MALL.
// Workgroups alternate between the products in groups of eight consecutive block ids (one per XCD), so that a body's own
// numbering idx keeps idx % 8 == blockIdx.x % 8: the XCD locality xcd_remap() arranges inside each product is preserved.
template <int TM, int TN, int WM, int WNN, bool LNP>
__global__ __launch_bounds__(256, 2) void x3p_nt_pair_kernel(P p, rcot_nt::NTP q, int nA, int nB) {
    const int b = blockIdx.x;
    const int pairs = min(nA, nB) >> 3;                  // complete alternating pairs of 8-workgroup groups
    const int g = b >> 3;
    int kind, idx;
    if (g < 2 * pairs) {
        kind = g & 1;
        idx = ((g >> 1) << 3) + (b & 7);
    } else {
        const int done = pairs << 3, r = b - 2 * done;   // what is left of each product follows, x3p first
        if (r < nA - done) { kind = 0; idx = done + r; }
        else { kind = 1; idx = done + r - (nA - done); }
    }
    if (kind == 0) x3p_body<false, false, 1, 3, false, false>(p, idx, nA);
    else rcot_nt::nt_body<TM, TN, WM, WNN, LNP, true>(q, idx, 0);
}

template <int TM, int TN, int WM, int WNN>
int launch_pair(const P& p, const rcot_nt::NTP& q, int nA, size_t smemA, hipStream_t st) {
    const int nB = q.tilesM * q.tilesN * q.S;
    const size_t smemB = sizeof(float) * (size_t)rcot_nt::NST * rcot_nt::STAGE;
    const size_t smem = smemA > smemB ? smemA : smemB;
#define PAIR_LAUNCH(L)                                                                                                             \
    do {                                                                                                                           \
        note_kernel("x3p_nt_pair_kernel<%d, %d, %d, %d, %s>", TM, TN, WM, WNN, tf(L));                                             \
        static bool once = (hipFuncSetAttribute((const void*)x3p_nt_pair_kernel<TM, TN, WM, WNN, L>,                               \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);            \
        (void)once;                                                                                                                \
        RCOT_LAUNCH((x3p_nt_pair_kernel<TM, TN, WM, WNN, L>), dim3(nA + nB), dim3(256), smem, st, p, q, nA, nB);            \
    } while (0)
    if (q.mu) PAIR_LAUNCH(true);
    else PAIR_LAUNCH(false);
#undef PAIR_LAUNCH
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

// CONV split-K: out[m][colmap group] = lrelu(sum_ks slab[ks][m][n] + bias[m])   (fixed summation order)
__global__ __launch_bounds__(256) void x3w_conv_reduce_kernel(const float* __restrict__ ws, int S, int M, int N4, const int* __restrict__ colmap,
                                                              const float* __restrict__ bias, float lrelu, float* __restrict__ C, long ldc) {
    const long per = (long)M * N4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N4), n4 = (int)(i - (long)m * N4);
        const int off = colmap[n4];
        if (off < 0) continue;
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + i * 4);
        for (int s = 1; s < S; ++s) a += *reinterpret_cast<const f32x4*>(ws + ((long)s * per + i) * 4);
        if (bias) a += bias[m];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = a[e] > 0.f ? a[e] : a[e] * lrelu;
        *reinterpret_cast<f32x4*>(C + (long)m * ldc + off) = a;
    }
}

template <int WN, int D>
int launch_conv(P p, hipStream_t st, size_t ws_bytes) {
    constexpr int slots = WN == 1 ? 512 : 256;
    p.tilesM = cdiv(p.M, 128);
    p.tilesN = p.N / (128 * WN);
    const int nk = cdiv(p.K, BK);
    const int base = p.tilesM * p.tilesN;
    int S = 1;
    if (p.ws && base * 2 <= slots && nk >= 16) {
        S = slots / base;
        if (S > nk / 8) S = nk / 8;
        while (S > 1 && (size_t)S * p.M * p.N * sizeof(float) > ws_bytes) --S;
        if (S < 1) S = 1;
    }
    p.kchunk = cdiv(nk, S);
    p.S = cdiv(nk, p.kchunk);
    if (p.S > 1 && nk - (p.S - 1) * p.kchunk < 2) {
        p.kchunk = cdiv(nk, p.S - 1);
        p.S = cdiv(nk, p.kchunk);
    }
    p.ntiles = base * p.S;
    const int rounds = cdiv(p.ntiles, slots);
    const int grid = cdiv(p.ntiles, rounds);
    const size_t smem = (size_t)(D + 1) * 8192 + (size_t)(D + 2) * 8192 * WN;
    static bool once = (hipFuncSetAttribute((const void*)x3p_kernel<false, false, WN, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024) == hipSuccess);
    (void)once;
    note_kernel("x3p_kernel<false, false, %d, %d, true, false>", WN, D);
    RCOT_LAUNCH((x3p_kernel<false, false, WN, D, true>), dim3(grid), dim3(256 * WN), smem, st, p);
    RCOT_LAUNCH_CHECK();
    if (p.S > 1) {
        const long per = (long)p.M * (p.N / 4);
        long nb = (per + 255) / 256;
        if (nb > 2048) nb = 2048;
        RCOT_LAUNCH(x3w_conv_reduce_kernel, dim3((int)nb), dim3(256), 0, st, p.ws, p.S, p.M, p.N / 4, p.colmap, p.cbias, p.clrelu,
                           p.ep.C, p.ep.ldc);
        RCOT_LAUNCH_CHECK();
    }
    return RCOT_OK;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace rcot_x3w

namespace rcot {

// Returns RCOT_OK after launching, or -100 when the shape is not eligible (the caller then uses another kernel).
// Apk (optional): the pre-split form of At (rcot_pack_weight), only for batch-invariant A (sAo == sAi == 0).
int try_gemm_kmajor_x3w(const float* At, long lda, long sAo, long sAi, const void* Apk, const float* Bm, long ldb, long sBo,
                        long sBi, const EpiP& ep, const float* ln_mu, const float* ln_rs, long sLN, const float* ln_c1,
                        const float* ln_c2, int Zo, int Zi, int M, int N, int K, float* ws, size_t ws_bytes, hipStream_t st,
                        bool ln_compute, int nterms) {
    using namespace rcot_x3w;
    if ((N % 128) || K < 17) return -100;          // the slab ring needs at least two slabs per tile
    if ((unsigned long)ldb * 4ul * 17ul >= (1ul << 32)) return -100;   // 32-bit per-lane DMA offsets
    const bool ln = ln_mu != nullptr;
    if (ln && ((sLN & 3) || !al16(ln_mu) || !al16(ln_rs) || !ln_c1 || !ln_c2 || !al16(ln_c1) || !al16(ln_c2))) return -100;
    P p{};
    p.M = M; p.N = N; p.K = K; p.Zi = Zi;
    p.At = At; p.lda = lda; p.sAo = sAo; p.sAi = sAi;
    p.Apk = ((sAo == 0 || Zo == 1) && (sAi == 0 || Zi == 1)) ? (const unsigned char*)Apk : nullptr;
    p.MT = cdiv(M, 32);
    p.B = Bm; p.ldb = ldb; p.sBo = sBo; p.sBi = sBi;
    p.mu = ln_mu; p.rs = ln_rs; p.sLN = sLN; p.c1 = ln_c1; p.c2 = ln_c2;
    p.mu_out = const_cast<float*>(ln_mu); p.rs_out = const_cast<float*>(ln_rs);   // ln_compute: outputs (the C entry point takes them non-const)
    p.ep = ep;
    p.ws = ws;
#ifdef X3_TRACE
    p.trace = g_x3w_trace;
#endif
    const int Z = Zo * Zi;
    if (ln_compute && (!ln || Zi != 1)) return -100;
    static const int force = getenv("RCOT_X3P_WN") ? atoi(getenv("RCOT_X3P_WN")) : 0;
    if ((nterms != 1 && !p.Apk) || (N % 128) || M <= 64 || ep.alpha != 1.f || ep.rowscale || (long)M * ep.ldc >= (1l << 31) ||
        (long)M * N >= (1l << 31) || (ep.R && (long)M * ep.ldr >= (1l << 31)))
        return -100;
    if (nterms == 1) {                                                  // exact fp32: the fp32 K-major operand itself (batch-dependent or not)
        if ((unsigned long)lda * 4ul * 17ul >= (1ul << 32) || (lda & 3) || !al16(At) || (sAo & 3) || (sAi & 3)) return -100;
        if (force == 1 || (N % 256)) return launch_p<1, 3, 1>(p, ln, Z, st, ws_bytes, ln_compute);
        return launch_p<2, 5, 1>(p, ln, Z, st, ws_bytes, ln_compute);  // six-stage rings of 8 + 16 KiB: 144 KiB
    }
    // Tile form by occupancy (round 5, profiles/r05_x3p_tile_forms_x6_x3.txt): 128 x 256 tiles leave half the chip idle when a product has
    // 48..200 of them (the small planes: 2042 <- 384 at 8 x 16x16 is 128 tiles) — 128 x 128 tiles double the workgroups: 24.2 -> 17.6 us
    // there, 29.0 -> 21.5 us for the data gradient of 510 <- 96 at 64x64 (bf16x3; bf16x6 34.6 -> 26.4, 45.2 -> 32.8).  Fewer than 48
    // tiles means a split reduction, where the wide tile's longer slabs win; more than 200 fill the chip either way.
    // In situ (scripts/ab_wn.sh, profiles/r05_ab_tile_rule.txt) the rule is worth 0.2 ms over the unit's blocks in bf16x3 (16x16 block forward
    // 124.5 -> 120.0 us) and LOSES in bf16x6 (32x32 block forward 128 -> 140 us: the narrow three-term form has a three-slab ring in 96 KiB,
    // one workgroup per CU either way): bf16x3 only.
    const long tiles256 = (long)cdiv(M, 128) * (N / 256) * Z;
    const bool narrow = force == 1 || (N % 256) || (force == 0 && nterms == 2 && tiles256 >= 48 && tiles256 <= 200);
    if (nterms == 3) {                                                  // bf16x6: Apk is the THREE-term pack (3 KiB records)
        if (narrow) return launch_p<1, 3, 3>(p, ln, Z, st, ws_bytes, ln_compute);
        return launch_p<2, 3, 3>(p, ln, Z, st, ws_bytes, ln_compute);  // ring of three slabs: 144 KiB of LDS with the wider fragment images
    }
    if (narrow) return launch_p<1, 3>(p, ln, Z, st, ws_bytes, ln_compute);
    return launch_p<2, 4>(p, ln, Z, st, ws_bytes, ln_compute);
}

// dX[b] (Ci x N) = W^T dY[b] on the pre-split pack WPs of W, and the split-K slabs of dW = sum_b dY[b] LN?(X[b])^T, in ONE launch
// (x3p_nt_pair_kernel).  -100: one of the two products is not eligible for its kernel (the caller runs them separately).
int pair_dgrad_wgrad_x3(const float* WP, long ldp, const void* WPs, const float* dY, long sdYb, float* dX, long sdXb, const float* X,
                        long sXb, int B, int Ci, int Co, int N, const float* ln_mu, const float* ln_rs, const float* ln_w,
                        const float* ln_b, float* ws, size_t ws_bytes, float* ws_slabs, size_t ws_slabs_bytes, int* S_out,
                        int* ld_out, hipStream_t st) {
    using namespace rcot_x3w;
    const int M = Ci, K = Co;
    // measured (scripts/small_levels.py, one call): 16x16 blocks 303 -> 265 us backward, 32x32 308 -> 297, 64x64 336 -> 317; at 128x128
    // both products are bandwidth-bound and the pair is SLOWER than the two launches (893 -> 925 us): planes above 4096 pixels
    // keep the separate launches (weight gradient on the side stream)
    static const int pair_maxn = getenv("RCOT_PAIR_MAXN") ? atoi(getenv("RCOT_PAIR_MAXN")) : 4096;
    if (N > pair_maxn) return -100;
    if (!WPs || (N % 128) || K < 17 || M <= 64 || (long)M * N >= (1l << 31) || (unsigned long)N * 4ul * 17ul >= (1ul << 32)) return -100;
    P p{};
    p.M = M; p.N = N; p.K = K; p.Zi = 1;
    p.At = WP; p.lda = ldp;
    p.Apk = (const unsigned char*)WPs;
    p.MT = cdiv(M, 32);
    p.B = dY; p.ldb = N; p.sBo = sdYb;
    p.ep.C = dX; p.ep.ldc = N; p.ep.sCo = sdXb; p.ep.alpha = 1.f; p.ep.lrelu = 1.f;
    p.ws = ws;
#ifdef X3_TRACE
    p.trace = nullptr;
#endif
    int nA = 0;
    size_t smemA = 0;
    static const int slotsA = getenv("RCOT_PAIR_SLOTS_A") ? atoi(getenv("RCOT_PAIR_SLOTS_A")) : 256;
    static const int slotsB = getenv("RCOT_PAIR_SLOTS_B") ? atoi(getenv("RCOT_PAIR_SLOTS_B")) : 320;
    if (configure_p<1, 3>(p, false, B, ws_bytes, false, &nA, &smemA, slotsA) != RCOT_OK) return -100;
    rcot_nt::NTP q{};
    int cfg = 0;
    if (nt_configure(Co, Ci, B * N, 1, 1, dY, N, 0, 0, X, N, 0, 0, N, sdYb, sXb, ln_mu, ln_rs, N, ln_w, ln_b, ws_slabs, ws_slabs_bytes, 1, 0,
                     &q, &cfg, slotsB) != RCOT_OK)
        return -100;
    *S_out = q.S;
    *ld_out = q.ldws;
    int rc;
    if (cfg == 1) rc = launch_pair<1, 3, 4, 1>(p, q, nA, smemA, st);
    else if (cfg == 2) rc = launch_pair<3, 1, 1, 4>(p, q, nA, smemA, st);
    else if (cfg == 3) rc = launch_pair<1, 2, 4, 1>(p, q, nA, smemA, st);
    else if (cfg == 4) rc = launch_pair<2, 1, 1, 4>(p, q, nA, smemA, st);
    else if (cfg == 5) rc = launch_pair<1, 1, 2, 2>(p, q, nA, smemA, st);
    else rc = launch_pair<2, 2, 2, 2>(p, q, nA, smemA, st);
    if (rc != RCOT_OK) return rc;
    if (p.S > 1) {
        launch_p_reduce(p, B, st);
        RCOT_LAUNCH_CHECK();
    }
    return RCOT_OK;
}

// Dense convolution as a K-major product over a padded, channel-major copy of the input (csrc/conv_pcm.hip): Y[m][colmap[n/4]] =
// lrelu(sum_{tap, c} A[m][(tap, c)] Xp[c][n + tapoff[tap]] + bias[m]).  Apk: pre-split pack of A (K = ntaps * Ci, Ci % 16 == 0).
int conv_pcm_x3w(const void* Apk, int M, int K, const float* Xp, long ldb, int N, const int* tapoff, int ntaps, const float* bias,
                 float lrelu, const int* colmap, float* Y, long ldy, float* ws, size_t ws_bytes, hipStream_t st) {
    using namespace rcot_x3w;
    if (!Apk || !Xp || !colmap || !Y || ntaps < 1 || ntaps > 16 || (K % (16 * ntaps)) || (N % 128) || M <= 0 || (ldb & 3)) return RCOT_EINVAL;
    if ((unsigned long)ldb * 4ul * 17ul >= (1ul << 32) || (long)M * ldy >= (1l << 31) || (long)M * N >= (1l << 31)) return RCOT_EINVAL;
    P p{};
    p.M = M; p.N = N; p.K = K; p.Zi = 1;
    p.Apk = (const unsigned char*)Apk;
    p.MT = cdiv(M, 32);
    p.B = Xp; p.ldb = ldb;
    p.ep.C = Y; p.ep.ldc = ldy; p.ep.alpha = 1.f;
    p.ws = ws;
    p.conv_cslabs = K / (16 * ntaps);
    for (int t = 0; t < 16; ++t) p.tapoff[t] = t < ntaps ? tapoff[t] : 0;
    p.colmap = colmap; p.cbias = bias; p.clrelu = lrelu;
    if (N % 256) return launch_conv<1, 3>(p, st, ws_bytes);
    return launch_conv<2, 4>(p, st, ws_bytes);
}

}  // namespace rcot
