// STUDY KERNEL (round 5, session 4) — kept for its timing decomposition, NOT part of the library and NOT numerically right as it
// stands: it waits for its ring stages with immediate vmcnt counts across a mix of LDS-DMA loads and buffer stores, and the padding
// stores (no records: dropped by the hardware) retire out of order, so a stage can be read before it landed.  What it measured is
// in profiles/r05_panel_gemm_study.txt and NOTES.md round 5 item 18: the LDS -> MFMA loop of this shape alone runs at 0.72-0.80 of
// the fp32 MFMA peak (one wavefront per SIMD), its stores add 15-24 us and its loads 5-20 us to 64-103 us even when the stores are
// spread over the whole next chunk and the B panel is resident: no faster than gemm_xx_kernel (87 / 140 / 73 vs 88 / 144 / 71 us).
//
// Column-panel form of the exact-fp32 K-major projection  C[z] (M x N) = A (M x K) * LN?(B[z]) (K x N)  for SHORT reductions and many
// output rows at the 128 x 128 level (K = 96: qkv 288 <- 96, project_in 510 <- 96, the data gradient 255 <- 96; reference sites
// Net_Restormer.py:25,73,78): there the product writes 3-5 x the bytes it reads and gemm_xx_kernel's time is
// (output bytes at ~5 TB/s) + (K x MFMA rate) — a CU's vector-memory path is in order, the loads of the next tile queue behind the
// 48-KiB store burst of the last one, and with six slabs per tile the ring never holds enough work to cover the drain (NOTES.md
// round 5 item 13; a persistent tile walker was built there and measured the same).
//
// This kernel removes the burst instead of hiding it:
//  * ONE workgroup of four wavefronts per CU walks pixel blocks of 128 columns; wavefront w owns columns [32 w, 32 w + 32) of the
//    block for EVERY output row.  Its slice of the B panel (K x 32 floats, 12 KiB at K = 96) is DMA'd by the wavefront itself into
//    LDS — resident for all row chunks of the block, no barrier (own vmcnt), normalised IN PLACE once (LayerNorm prologue: the same
//    operations per element as gemm_xx_kernel's fragment-read form, so the same bits) — and double-buffered: the next block's panel
//    arrives PBS 1-KiB pieces per step.
//  * The output rows are walked in chunks of MB 32-row blocks (MB 32 x 32 accumulators per wavefront); the K-major weight operand
//    streams through a DEPTH-stage ring of 16-row stages of the chunk (L2-resident: the same K x M matrix for every block), one
//    s_barrier per stage, MB x 8 MFMAs per stage and wavefront.
//  * TWO accumulator sets: while chunk v is multiplied, chunk v - 1 is stored — a fixed number of 4-byte buffer stores per step
//    (128-byte row segments per half wave; rows >= M fall outside the buffer resource and are dropped by the hardware), so every
//    step issues at least PA + PBS + NST vector-memory operations, the ring waits are immediate vmcnt counts, and a stage load never
//    queues behind more than one step's stores.
// Summation order per output element = gemm_xx_kernel's (k ascending, two k per v_mfma_f32_32x32x2_f32): bit-identical results.
#pragma once
#include <hip/hip_runtime.h>
#ifndef PANEL_LOADS            // tuning builds of scripts/micro/panel_gemm.hip: -DPANEL_LOADS=0 / -DPANEL_STORES=0 (results are garbage)
#define PANEL_LOADS 1
#endif
#ifndef PANEL_STORES
#define PANEL_STORES 1
#endif
#include <stdint.h>

namespace rcot_panel {

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* pgptr_t;
typedef __attribute__((address_space(3))) void* plptr_t;

struct PP {
    int M, N, K, Zi, nblk, NC;                 // N = pixels per image (% 128 == 0); nblk = images x N / 128; NC row chunks
    const float* At; long lda;                  // K-major weights [ceil16(K)][lda], zero rows beyond K
    const float* B; long ldb, sBo, sBi;
    float* C; long ldc, sCo, sCi;
    const float* mu; const float* rs; long sLN; // LayerNorm statistics per image [N] (LNP)
    const float* lnw; const float* lnb;
};

template <int N> __device__ __forceinline__ void pwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pfence() { asm volatile("" ::: "memory"); }

template <int MB, int NS> struct PanelCfg {
    static constexpr int W = MB * 32;                       // rows of a chunk = columns of an A stage
    static constexpr int NPA = W / 16;                      // 1-KiB pieces of a 16 x W stage
    static constexpr int PA = (NPA + 3) / 4;                // per wavefront (padded)
    static constexpr int ASTAGE = PA * 4 * 256;             // floats
    static constexpr int DEPTH = MB <= 4 ? 5 : 4;
    static constexpr int KK = NS * 16;
    static constexpr int NPB = KK / 8;                      // 1-KiB pieces of one wavefront's K x 32 slice
    static constexpr int PBS = 2;                           // panel pieces of the NEXT block issued per step
    static constexpr int BSLICE = KK * 32;                  // floats
    static constexpr int NST = (MB * 16 + NS - 1) / NS;     // stores per step (of the previous chunk's MB x 16 registers)
    static constexpr int PER_STEP = PA + PBS + NST;
    static constexpr int NWAIT = (DEPTH - 1) * PER_STEP - PA;   // operations younger than the stage a step waits for
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)DEPTH * ASTAGE + 2 * 4 * BSLICE + 2 * KK + 2 * 4 * 64);
    static_assert(NWAIT <= 63, "vmcnt is a 6-bit field");
};

template <int MB, int NS, bool LNP, bool NTS>
__global__ __launch_bounds__(256, 1) void gemm_panel_kernel(PP p) {
    using Cfg = PanelCfg<MB, NS>;
    constexpr int W = Cfg::W, PA = Cfg::PA, ASTAGE = Cfg::ASTAGE, DEPTH = Cfg::DEPTH, KK = Cfg::KK, NPB = Cfg::NPB, PBS = Cfg::PBS;
    constexpr int BSLICE = Cfg::BSLICE, NST = Cfg::NST;
    constexpr int AUX = NTS ? 2 : 0;                         // nt: the output is read by a later launch and does not fit the L2s
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Aring = lds;
    float* const Bbuf = lds + DEPTH * ASTAGE;                // [2][4 waves][KK][32]
    float* const lnwb = Bbuf + 2 * 4 * BSLICE;               // lnw[KK] | lnb[KK]
    float* const statb = lnwb + 2 * KK;                      // [2][4 waves][mu 32 | rstd 32] of the panel's columns (LNP)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, lk = lane >> 5;
    const int G = gridDim.x;
    const int nbpi = p.N / 128;                              // pixel blocks per image
    const int nmine = ((int)blockIdx.x < p.nblk) ? (p.nblk - 1 - (int)blockIdx.x) / G + 1 : 0;   // pixel blocks blockIdx.x, + G, ...
    if (nmine == 0) return;
    const int NC = p.NC;
    const int SPB = NC * NS;                                 // steps per pixel block
    const int nsteps = nmine * SPB;
    const int nv = nmine * NC;                               // (pixel block, row chunk) pairs in walking order
    const int ldc4 = (int)p.ldc * 4;

    if (LNP) {
        for (int k = tid; k < KK; k += 256) {
            lnwb[k] = k < p.K ? p.lnw[k] : 0.f;
            lnwb[KK + k] = k < p.K ? p.lnb[k] : 0.f;
        }
    }

    // ---- A stage DMA: piece q = wave + 4 h covers floats [256 q, 256 q + 256) of the [16][W] stage image
    int a_row[PA], a_col[PA];
#pragma unroll
    for (int h = 0; h < PA; ++h) {
        int e0 = 256 * (wave + 4 * h) + 4 * lane;
        if (e0 >= 16 * W) e0 -= 16 * W;                      // padding pieces re-read the head of the stage (finite data, never read back)
        a_row[h] = e0 / W;
        a_col[h] = e0 - a_row[h] * W;
    }
    const int lda4 = (int)p.lda - 4;
    auto issueA = [&](int t) {                               // the stage of global step t -> ring slot t % DEPTH
        const int s = t % NS, mc = (t / NS) % NC;
        float* st = Aring + (t % DEPTH) * ASTAGE;
#pragma unroll
        for (int h = 0; h < PA; ++h) {
            int col = mc * W + a_col[h];
            col = col < lda4 ? col : lda4;                   // stay inside the row: columns >= M are never stored
            const float* src = p.At + (long)(16 * s + a_row[h]) * p.lda + col;
            __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(st + (wave + 4 * h) * 256), 16, 0, 0);
        }
    };
    // ---- B panel DMA: this wavefront's slice [KK][32] of a pixel block -> buffer par; piece q = rows 8 q .. 8 q + 7
    const int b_off = (lane >> 3) * (int)p.ldb + 32 * wave + 4 * (lane & 7);
    struct Blk { int zo, zi, n0; };
    auto blk_of = [&](int i) {                               // i-th pixel block of this workgroup
        const int b = blockIdx.x + i * G;
        const int z = b / nbpi;
        Blk r;
        r.n0 = (b - z * nbpi) * 128;
        r.zo = z / p.Zi;
        r.zi = z - r.zo * p.Zi;
        return r;
    };
    auto issueB = [&](const float* base, int par, int q) {   // q < NPB
        float* dst = Bbuf + (par * 4 + wave) * BSLICE + q * 256;
        __builtin_amdgcn_global_load_lds((pgptr_t)(base + (long)(8 * q) * p.ldb + b_off), (plptr_t)dst, 16, 0, 0);
    };
    // the LayerNorm statistics of the panel's columns travel the same way (one 4-byte-per-lane DMA: lanes 0-31 mu, 32-63 rstd): a
    // plain load would make the compiler wait for EVERY outstanding operation where the value is first used
    auto issueS = [&](const Blk& b, int par) {
        const float* src = (lk ? p.rs : p.mu) + b.zo * p.sLN + b.n0 + 32 * wave + lm;
        __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(statb + (par * 4 + wave) * 64), 4, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t nullrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0, 0x00020000);   // no records: every store dropped
    auto dummy = [&]() { __builtin_amdgcn_raw_buffer_store_b32(0, nullrs, 0, 0, 0); };

    // ---- prologue: the first panel whole, the first DEPTH - 1 stages; everything landed before step 0
    Blk nextblk{0, 0, 0};
    {
        const Blk b0 = blk_of(0);
        const float* base = p.B + b0.zo * p.sBo + b0.zi * p.sBi + b0.n0;
#pragma unroll
        for (int q = 0; q < NPB; ++q) issueB(base, 0, q);
#pragma unroll
        for (int t = 0; t < DEPTH - 1; ++t)
            if (t < nsteps) issueA(t);
        if (LNP) issueS(b0, 0);
        pwait_vm<0>();
    }
    __syncthreads();

    pf32x16 acc0[MB], acc1[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.f; acc1[i][r] = 0.f; }
    __amdgpu_buffer_rsrc_t prs = nullrs;                     // where the chunk being STORED goes (the previous pair)
    int pvoff = 0;
    const float* nextB = nullptr;                            // panel base of the NEXT pixel block (nullptr: none)
    int t = 0;                                               // global step

#define RCOT_PANEL_VBLOCK(ACC, OLD)                                                                                              \
    {                                                                                                                             \
        const int blk_i = v / NC, mc = v - blk_i * NC, par_b = blk_i & 1;                                                          \
        const Blk cb = blk_of(blk_i);                                                                                              \
        if (mc == 0) {                                                                                                             \
            if (blk_i + 1 < nmine) {                                                                                               \
                nextblk = blk_of(blk_i + 1);                                                                                       \
                nextB = p.B + nextblk.zo * p.sBo + nextblk.zi * p.sBi + nextblk.n0;                                                \
            } else {                                                                                                               \
                nextB = nullptr;                                                                                                   \
            }                                                                                                                      \
        }                                                                                                                          \
        const float* Bsl = Bbuf + (par_b * 4 + wave) * BSLICE;                                                                     \
        _Pragma("unroll") for (int S = 0; S < NS; ++S, ++t) {                                                                      \
            pwait_vm<Cfg::NWAIT>();                      /* stage t landed (and everything older, this block's panel included) */  \
            __builtin_amdgcn_s_barrier();                /* all pieces of stage t in LDS; ring slot (t - 1) % DEPTH is free */      \
            if (PANEL_LOADS && t + DEPTH - 1 < nsteps) issueA(t + DEPTH - 1);                                                      \
            else { _Pragma("unroll") for (int h = 0; h < PA; ++h) dummy(); }                                                       \
            {                                                                                                                      \
                const int sib = mc * NS + S;             /* step within the pixel block */                                          \
                _Pragma("unroll") for (int u = 0; u < PBS; ++u) {                                                                  \
                    const int q = sib * PBS + u;                                                                                   \
                    if (PANEL_LOADS && nextB != nullptr && q < NPB) issueB(nextB, par_b ^ 1, q);                                   \
                    else if (LNP && nextB != nullptr && q == NPB) issueS(nextblk, par_b ^ 1);                                      \
                    else dummy();                                                                                                  \
                }                                                                                                                  \
            }                                                                                                                      \
            pfence();                                                                                                              \
            if (LNP && S == 0 && mc == 0) {              /* normalise this wavefront's slice of the panel in place, once per block */ \
                float* sl = const_cast<float*>(Bsl);                                                                               \
                const float mu_c = statb[(par_b * 4 + wave) * 64 + lm], rs_c = statb[(par_b * 4 + wave) * 64 + 32 + lm];           \
                _Pragma("unroll 8") for (int j = 0; j < KK / 2; ++j) {                                                             \
                    const int k = 2 * j + lk;                                                                                      \
                    const float x = sl[k * 32 + lm];                                                                               \
                    sl[k * 32 + lm] = (x - mu_c) * rs_c * lnwb[k] + lnwb[KK + k];                                                  \
                }                                                                                                                  \
            }                                                                                                                      \
            const float* As = Aring + (t % DEPTH) * ASTAGE;                                                                        \
            const float* Bs = Bsl + (S * 16) * 32;                                                                                 \
            _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                                     \
                const float bv = Bs[(2 * ks + lk) * 32 + lm];                                                                      \
                float av[MB];                                                                                                      \
                _Pragma("unroll") for (int i = 0; i < MB; ++i) av[i] = As[(2 * ks + lk) * W + i * 32 + lm];                        \
                _Pragma("unroll") for (int i = 0; i < MB; ++i) {                                                                   \
                    if (S == 0 && ks == 0) {                                                                                       \
                        const pf32x16 z0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       \
                        ACC[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, z0, 0, 0, 0);                                     \
                    } else {                                                                                                       \
                        ACC[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, ACC[i], 0, 0, 0);                                 \
                    }                                                                                                              \
                }                                                                                                                  \
                /* this step's share of the previous chunk's registers, behind the MFMAs of the k pair */                          \
                constexpr int per = (NST + 7) / 8;                                                                                 \
                _Pragma("unroll") for (int u = 0; u < per; ++u) {                                                                  \
                    const int jj = ks * per + u;                                                                                   \
                    if (jj < NST) {                                                                                                \
                        const int j = S * NST + jj;                                                                                \
                        if (PANEL_STORES && j < MB * 16) {                                                                         \
                            const int i = j / 16, r = j % 16;                                                                      \
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2);                                                       \
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, OLD[i][r]), prs, pvoff + row * ldc4, 0, AUX); \
                        } else {                                                                                                   \
                            dummy();                                                                                               \
                        }                                                                                                          \
                    }                                                                                                              \
                }                                                                                                                  \
            }                                                                                                                      \
            pfence();                                                                                                              \
        }                                                                                                                          \
        /* this pair is stored during the next one */                                                                              \
        prs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + cb.zo * p.sCo + cb.zi * p.sCi), 0, p.M * ldc4, 0x00020000);          \
        pvoff = ((mc * W + 4 * lk) * (int)p.ldc + cb.n0 + 32 * wave + lm) * 4;                                                     \
    }

    for (int v0 = 0; v0 < nv; v0 += 2) {
        {
            const int v = v0;
            RCOT_PANEL_VBLOCK(acc0, acc1)
        }
        if (v0 + 1 < nv) {
            const int v = v0 + 1;
            RCOT_PANEL_VBLOCK(acc1, acc0)
        }
    }
#undef RCOT_PANEL_VBLOCK
    // ---- the last pair's registers
    if (nv & 1) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc0[i][r]), prs, pvoff + (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4, 0, AUX);
    } else {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc1[i][r]), prs, pvoff + (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4, 0, AUX);
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side
struct PanelArgs {
    int M, N, K, Zo, Zi;
    const float* At; long lda;
    const float* B; long ldb, sBo, sBi;
    float* C; long ldc, sCo, sCi;
    const float* mu; const float* rs; long sLN; const float* lnw; const float* lnb;   // mu == nullptr: no LayerNorm
    int nts;
};

// chunking of M into NC chunks of MB 32-row blocks: (MB, NC) with MB * NC == ceil(M / 32), MB in {3, 4}
inline bool panel_shape(int M, int* MB, int* NC) {
    const int nb = (M + 31) / 32;
    if (nb % 4 == 0) { *MB = 4; *NC = nb / 4; return true; }
    if (nb % 3 == 0) { *MB = 3; *NC = nb / 3; return true; }
    return false;
}

template <int MB, int NS, bool LNP, bool NTS>
inline hipError_t panel_launch(const PP& p, int grid, hipStream_t st) {
    using Cfg = PanelCfg<MB, NS>;
    static bool once = (hipFuncSetAttribute((const void*)gemm_panel_kernel<MB, NS, LNP, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL((gemm_panel_kernel<MB, NS, LNP, NTS>), dim3(grid), dim3(256), Cfg::LDS_BYTES, st, p);
    return hipGetLastError();
}

// returns 0 after launching, -100 when the product is not one this kernel takes
inline int try_gemm_panel(const PanelArgs& a, int num_cus, hipStream_t st) {
    int MB = 0, NC = 0;
    if (a.K != 96 || (a.N & 127) || a.M < 96 || a.M > 512 || !panel_shape(a.M, &MB, &NC)) return -100;
    if ((a.lda & 3) || (a.ldb & 3) || (a.sBo & 3) || (a.sBi & 3) || (reinterpret_cast<uintptr_t>(a.At) & 15) ||
        (reinterpret_cast<uintptr_t>(a.B) & 15) || a.lda < 4)
        return -100;
    if ((long)a.M * a.ldc * 4 >= (1L << 31) || (long)96 * a.ldb * 4 >= (1L << 31)) return -100;      // 32-bit buffer offsets
    const int Z = a.Zo * a.Zi;
    PP p{};
    p.M = a.M; p.N = a.N; p.K = a.K; p.Zi = a.Zi; p.nblk = Z * (a.N / 128); p.NC = NC;
    if (NC * 6 * 2 < 13 + 2 * 5) return -100;                // the next panel (12 pieces + statistics) must be issued DEPTH steps before its block starts
    p.At = a.At; p.lda = a.lda;
    p.B = a.B; p.ldb = a.ldb; p.sBo = a.sBo; p.sBi = a.sBi;
    p.C = a.C; p.ldc = a.ldc; p.sCo = a.sCo; p.sCi = a.sCi;
    p.mu = a.mu; p.rs = a.rs; p.sLN = a.sLN; p.lnw = a.lnw; p.lnb = a.lnb;
    const int grid = p.nblk < num_cus ? p.nblk : num_cus;
    const bool ln = a.mu != nullptr;
    hipError_t e;
#define RCOT_PANEL_GO(MBv)                                                                                     \
    e = ln ? (a.nts ? panel_launch<MBv, 6, true, true>(p, grid, st) : panel_launch<MBv, 6, true, false>(p, grid, st)) \
           : (a.nts ? panel_launch<MBv, 6, false, true>(p, grid, st) : panel_launch<MBv, 6, false, false>(p, grid, st))
    if (MB == 4) { RCOT_PANEL_GO(4); } else { RCOT_PANEL_GO(3); }
#undef RCOT_PANEL_GO
    return e == hipSuccess ? 0 : -1;
}

}  // namespace rcot_panel
