// fp32 MFMA GEMM engine for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, == fmaf chain).
//
//   C[z] (M x N) = epilogue( sum_k A[z](m,k) * B[z](k,n) )
//
// One workgroup = 4 wavefronts (2x2) computing a BM x BN tile, BK = 16 per LDS stage,
// two LDS stages with register-staged prefetch (global -> VGPR while the MFMAs of the
// current stage run, VGPR -> LDS afterwards, one barrier per K-tile).
// LDS images are K-major ( As[k][m], Bs[k][n] ) so that the 32x32x2 operand fetch
// (lane l: m|n = l&31, k = l>>5) is a conflict-free ds_read_b32 per half-wave.
// Operands are supplied by small loader structs (how to get tile elements from HBM):
// the same engine serves the 1x1 projections (NCHW pixels are the N axis), the
// H*W-reductions (Gram, weight gradients: K = pixels, split-K slabs), and the
// implicit-GEMM convolutions of the critic / resamplers.
#pragma once
#include "common.h"

namespace rcot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int LDS_PAD = 4;
constexpr int GEMM_NT = 256;

template <int BM_, int BN_, int WM_ = 2, int WN_ = 2>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;   // WM x WN = 4 wavefronts
    static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN); // 32x32 MFMA tiles per wave
    static_assert(WM * WN == 4 && TM * 32 * WM == BM && TN * 32 * WN == BN, "tile/wave grid mismatch");
    static constexpr int SA = BM + LDS_PAD, SB = BN + LDS_PAD;
    static constexpr int STAGE = BK * SA + BK * SB;        // floats per LDS stage
};

// ------------------------------------------------------------------ epilogue
struct EpiP {
    float* C;
    long ldc, sCo, sCi;
    const float* R;           // optional residual / second operand, same (m,n) indexing
    long ldr, sRo, sRi;
    const float* rowscale;    // optional per-row multiplier of R
    long sSo, sSi;
    const float* bias;        // optional per-row bias
    float alpha, beta, lrelu; // out = act(alpha*acc + bias + rowscale*R + beta*C_old); lrelu==1 -> identity
    int transC;               // C(m,n) stored at n*ldc + m
    int cmap;                 // 0 none, 1 PixelUnshuffle(2), 2 PixelShuffle(2) folded into the store,
                              // 3 parity scatter: n = (y',x') of the (zo>>1, zo&1) parity class of a 2H x 2W plane
    int mapW;                 // conv output width for cmap (N = mapH*mapW)
    int mapH;
    int vec;                  // C (and R) rows are 16-byte aligned: the lean epilogue may use float4 accesses
    int fold;                 // the GEMM N axis is (image, pixel): n -> (b = n / foldP, pixel); sCo / sRo step per image
    FastDiv foldP;
    const float* mask;        // optional (epi_store only), same layout as C: out = mask > 0 ? out : out * mslope — the LeakyReLU
    float mslope;             // backward (rcot_lrelu_bwd) of the tensor the result is multiplied into, folded into the store
    int nts;                  // gemm_xx_kernel: streaming (non-temporal) tile stores
    int one;                  // split-bf16 kernels only (RCOT_PREC_BF16X1): the hi*hi product alone — the two cross products are skipped
};

__device__ __forceinline__ void epi_store(const EpiP& e, int zo, int zi, int m, int n, float v) {
    long coff = zo * e.sCo + zi * e.sCi, roff = zo * e.sRo + zi * e.sRi;
    if (e.fold) {
        uint32_t b, pix;
        e.foldP.divmod((uint32_t)n, b, pix);
        coff = (long)b * e.sCo;
        roff = (long)b * e.sRo;
        n = (int)pix;
    }
    v *= e.alpha;
    if (e.bias) v += e.bias[m];
    if (e.R) {
        float r = e.R[roff + (long)m * e.ldr + n];
        if (e.rowscale) r *= e.rowscale[zo * e.sSo + zi * e.sSi + m];
        v += r;
    }
    long addr;
    if (e.cmap == 0) {
        addr = e.transC ? ((long)n * e.ldc + m) : ((long)m * e.ldc + n);
    } else {
        const int y = n / e.mapW, x = n - y * e.mapW;
        if (e.cmap == 1) {   // out[4m + 2(y&1) + (x&1)][y/2][x/2], plane (H/2)*(W/2)
            const int ch = 4 * m + 2 * (y & 1) + (x & 1);
            addr = ((long)ch * (e.mapH >> 1) + (y >> 1)) * (e.mapW >> 1) + (x >> 1);
        } else if (e.cmap == 2) {   // out[m/4][2y + (m/2&1)][2x + (m&1)], plane (2H)*(2W)
            const int ch = m >> 2, i = (m >> 1) & 1, j = m & 1;
            addr = ((long)ch * (2 * e.mapH) + (2 * y + i)) * (2 * e.mapW) + (2 * x + j);
        } else {             // out[m][2y + py][2x + px]
            addr = ((long)m * (2 * e.mapH) + (2 * y + (zo >> 1))) * (2 * e.mapW) + (2 * x + (zo & 1));
        }
    }
    float* p = e.C + coff + addr;
    if (e.beta != 0.f) v += e.beta * (*p);
    if (e.lrelu != 1.f) v = v > 0.f ? v : v * e.lrelu;
    if (e.mask) v = e.mask[coff + addr] > 0.f ? v : v * e.mslope;       // exactly lrelu_bwd_kernel's expression
    *p = v;
}

// ------------------------------------------------------------------ vectorised lean epilogue
// The 32x32 MFMA accumulator holds, per lane, 16 values of ONE output column; storing it directly costs 16 scalar
// 4-byte stores per tile and the epilogue becomes store-issue bound.  Instead every wavefront transposes one
// 32x32 tile at a time through 4 KiB of private LDS and then moves whole 16-byte pieces: lane l handles row
// (l>>3) + 8*pass and columns 4*(l&7)..+3, so 8 lanes cover one 128-byte row segment and a wave-instruction
// covers 8 rows.  out = alpha*acc + rowscale[m]*R + beta*C_old.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_vec(f32x16 (&acc)[TM][TN], float* scr, float* Cb, long ldc, const float* Rb,
                                             long ldr, const float* Sb, float alpha, float beta, int mbase, int nbase,
                                             int M, int N, int lane, bool nts = false) {
    const int lm = lane & 31, lk = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    // Every addend (residual R, or the old C when only beta is set) of the wave's whole region is requested FIRST:
    // one HBM round trip per tile instead of one per 32x32 piece (the epilogue was latency-, not bandwidth-bound).
    const float* Ab = Rb ? Rb : (beta != 0.f ? Cb : nullptr);
    const long lda_ = Rb ? ldr : ldc;
    const float amul = Rb ? 1.f : beta;
    float4 q[TM][TN][4];
    float sc[TM][4];
    if (Ab) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int m = mbase + i * 32 + ps * 8 + rr;
                sc[i][ps] = (Rb && Sb && m < M) ? Sb[m] : amul;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = nbase + j * 32 + c4;
                    q[i][j][ps] = (m < M && n < N) ? *reinterpret_cast<const float4*>(Ab + (long)m * lda_ + n)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
    }
    const bool both = Rb && beta != 0.f;      // rare: residual AND accumulate into C
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + lm] = acc[i][j][r];
            const int n = nbase + j * 32 + c4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + rr;
                const int m = mbase + i * 32 + row;
                float4 v = *reinterpret_cast<const float4*>(scr + row * 32 + c4);
                if (m < M && n < N) {
                    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
                    if (Ab) {
                        const float4 a = q[i][j][ps];
                        const float s_ = sc[i][ps];
                        v.x += s_ * a.x; v.y += s_ * a.y; v.z += s_ * a.z; v.w += s_ * a.w;
                    }
                    float* dst = Cb + (long)m * ldc + n;
                    if (both) {
                        const float4 o = *reinterpret_cast<const float4*>(dst);
                        v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
                    }
                    // nts: the tile is read by a LATER launch and the whole output does not fit the L2s — a streaming store keeps the
                    // operand panels the neighbouring row tiles re-read in cache (EpiP::nts, set by the entry point from the output size)
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
}

// ------------------------------------------------------------------ loaders
// Each loader: init(P, zo, zi, tile origin, tid) ; fetch(k0, kend) global->regs ; commit(lds) regs->LDS.
// EXT = BM or BN, LD = LDS row stride (SA or SB).  The LDS image is [k][x] (x = m or n).

// x-contiguous operand: elem(k, x) = base + k*ld + x, float4 along x.  (activations as B of a
// 1x1 conv: k = channel, x = pixel).  Optional fused WithBias-LayerNorm prologue:
// value = (v - mu[x]) * rs[x] * w[k] + b[k].
struct XContigP {
    const float* base;
    long ld, so, si;
    int X, K;                 // extents
    const float* mu;          // LayerNorm stats per x (pixel), per outer batch (stride sLN); nullptr = off
    const float* rs;
    long sLN;
    const float* lnw;         // per k (channel)
    const float* lnb;
};
template <int EXT, int LD>
struct XContigLoader {
    static constexpr bool KTAB = false;
    static constexpr int NV = EXT * BK / 4 / GEMM_NT;      // float4 per thread (2 for 128, 1 for 64)
    static constexpr int XQ = EXT / 4;
    static_assert(EXT * BK % (4 * GEMM_NT) == 0 && GEMM_NT % XQ == 0, "XContig needs EXT in {64,128}");
    const float* p;
    long ld;
    int x4, xvalid, K;
    const float *lnw, *lnb;
    bool ln;
    float4 mu4, rs4;
    float4 v[NV];
    int kk[NV];
    __device__ __forceinline__ void init(const XContigP& P, int zo, int zi, int x0, int tid) {
        p = P.base + zo * P.so + zi * P.si;
        ld = P.ld;
        K = P.K;
        x4 = (tid % XQ) * 4;
        const int gx = x0 + x4;
        xvalid = P.X - gx;           // >=4: full vector
        p += gx;
#pragma unroll
        for (int i = 0; i < NV; ++i) kk[i] = (tid + i * GEMM_NT) / XQ;
        ln = P.mu != nullptr;
        lnw = P.lnw;
        lnb = P.lnb;
        if (ln) {
            const float* m = P.mu + zo * P.sLN + gx;
            const float* r = P.rs + zo * P.sLN + gx;
            if (xvalid >= 4) {
                mu4 = *reinterpret_cast<const float4*>(m);
                rs4 = *reinterpret_cast<const float4*>(r);
            } else {
                mu4 = make_float4(0, 0, 0, 0);
                rs4 = make_float4(0, 0, 0, 0);
                if (xvalid > 0) { mu4.x = m[0]; rs4.x = r[0]; }
                if (xvalid > 1) { mu4.y = m[1]; rs4.y = r[1]; }
                if (xvalid > 2) { mu4.z = m[2]; rs4.z = r[2]; }
            }
        }
    }
    __device__ __forceinline__ void fetch(int k0, int kend) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = k0 + kk[i];
            float4 t = make_float4(0, 0, 0, 0);
            if (k < kend) {
                const float* q = p + (long)k * ld;
                if (xvalid >= 4) {
                    t = *reinterpret_cast<const float4*>(q);
                } else {
                    if (xvalid > 0) t.x = q[0];
                    if (xvalid > 1) t.y = q[1];
                    if (xvalid > 2) t.z = q[2];
                }
                if (ln) {
                    const float w = lnw[k], b = lnb[k];
                    t.x = (t.x - mu4.x) * rs4.x * w + b;
                    t.y = (t.y - mu4.y) * rs4.y * w + b;
                    t.z = (t.z - mu4.z) * rs4.z * w + b;
                    t.w = (t.w - mu4.w) * rs4.w * w + b;
                    if (xvalid < 4) {          // keep padding columns exactly zero
                        if (xvalid < 1) t.x = 0.f;
                        if (xvalid < 2) t.y = 0.f;
                        if (xvalid < 3) t.z = 0.f;
                        t.w = 0.f;
                    }
                }
            }
            v[i] = t;
        }
    }
    __device__ __forceinline__ void commit(float* lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(lds + kk[i] * LD + x4) = v[i];
    }
};

// k-contiguous operand: elem(k, x) = base + x*ld + k, float4 along k.  Used where the
// reduction runs over pixels (Gram, weight gradients).  Optional "batch folded into K":
// global k -> (b = k / Kb, kk = k % Kb), address += b*sK (Kb % 16 == 0 required).
// Optional LayerNorm prologue with stats indexed by k (pixel) and affine by x (channel).
struct KContigP {
    const float* base;
    long ld, so, si;
    int X, K;
    int Kb;                   // per-batch K extent when batch is folded into K (0 = off)
    long sK;                  // batch stride
    const float* mu;
    const float* rs;
    long sLNb;                // stats stride per folded batch
    const float* lnw;
    const float* lnb;
};
template <int EXT, int LD>
struct KContigLoader {
    static constexpr bool KTAB = false;
    static constexpr int NV = EXT * BK / 4 / GEMM_NT;
    const float* p;
    long ld, sK, sLNb;
    int k4, X, x0, Kb;
    const float *mu, *rs, *lnw, *lnb;
    float4 v[NV];
    int xx[NV];
    __device__ __forceinline__ void init(const KContigP& P, int zo, int zi, int x0_, int tid) {
        p = P.base + zo * P.so + zi * P.si;
        ld = P.ld; sK = P.sK; sLNb = P.sLNb; Kb = P.Kb;
        X = P.X; x0 = x0_;
        k4 = (tid & 3) * 4;
        mu = P.mu; rs = P.rs; lnw = P.lnw; lnb = P.lnb;
#pragma unroll
        for (int i = 0; i < NV; ++i) xx[i] = (tid + i * GEMM_NT) >> 2;
    }
    __device__ __forceinline__ void fetch(int k0, int kend) {
        int kb = k0, b = 0;
        if (Kb) { b = k0 / Kb; kb = k0 - b * Kb; }
        const float* q0 = p + (long)b * sK + kb + k4;
        const bool kok = (k0 + k4) < kend;     // K extents are multiples of 4 on this path
        float4 m4, r4;
        if (mu && kok) {
            m4 = *reinterpret_cast<const float4*>(mu + (long)b * sLNb + kb + k4);
            r4 = *reinterpret_cast<const float4*>(rs + (long)b * sLNb + kb + k4);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int x = x0 + xx[i];
            float4 t = make_float4(0, 0, 0, 0);
            if (kok && x < X) {
                t = *reinterpret_cast<const float4*>(q0 + (long)x * ld);
                if (mu) {
                    const float w = lnw[x], bb = lnb[x];
                    t.x = (t.x - m4.x) * r4.x * w + bb;
                    t.y = (t.y - m4.y) * r4.y * w + bb;
                    t.z = (t.z - m4.z) * r4.z * w + bb;
                    t.w = (t.w - m4.w) * r4.w * w + bb;
                }
            }
            v[i] = t;
        }
    }
    __device__ __forceinline__ void commit(float* lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float* d = lds + k4 * LD + xx[i];
            d[0] = v[i].x;
            d[LD] = v[i].y;
            d[2 * LD] = v[i].z;
            d[3 * LD] = v[i].w;
        }
    }
};

// Scalar strided operand (weights; arbitrary leading dimensions such as hidden = 127/255/510/1021):
// elem(k, x) = base + x*sx + k*sk.  KFAST chooses which index runs across lanes (coalescing).
struct StridedP {
    const float* base;
    long sx, sk, so, si;
    int X, K;
};
template <int EXT, int LD, bool KFAST>
struct StridedLoader {
    static constexpr bool KTAB = false;
    static constexpr int NE = EXT * BK / GEMM_NT;          // 8 (128), 6 (96) or 4 (64)
    static constexpr int NXS = KFAST ? NE : 1;
    static_assert(EXT * BK % GEMM_NT == 0, "tile not divisible over the workgroup");
    static_assert(KFAST || GEMM_NT % EXT == 0, "x-fast element map needs the workgroup to cover whole rows");
    // element e = tid + i*GEMM_NT:  KFAST: k = e % BK (fixed), x = e / BK (steps by GEMM_NT/BK)
    //                               x-fast: x = e % EXT (fixed), k = e / EXT (steps by GEMM_NT/EXT)
    // The x part of every address is formed once (the 64-bit products used to be redone per element and slab).
    const float* px[NXS];
    bool xok[NXS];
    long sk;
    int kk0, xx0;
    float v[NE];
    bool vok[NE];
    __device__ __forceinline__ void init(const StridedP& P, int zo, int zi, int x0_, int tid_) {
        const float* p = P.base + zo * P.so + zi * P.si;
        sk = P.sk;
        kk0 = KFAST ? (tid_ % BK) : (tid_ / EXT);
        xx0 = KFAST ? (tid_ / BK) : (tid_ % EXT);
#pragma unroll
        for (int i = 0; i < NXS; ++i) {
            const int x = x0_ + xx0 + i * (GEMM_NT / BK);
            xok[i] = x < P.X;
            px[i] = p + (long)(xok[i] ? x : 0) * P.sx;
        }
    }
    __device__ __forceinline__ void fetch(int k0, int kend) {
        if (KFAST) {
            const int k = k0 + kk0;
            const bool kok = k < kend;
            const long ko = (long)(kok ? k : 0) * sk;
#pragma unroll
            for (int i = 0; i < NE; ++i) {                   // (rows past the extent were clamped to row 0, ko to 0: always readable)
                vok[i] = kok && xok[i];
                v[i] = px[i][ko];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int k = k0 + kk0 + i * (GEMM_NT / EXT);
                vok[i] = k < kend && xok[0];
                v[i] = px[0][(long)(k < kend ? k : 0) * sk];
            }
        }
    }
    __device__ __forceinline__ void commit(float* lds) const {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int kk = KFAST ? kk0 : kk0 + i * (GEMM_NT / EXT);
            const int xx = KFAST ? xx0 + i * (GEMM_NT / BK) : xx0;
            lds[kk * LD + xx] = vok[i] ? v[i] : 0.f;
        }
    }
};

// Generic functor operand (implicit-GEMM convolution gathers).  The index decomposition is split so that nothing
// invariant is recomputed inside the K loop (the gathers used to be VALU-bound on integer divisions):
//   F::XS F::px(P, zo, x)   state of the x index (pixel or channel) — computed ONCE per thread, x never changes;
//   F::KS F::pk(P, zo, k)   state of the k index — once per slab (KFAST: the thread's k is fixed within a slab)
//                            or once per element (x-fast operands: four k values per thread and slab);
//   int F::idx(P, KS, XS, bool& ok) element offset into P.src + inside-the-tensor flag.
template <int EXT, int LD, class F, bool KFAST>
struct FunctorLoader {
    // x-fast gathers whose k state is (offset, ky, kx) read it from a per-slab table in LDS (prep(), one slab ahead, 16
    // threads) instead of decomposing four different k per thread and slab
    static constexpr bool KTAB = !KFAST && F::TABLE;
    static constexpr int NE = EXT * BK / GEMM_NT;
    static constexpr int NXS = KFAST ? NE : 1;
    typename F::P P;
    int zo, kk0, xx0;
    typename F::XS xs[NXS];
    bool xok[NXS];
    float v[NE];
    bool vok[NE];                                 // element lies inside the operand: v[i] is real data (else it is whatever sits at offset 0)
    // one element: offset + flag from the functor, an UNCONDITIONAL load (offset 0 when outside), the flag kept for commit()
    __device__ __forceinline__ void load1(int i, bool pre, const typename F::KS& ks, const typename F::XS& x) {
        bool inb;
        const int o = F::idx(P, ks, x, inb);
        vok[i] = pre && inb;
        v[i] = P.src[vok[i] ? o : 0];
    }
    __device__ __forceinline__ void init(const typename F::P& P_, int zo_, int /*zi*/, int x0_, int tid_) {
        P = P_; zo = zo_;
        const int X = F::extent(P_);
        // element e = tid + i*GEMM_NT:  KFAST: k = e % BK (fixed), x = e / BK (steps by GEMM_NT/BK)
        //                               x-fast: x = e % EXT (fixed), k = e / EXT (steps by GEMM_NT/EXT)
        kk0 = KFAST ? (tid_ % BK) : (tid_ / EXT);
        xx0 = KFAST ? (tid_ / BK) : (tid_ % EXT);
#pragma unroll
        for (int i = 0; i < NXS; ++i) {
            const int x = x0_ + xx0 + i * (GEMM_NT / BK);
            xok[i] = x < X;
            xs[i] = F::px(P, zo, xok[i] ? x : 0);
        }
    }
    __device__ __forceinline__ void fetch(int k0, int kend) {
        if (KFAST) {
            const int k = k0 + kk0;
            const bool kok = k < kend;
            const typename F::KS ks = F::pk(P, zo, kok ? k : 0);
#pragma unroll
            for (int i = 0; i < NE; ++i) load1(i, kok && xok[i], ks, xs[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int k = k0 + kk0 + i * (GEMM_NT / EXT);
                const bool kok = k < kend;
                const typename F::KS ks = F::pk(P, zo, kok ? k : 0);
                load1(i, kok && xok[0], ks, xs[0]);
            }
        }
    }
    __device__ __forceinline__ void prep(int k0, int kend, int* tab, int tid) const {
        if constexpr (KTAB) {
            if (tid < BK) {
                const int k = k0 + tid;
                const typename F::KS ks = F::pk(P, zo, k < kend ? k : 0);
                tab[2 * tid] = ks.off;
                tab[2 * tid + 1] = ks.ky | (ks.kx << 16);
            }
        }
    }
    __device__ __forceinline__ void fetch_tab(int k0, int kend, const int* tab) {
        if constexpr (KTAB) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int kk = kk0 + i * (GEMM_NT / EXT);
                const bool kok = (k0 + kk) < kend;
                const int t1 = tab[2 * kk + 1];
                const typename F::KS ks{tab[2 * kk], t1 & 0xffff, t1 >> 16};
                load1(i, kok && xok[0], ks, xs[0]);
            }
        }
    }
    __device__ __forceinline__ void commit(float* lds) const {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int kk = KFAST ? kk0 : kk0 + i * (GEMM_NT / EXT);
            const int xx = KFAST ? xx0 + i * (GEMM_NT / BK) : xx0;
            lds[kk * LD + xx] = vok[i] ? v[i] : 0.f;
        }
    }
};

// ------------------------------------------------------------------ kernel
struct GemmDims {
    int M, N, K;
    int Zi;        // inner batch count: grid z-batch = zo*Zi + zi
    int S;         // split-K factor (slabs)
    int kchunk;    // K per split, multiple of BK
    int tilesM, tilesN;
    float* ws;     // split-K slabs [(z*S+s)][M][N] when S > 1
};

// XCD-aware block remap (bijective): consecutive blocks on one XCD share the same operand panel.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, o = bid >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + o;
}

// GEN = false: lean epilogue (alpha, residual with optional row scale, beta) for the projections / MDTA products;
// GEN = true : full epi_store (bias, LeakyReLU, transposed or pixel-(un)shuffled store) for convs and Linear.
template <class Cfg, class AL, class AP, class BL, class BP, bool GEN>
__global__ __launch_bounds__(GEMM_NT) void gemm_kernel(GemmDims d, AP ap, BP bp, EpiP ep) {
    __shared__ __attribute__((aligned(16))) float lds[2 * Cfg::STAGE];
    constexpr bool BTAB = BL::KTAB;
    __shared__ int ktab[BTAB ? 4 : 1][2 * BK];          // per-slab k-state tables of the B gather (4-slab ring)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;

    const int nblk = d.tilesM * d.tilesN;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int tm = bid % d.tilesM, tn = bid / d.tilesM;      // m-fastest: neighbours share the B panel
    const int zs = blockIdx.z;
    const int z = zs / d.S, s = zs - z * d.S;
    const int zo = z / d.Zi, zi = z - zo * d.Zi;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    const int kbeg = s * d.kchunk;
    const int kend = min(d.K, kbeg + d.kchunk);

    // Wavefronts that own a single 32x32 tile (64x64 workgroup tiles) run with TWO loader instances per operand (even /
    // odd slabs): two slabs are in flight in registers while a third is multiplied from LDS.  A slab is then only 8
    // MFMAs (~0.25 us), far less than the latency of the gathered loads: one slab of lookahead left those waves parked
    // ~30 % of their cycles (PMC).  Larger tiles keep one slab of lookahead (register budget).
    constexpr bool DEEP = Cfg::TM * Cfg::TN == 1;
    AL al0, al1;
    BL bl0, bl1;
    al0.init(ap, zo, zi, m0, tid);
    bl0.init(bp, zo, zi, n0, tid);
    if constexpr (DEEP) {
        al1.init(ap, zo, zi, m0, tid);
        bl1.init(bp, zo, zi, n0, tid);
    }

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    auto prepB = [&](int slab) {                              // table of slab `slab` (threads < BK); visible after the next barrier
        if constexpr (BTAB)
            if (slab < nk) bl0.prep(kbeg + slab * BK, kend, ktab[slab & 3], tid);
    };
    auto fetchB = [&](BL& bl, int slab) {
        if constexpr (BTAB) bl.fetch_tab(kbeg + slab * BK, kend, ktab[slab & 3]);
        else bl.fetch(kbeg + slab * BK, kend);
    };
    if constexpr (BTAB) {
        prepB(0); prepB(1); prepB(2);
        __syncthreads();
    }
    if (nk > 0 || DEEP) {
        al0.fetch(kbeg, kend);
        fetchB(bl0, 0);
    }
    if constexpr (DEEP) {
        al1.fetch(kbeg + BK, kend);             // (zeros when nk < 2: the slab loop below has no conditional parts)
        fetchB(bl1, 1);
    }
    if (nk > 0 || DEEP) {
        al0.commit(lds);
        bl0.commit(lds + BK * Cfg::SA);
    }
    __syncthreads();
    const int lm = lane & 31, lk = lane >> 5;
    auto mma = [&](const float* As) {
        const float* Bs = As + BK * Cfg::SA;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[i] = As[(kk + lk) * Cfg::SA + (wm * Cfg::TM + i) * 32 + lm];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[j] = Bs[(kk + lk) * Cfg::SB + (wn * Cfg::TN + j) * 32 + lm];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    float* S0 = lds;
    float* S1 = lds + Cfg::STAGE;
    if constexpr (DEEP) {
        // NO conditional fetch / commit in this loop (round 4): with the fetch of slab kt + 2 under `if (kt + 2 < nk)` the compiler
        // cannot count the younger loads at the commit of the other register set and drains EVERY outstanding load there
        // (s_waitcnt vmcnt(0): lookahead zero, the memory latency exposed once per slab).  Every loader returns zeros for k >= kend,
        // so slabs past the range are fetched as zeros, and with an odd slab count one of them is multiplied (acc + 0 * 0).
        for (int kt = 0; kt < nk; kt += 2) {
            // slab kt (even) is in S0; registers of loader pair 0 are free, pair 1 holds slab kt+1
            al0.fetch(kbeg + (kt + 2) * BK, kend);
            fetchB(bl0, kt + 2);
            mma(S0);
            prepB(kt + 3);
            al1.commit(S1);
            bl1.commit(S1 + BK * Cfg::SA);
            __syncthreads();
            al1.fetch(kbeg + (kt + 3) * BK, kend);
            fetchB(bl1, kt + 3);
            mma(S1);
            prepB(kt + 4);
            al0.commit(S0);
            bl0.commit(S0 + BK * Cfg::SA);
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1) < nk;
            if (more) {
                al0.fetch(kbeg + (kt + 1) * BK, kend);
                fetchB(bl0, kt + 1);
            }
            mma((kt & 1) ? S1 : S0);
            prepB(kt + 3);
            if (more) {
                float* Ad = (kt & 1) ? S0 : S1;
                al0.commit(Ad);
                bl0.commit(Ad + BK * Cfg::SA);
            }
            __syncthreads();
        }
    }

    // epilogue: acc[i][j][r] -> row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31.
    // Row-major outer loop so that everything that depends only on the row (pointers, bias, row scale) is
    // computed once per row; a half-wave stores 32 consecutive floats (128 B) of one output row.
    const int mrow0 = m0 + wm * Cfg::TM * 32 + 4 * lk;
    const int ncol0 = n0 + wn * Cfg::TN * 32 + lm;
    if (d.S > 1) {
        float* wsb = d.ws + (long)zs * d.M * d.N;
        if ((d.N & 3) == 0) {          // slabs are 16-byte aligned by construction
            __syncthreads();           // the staging ring is free: reuse it as per-wave transpose scratch
            epilogue_vec<Cfg::TM, Cfg::TN>(acc, lds + wave * 1024, wsb, d.N, nullptr, 0, nullptr, 1.f, 0.f,
                                           m0 + wm * Cfg::TM * 32, n0 + wn * Cfg::TN * 32, d.M, d.N, lane);
            return;
        }
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (m >= d.M) continue;
                float* row = wsb + (long)m * d.N;
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    const int n = ncol0 + j * 32;
                    if (n < d.N) row[n] = acc[i][j][r];
                }
            }
    } else if (!GEN) {
        float* Cb = ep.C + zo * ep.sCo + zi * ep.sCi;
        const float* Rb = ep.R ? ep.R + zo * ep.sRo + zi * ep.sRi : nullptr;
        const float* Sb = ep.rowscale ? ep.rowscale + zo * ep.sSo + zi * ep.sSi : nullptr;
        const bool has_beta = ep.beta != 0.f;
        if (ep.vec) {
            __syncthreads();
            epilogue_vec<Cfg::TM, Cfg::TN>(acc, lds + wave * 1024, Cb, ep.ldc, Rb, ep.ldr, Sb, ep.alpha, ep.beta,
                                           m0 + wm * Cfg::TM * 32, n0 + wn * Cfg::TN * 32, d.M, d.N, lane);
            return;
        }
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (m >= d.M) continue;
                float* crow = Cb + (long)m * ep.ldc;
                const float* rrow = Rb ? Rb + (long)m * ep.ldr : nullptr;
                const float rsc = Sb ? Sb[m] : 1.f;
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    const int n = ncol0 + j * 32;
                    if (n >= d.N) continue;
                    float v = acc[i][j][r] * ep.alpha;
                    if (rrow) v += rsc * rrow[n];
                    if (has_beta) v += ep.beta * crow[n];
                    crow[n] = v;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                const int n = ncol0 + j * 32;
                if (n >= d.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < d.M) epi_store(ep, zo, zi, m, n, acc[i][j][r]);
                }
            }
    }
}

// Sum the split-K slabs and apply the epilogue.
// 64 outputs per workgroup (one per lane); the 4 wavefronts take interleaved quarters of the slabs, 8 loads in
// flight each, fixed combination order (deterministic).
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmDims d, EpiP ep, int Z) {
    __shared__ float part[4][64];
    const long mn = (long)d.M * d.N;
    const long total = mn * Z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long idx = base + lane;
        float a = 0.f;
        int z = 0, m = 0, n = 0;
        if (idx < total) {
            z = (int)(idx / mn);
            const long r = idx - (long)z * mn;
            m = (int)(r / d.N);
            n = (int)(r - (long)m * d.N);
            const float* w = d.ws + (long)z * d.S * mn + r;
            // 8 loads in flight per wavefront (the kernel is latency-bound: up to 64 slabs per wavefront), fixed order
            float acc8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc8[q] = 0.f;
            int s = wave;
            for (; s + 28 < d.S; s += 32) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc8[q] += w[(long)(s + 4 * q) * mn];
            }
            for (; s < d.S; s += 4) acc8[0] += w[(long)s * mn];
            a = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
        }
        __syncthreads();
        part[wave][lane] = a;
        __syncthreads();
        if (wave == 0 && idx < total)
            epi_store(ep, z / d.Zi, z % d.Zi, m, n, (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
    }
}

// The same sum for S <= 8 with one output per THREAD and all slabs in flight (the form above gives a slab to each of four
// wavefronts and combines through LDS: for few slabs most of the workgroup idles, 16 us per launch on the critic's small layers).
// The summation ORDER is exactly the one above — wave w there holds (v[w] + v[w + 4]), combined as (p0 + p1) + (p2 + p3) —
// so results are bit-identical (the critic amplifies last-bit differences).
static __global__ __launch_bounds__(256) void splitk_reduce_few_kernel(GemmDims d, EpiP ep, int Z) {
    const long mn = (long)d.M * d.N;
    const long total = mn * Z;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int z = (int)(idx / mn);
        const long r = idx - (long)z * mn;
        const int m = (int)(r / d.N), n = (int)(r - (long)m * d.N);
        const float* w = d.ws + (long)z * d.S * mn + r;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = u < d.S ? w[(long)u * mn] : 0.f;
        const float a = ((v[0] + v[4]) + (v[1] + v[5])) + ((v[2] + v[6]) + (v[3] + v[7]));
        epi_store(ep, z / d.Zi, z % d.Zi, m, n, a);
    }
}

// The same for four consecutive columns per thread (16-byte slab loads and stores, 32-bit index arithmetic — the scalar form spends
// most of its time in two 64-bit divisions per element): the plain stores of the convolutions (no residual, no transposed /
// shuffled store; batch fold with a pixel count that is a multiple of 4).  Per element the same operations in the same order as
// splitk_reduce_few_kernel + epi_store: results are bit-identical.
static __global__ __launch_bounds__(256) void splitk_reduce_few4_kernel(GemmDims d, EpiP ep, int Z) {
    const unsigned n4 = (unsigned)d.N >> 2, mn4 = (unsigned)d.M * n4, total = mn4 * (unsigned)Z;
    const long mn = (long)d.M * d.N;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned z = idx / mn4, r4 = idx - z * mn4;
        const unsigned m = r4 / n4, n = (r4 - m * n4) << 2;
        const float* w = d.ws + (long)z * d.S * mn + (long)m * d.N + n;
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = u < d.S ? *reinterpret_cast<const float4*>(w + (long)u * mn) : make_float4(0.f, 0.f, 0.f, 0.f);
        float a[4];
        a[0] = ((v[0].x + v[4].x) + (v[1].x + v[5].x)) + ((v[2].x + v[6].x) + (v[3].x + v[7].x));
        a[1] = ((v[0].y + v[4].y) + (v[1].y + v[5].y)) + ((v[2].y + v[6].y) + (v[3].y + v[7].y));
        a[2] = ((v[0].z + v[4].z) + (v[1].z + v[5].z)) + ((v[2].z + v[6].z) + (v[3].z + v[7].z));
        a[3] = ((v[0].w + v[4].w) + (v[1].w + v[5].w)) + ((v[2].w + v[6].w) + (v[3].w + v[7].w));
        const int zo = (int)z / d.Zi, zi = (int)z % d.Zi;
        long coff = zo * ep.sCo + zi * ep.sCi;
        unsigned nn = n;
        if (ep.fold) {
            uint32_t b, pix;
            ep.foldP.divmod(n, b, pix);
            coff = (long)b * ep.sCo;
            nn = pix;
        }
        float* p = ep.C + coff + (long)m * ep.ldc + nn;
        const float bia = ep.bias ? ep.bias[m] : 0.f;
        float4 old = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (ep.beta != 0.f) old = *reinterpret_cast<const float4*>(p);
        if (ep.mask) mk = *reinterpret_cast<const float4*>(ep.mask + coff + (long)m * ep.ldc + nn);
        const float o[4] = {old.x, old.y, old.z, old.w}, k[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = a[q] * ep.alpha;
            if (ep.bias) x += bia;
            if (ep.beta != 0.f) x += ep.beta * o[q];
            if (ep.lrelu != 1.f) x = x > 0.f ? x : x * ep.lrelu;
            if (ep.mask) x = k[q] > 0.f ? x : x * ep.mslope;
            a[q] = x;
        }
        *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    }
}
// the cases splitk_reduce_few4_kernel takes
inline bool reduce4_ok(const GemmDims& d, const EpiP& e, int Z) {
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((d.N & 3) || e.R || e.transC || e.cmap || (e.ldc & 3) || (e.sCo & 3) || (e.sCi & 3) || !a16(e.C) || !a16(d.ws)) return false;
    if (e.fold && (e.foldP.d & 3)) return false;
    if (e.mask && !a16(e.mask)) return false;
    return (long)d.M * d.N * Z < (1L << 31);
}

// ------------------------------------------------------------------ host-side launch
// float4 epilogue is legal when every row start of C (and R) is 16-byte aligned
inline bool epi_vec_ok(const EpiP& e, int N) {
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((N & 3) || (e.ldc & 3) || (e.sCo & 3) || (e.sCi & 3) || !a16(e.C) || e.transC || e.cmap || e.fold) return false;
    if (e.R && ((e.ldr & 3) || (e.sRo & 3) || (e.sRi & 3) || !a16(e.R))) return false;
    return true;
}

struct LaunchPlan {
    bool big;      // 128x128 tile (else 64x64)
    int S;
};

// Pick tile config and split-K so that the launch has >= ~3 workgroups per CU when possible.
inline LaunchPlan plan_gemm(int M, int N, int K, int Z, bool allow_split, size_t ws_bytes) {
    LaunchPlan p;
    const long blocks_big = (long)cdiv(M, 128) * cdiv(N, 128) * Z;
    // the big tile wastes work when M or N is small; prefer it when both extents fill it
    const double util_big = ((double)M * N) / ((double)cdiv(M, 128) * 128 * cdiv(N, 128) * 128);
    const double util_small = ((double)M * N) / ((double)cdiv(M, 64) * 64 * cdiv(N, 64) * 64);
    p.big = (blocks_big >= 384) && (util_big >= 0.8 * util_small);
    const int bm = p.big ? 128 : 64;
    const long blocks = (long)cdiv(M, bm) * cdiv(N, bm) * Z;
    p.S = 1;
    if (allow_split) {
        const int nkt = cdiv(K, BK);
        // (RCOT_SPLIT_BLOCKS: tuning knob — the grid size a split aims at; default 768 = three workgroups per CU)
        static const long target = getenv("RCOT_SPLIT_BLOCKS") ? atol(getenv("RCOT_SPLIT_BLOCKS")) : 768;
        long S = (target + blocks - 1) / blocks;
        if (S > nkt / 2) S = nkt / 2;          // keep >= 2 K-tiles per split
        if (S < 1) S = 1;
        const size_t per = (size_t)M * N * Z * sizeof(float);
        while (S > 1 && per * S > ws_bytes) --S;
        if ((long)Z * S > 65535) S = 65535 / Z;
        if (S < 1) S = 1;
        p.S = (int)S;
    }
    return p;
}

template <class Cfg, class AL, class AP, class BL, class BP, bool GEN = false>
inline int launch_gemm_cfg(GemmDims d, const AP& ap, const BP& bp, const EpiP& ep, int Z, hipStream_t st) {
    if (!GEN && (ep.bias || ep.lrelu != 1.f || ep.transC || ep.cmap || ep.fold)) return RCOT_EINVAL;
    EpiP epv = ep;
    epv.vec = epi_vec_ok(ep, d.N);
    d.tilesM = cdiv(d.M, Cfg::BM);
    d.tilesN = cdiv(d.N, Cfg::BN);
    dim3 grid(d.tilesM * d.tilesN, 1, Z * d.S);
    note_kernel("gemm_kernel<TileCfg<%d, %d, ..>, ..> (general engine)", Cfg::BM, Cfg::BN);
    RCOT_LAUNCH((gemm_kernel<Cfg, AL, AP, BL, BP, GEN>), grid, dim3(GEMM_NT), 0, st, d, ap, bp, epv);
    RCOT_LAUNCH_CHECK();
    if (d.S > 1) {
        const long total = (long)d.M * d.N * Z;
        if (d.S <= 8 && reduce4_ok(d, ep, Z)) {
            long nb = (total / 4 + 255) / 256;
            if (nb > 8192) nb = 8192;
            RCOT_LAUNCH(splitk_reduce_few4_kernel, dim3((int)nb), dim3(256), 0, st, d, ep, Z);
        } else if (d.S <= 8) {
            long nb = (total + 255) / 256;
            if (nb > 8192) nb = 8192;
            RCOT_LAUNCH(splitk_reduce_few_kernel, dim3((int)nb), dim3(256), 0, st, d, ep, Z);
        } else {
            long nb = (total + 63) / 64;
            if (nb > 8192) nb = 8192;
            RCOT_LAUNCH(splitk_reduce_kernel, dim3((int)nb), dim3(256), 0, st, d, ep, Z);
        }
        RCOT_LAUNCH_CHECK();
    }
    return RCOT_OK;
}

}  // namespace rcot
