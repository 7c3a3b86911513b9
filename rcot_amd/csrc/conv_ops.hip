// Implicit-GEMM dense convolutions on the MFMA fp32 engine: forward (bias + LeakyReLU, residual,
// PixelShuffle/PixelUnshuffle folded into the store), data gradient, weight gradient.
// Serves the critic's k5s1 / k4s2 / k3s1 stack and the transport map's patch-embed / Downsample /
// Upsample / output 3x3 convolutions.
//
// GEMM view: M = output channels, N = (image, output pixel) — the batch is FOLDED into N so that the deep
// critic layers (4x4 .. 16x16 pixels per image) still fill 64/128-wide tiles — and K = (ci, ky, kx); when
// M*N gives too few workgroups the K loop is split into slabs (deterministic reduce).  The stride-2 data
// gradient is decomposed by OUTPUT PARITY: each of the 4 parity classes of dX only ever sees 2x2 of the
// 4x4 taps, so it is run as 4 dense sub-problems with K = Co*4 instead of one with K = Co*16 of which 3/4
// would multiply zeros.
#include "gemm_core.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace rcot {   // thin (3-channel output side) direct kernels, conv_thin.hip; -100 = not one of those shapes
int try_conv_few_out(const float* in, const float* wt, long wb, long sco, long sci, long sky, long skx, const float* bias,
                     const float* R, float* out, int B, int Cin, int H, int W, int Cout, int KS, int pad, float lrelu,
                     float beta, hipStream_t st);
int try_wgrad_few_out(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout, int KS, int pad,
                      float beta, hipStream_t st);
}  // namespace rcot

namespace {

using CfgL = TileCfg<128, 128>;
using CfgS = TileCfg<64, 64>;

struct ConvGeom {
    const float* src;      // tensor the gather reads
    int B, Ci, Co, H, W, OH, OW, KH, KW, stride, pad;
    FastDiv dKHW, dKW, dOW, dW, dP, dHW, dW2, dHW2;
};

// Every functor splits its gather into px() (x index: once per thread), pk() (k index: once per slab or element)
// and get() (bounds + load) — see FunctorLoader.  Offsets are ints: every tensor has < 2^31 elements (checked).
struct TapK { int off, ky, kx; };        // channel offset + filter tap
struct PixX { int off, y, x; };          // image offset + (pre-shifted) pixel coordinates
struct OffS { int off; };

// forward: B(k=(ci,ky,kx), n=(b,oy,ox)) = X[b][ci][oy*s+ky-p][ox*s+kx-p]
struct FwdB {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = true;     // KS = TapK: eligible for the per-slab LDS table
    typedef TapK KS;
    typedef PixX XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.B * (int)g.dP.d; }
    __device__ static __forceinline__ KS pk(const P& g, int, int k) {
        uint32_t ci, r, ky, kx;
        g.dKHW.divmod(k, ci, r);
        g.dKW.divmod(r, ky, kx);
        return KS{(int)ci * g.H * g.W, (int)ky, (int)kx};
    }
    __device__ static __forceinline__ XS px(const P& g, int, int n) {
        uint32_t b, pix, oy, ox;
        g.dP.divmod(n, b, pix);
        g.dOW.divmod(pix, oy, ox);
        return XS{(int)b * g.Ci * g.H * g.W, (int)(oy * g.stride) - g.pad, (int)(ox * g.stride) - g.pad};
    }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) {
        const int iy = x.y + k.ky, ix = x.x + k.kx;
        if ((unsigned)iy >= (unsigned)g.H || (unsigned)ix >= (unsigned)g.W) return 0.f;
        return g.src[x.off + k.off + iy * g.W + ix];
    }
};

// stride-1 data gradient: A(m=ci, k=(co,ky,kx)) = Wt[co][ci][ky][kx]
struct DgradA {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = false;
    typedef OffS KS;
    typedef OffS XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.Ci; }
    __device__ static __forceinline__ KS pk(const P& g, int, int k) {
        uint32_t co, r;
        g.dKHW.divmod(k, co, r);
        return KS{(int)co * g.Ci * (int)g.dKHW.d + (int)r};
    }
    __device__ static __forceinline__ XS px(const P& g, int, int m) { return XS{m * (int)g.dKHW.d}; }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) { return g.src[k.off + x.off]; }
};
// stride-1 data gradient: B(k=(co,ky,kx), n=(b,y,x)) = dY[b][co][y+p-ky][x+p-kx]
struct DgradB {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = true;     // KS = TapK: eligible for the per-slab LDS table
    typedef TapK KS;
    typedef PixX XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.B * (int)g.dHW.d; }
    __device__ static __forceinline__ KS pk(const P& g, int, int k) {
        uint32_t co, r, ky, kx;
        g.dKHW.divmod(k, co, r);
        g.dKW.divmod(r, ky, kx);
        return KS{(int)co * g.OH * g.OW, (int)ky, (int)kx};
    }
    __device__ static __forceinline__ XS px(const P& g, int, int n) {
        uint32_t b, pix, y, x;
        g.dHW.divmod(n, b, pix);
        g.dW.divmod(pix, y, x);
        return XS{(int)b * g.Co * g.OH * g.OW, (int)y + g.pad, (int)x + g.pad};
    }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) {
        const int oy = x.y - k.ky, ox = x.x - k.kx;
        if ((unsigned)oy >= (unsigned)g.OH || (unsigned)ox >= (unsigned)g.OW) return 0.f;
        return g.src[x.off + k.off + oy * g.OW + ox];
    }
};

// stride-2 (k4) data gradient, parity class cls = 2*py + px:  taps ky = ky0 + 2*jy, ky0 = (py + pad) & 1
// A(m=ci, k=(co,jy,jx)) = Wt[co][ci][ky][kx]
struct Dgrad2A {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = false;
    typedef OffS KS;
    typedef OffS XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.Ci; }
    __device__ static __forceinline__ KS pk(const P& g, int cls, int k) {
        const int co = k >> 2, jy = (k >> 1) & 1, jx = k & 1;
        const int ky = (((cls >> 1) + g.pad) & 1) + 2 * jy, kx = (((cls & 1) + g.pad) & 1) + 2 * jx;
        return KS{co * g.Ci * 16 + ky * 4 + kx};
    }
    __device__ static __forceinline__ XS px(const P&, int, int m) { return XS{m * 16}; }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) { return g.src[k.off + x.off]; }
};
// B(k=(co,jy,jx), n=(b,y',x')) = dY[b][co][(2y'+py+p-ky)/2][(2x'+px+p-kx)/2]
struct Dgrad2B {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = true;     // KS = TapK: eligible for the per-slab LDS table
    typedef TapK KS;
    typedef PixX XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.B * (int)g.dHW2.d; }
    __device__ static __forceinline__ KS pk(const P& g, int cls, int k) {
        const int co = k >> 2, jy = (k >> 1) & 1, jx = k & 1;
        const int ky = (((cls >> 1) + g.pad) & 1) + 2 * jy, kx = (((cls & 1) + g.pad) & 1) + 2 * jx;
        return KS{co * g.OH * g.OW, ky, kx};
    }
    __device__ static __forceinline__ XS px(const P& g, int cls, int n) {
        uint32_t b, pix, y2, x2;
        g.dHW2.divmod(n, b, pix);
        g.dW2.divmod(pix, y2, x2);
        return XS{(int)b * g.Co * g.OH * g.OW, 2 * (int)y2 + (cls >> 1) + g.pad, 2 * (int)x2 + (cls & 1) + g.pad};
    }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) {
        const int ty = x.y - k.ky, tx = x.x - k.kx;                 // even by construction
        if (ty < 0 || tx < 0) return 0.f;
        const int oy = ty >> 1, ox = tx >> 1;
        if (oy >= g.OH || ox >= g.OW) return 0.f;
        return g.src[x.off + k.off + oy * g.OW + ox];
    }
};

// weight gradient: A(m=co, k=(b,pix)) = dY[b][co][pix]
struct WgradA {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = false;
    typedef OffS KS;
    typedef OffS XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.Co; }
    __device__ static __forceinline__ KS pk(const P& g, int, int k) {
        uint32_t b, pix;
        g.dP.divmod(k, b, pix);
        return KS{(int)b * g.Co * (int)g.dP.d + (int)pix};
    }
    __device__ static __forceinline__ XS px(const P& g, int, int m) { return XS{m * (int)g.dP.d}; }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) { return g.src[k.off + x.off]; }
};
// weight gradient: B(k=(b,oy,ox), n=(ci,ky,kx)) = X[b][ci][oy*s+ky-p][ox*s+kx-p]
struct WgradB {
    typedef ConvGeom P;
    [[maybe_unused]] static constexpr bool TABLE = false;
    typedef PixX KS;
    typedef TapK XS;
    __device__ static __forceinline__ int extent(const P& g) { return g.Ci * (int)g.dKHW.d; }
    __device__ static __forceinline__ KS pk(const P& g, int, int k) {
        uint32_t b, pix, oy, ox;
        g.dP.divmod(k, b, pix);
        g.dOW.divmod(pix, oy, ox);
        return KS{(int)b * g.Ci * g.H * g.W, (int)(oy * g.stride) - g.pad, (int)(ox * g.stride) - g.pad};
    }
    __device__ static __forceinline__ XS px(const P& g, int, int n) {
        uint32_t ci, r, ky, kx;
        g.dKHW.divmod(n, ci, r);
        g.dKW.divmod(r, ky, kx);
        return XS{(int)ci * g.H * g.W, (int)ky, (int)kx};
    }
    __device__ static __forceinline__ float get(const P& g, const KS& k, const XS& x) {
        const int iy = k.y + x.ky, ix = k.x + x.kx;
        if ((unsigned)iy >= (unsigned)g.H || (unsigned)ix >= (unsigned)g.W) return 0.f;
        return g.src[k.off + x.off + iy * g.W + ix];
    }
};

ConvGeom make_geom(const float* src, int B, int Ci, int Co, int H, int W, int KH, int KW, int stride, int pad) {
    ConvGeom g{};
    g.src = src;
    g.B = B; g.Ci = Ci; g.Co = Co; g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.OH = (H + 2 * pad - KH) / stride + 1;
    g.OW = (W + 2 * pad - KW) / stride + 1;
    g.dKHW.init(KH * KW); g.dKW.init(KW); g.dOW.init(g.OW); g.dW.init(W);
    g.dP.init(g.OH * g.OW); g.dHW.init(H * W);
    g.dW2.init(W >> 1 ? W >> 1 : 1); g.dHW2.init((H >> 1) * (W >> 1) ? (H >> 1) * (W >> 1) : 1);
    return g;
}

template <class Cfg> using AStrK = StridedLoader<Cfg::BM, Cfg::SA, true>;
template <class Cfg> using BFwd = FunctorLoader<Cfg::BN, Cfg::SB, FwdB, false>;
template <class Cfg> using ADg = FunctorLoader<Cfg::BM, Cfg::SA, DgradA, true>;
template <class Cfg> using BDg = FunctorLoader<Cfg::BN, Cfg::SB, DgradB, false>;
template <class Cfg> using ADg2 = FunctorLoader<Cfg::BM, Cfg::SA, Dgrad2A, true>;
template <class Cfg> using BDg2 = FunctorLoader<Cfg::BN, Cfg::SB, Dgrad2B, false>;
template <class Cfg> using AWg = FunctorLoader<Cfg::BM, Cfg::SA, WgradA, true>;
template <class Cfg> using BWg = FunctorLoader<Cfg::BN, Cfg::SB, WgradB, true>;

bool valid(int B, int Ci, int H, int W, int Co, int KH, int KW, int stride, int pad) {
    if (!(B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && (stride == 1 || stride == 2) && pad >= 0 &&
          H + 2 * pad >= KH && W + 2 * pad >= KW))
        return false;
    const long OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const long lim = (1L << 31) - 1;                          // the gathers index with 32-bit offsets
    return (long)B * Ci * H * W <= lim && (long)B * Co * OH * OW <= lim && (long)Co * Ci * KH * KW <= lim;
}

// tile + split-K plan shared by the three conv GEMMs
void plan_conv(GemmDims& d, int Z, float* ws, size_t ws_bytes, bool& big) {
    LaunchPlan pl = plan_gemm(d.M, d.N, d.K, Z, ws != nullptr, ws_bytes);
    big = pl.big;
    d.S = pl.S;
    d.kchunk = cdiv(cdiv(d.K, d.S), BK) * BK;
    d.S = cdiv(d.K, d.kchunk);
    d.ws = ws;
}

}  // namespace

extern "C" {

int rcot_conv2d_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int Ci, int H, int W,
                    int Co, int KH, int KW, int stride, int pad, float lrelu, int cmap, const float* R, const float* mask,
                    float mslope, float* ws, size_t ws_bytes, void* stream) {
    if (!X || !Wt || !Y || !valid(B, Ci, H, W, Co, KH, KW, stride, pad) || cmap < 0 || cmap > 2) return RCOT_EINVAL;
    if (mask && cmap != 0) return RCOT_EINVAL;
    ConvGeom g = make_geom(X, B, Ci, Co, H, W, KH, KW, stride, pad);
    if (cmap == 1 && ((g.OH | g.OW) & 1)) return RCOT_EINVAL;
    if (cmap == 2 && (Co & 3)) return RCOT_EINVAL;
    if (cmap != 0 && R) return RCOT_EINVAL;
    if (stride == 1 && KH == KW && cmap == 0 && !mask) {     // RGB output: direct kernel
        const int rc = try_conv_few_out(X, Wt, 0, (long)Ci * KH * KW, KH * KW, KW, 1, bias, R, Y, B, Ci, H, W, Co, KH, pad, lrelu,
                                        0.f, (hipStream_t)stream);
        if (rc != -100) return rc;
    }
    const int P = g.OH * g.OW;
    if ((long)B * P > 0x7fffffffL) return RCOT_EINVAL;
    GemmDims d{};
    d.M = Co; d.N = B * P; d.K = Ci * KH * KW; d.Zi = 1;
    StridedP ap{Wt, (long)d.K, 1, 0, 0, Co, d.K};
    EpiP ep{};
    ep.C = Y; ep.ldc = P; ep.sCo = (long)Co * P;            // shuffles permute within the same per-image volume
    ep.R = R; ep.ldr = P; ep.sRo = (long)Co * P;
    ep.bias = bias;
    ep.alpha = 1.f; ep.beta = 0.f; ep.lrelu = lrelu;
    ep.cmap = cmap; ep.mapW = g.OW; ep.mapH = g.OH;
    ep.fold = 1; ep.foldP.init(P);
    ep.mask = mask; ep.mslope = mslope;
    bool big;
    plan_conv(d, 1, ws, ws_bytes, big);
    if (d.S > 1 && (size_t)d.M * d.N * d.S * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
    if (big) return launch_gemm_cfg<CfgL, AStrK<CfgL>, StridedP, BFwd<CfgL>, ConvGeom, true>(d, ap, g, ep, 1, (hipStream_t)stream);
    return launch_gemm_cfg<CfgS, AStrK<CfgS>, StridedP, BFwd<CfgS>, ConvGeom, true>(d, ap, g, ep, 1, (hipStream_t)stream);
}

int rcot_conv2d_dgrad(const float* dY, const float* Wt, float* dX, int B, int Ci, int H, int W, int Co, int KH,
                      int KW, int stride, int pad, float beta, const float* mask, float mslope, float* ws, size_t ws_bytes,
                      void* stream) {
    if (!dY || !Wt || !dX || !valid(B, Ci, H, W, Co, KH, KW, stride, pad)) return RCOT_EINVAL;
    if (stride == 1 && KH == KW && !mask) {                   // gradient w.r.t. an RGB image: the transposed, rotated filter
        const int rc = try_conv_few_out(dY, Wt, (long)KH * KW - 1, KH * KW, (long)Ci * KH * KW, -KW, -1, nullptr, nullptr, dX, B, Co,
                                        H, W, Ci, KH, pad, 1.f, beta, (hipStream_t)stream);
        if (rc != -100) return rc;
    }
    ConvGeom gb = make_geom(dY, B, Ci, Co, H, W, KH, KW, stride, pad);
    ConvGeom ga = gb;
    ga.src = Wt;
    if ((long)B * H * W > 0x7fffffffL) return RCOT_EINVAL;
    EpiP ep{};
    ep.C = dX; ep.sCo = (long)Ci * H * W;
    ep.alpha = 1.f; ep.beta = beta; ep.lrelu = 1.f;
    ep.fold = 1;
    ep.mask = mask; ep.mslope = mslope;
    GemmDims d{};
    d.M = Ci; d.Zi = 1;
    bool big;
    if (stride == 2) {
        if (KH != 4 || KW != 4 || (H & 1) || (W & 1)) return RCOT_EINVAL;   // the critic's k4s2p1 only
        const int P2 = (H >> 1) * (W >> 1);
        d.N = B * P2; d.K = Co * 4;
        ep.cmap = 3; ep.mapH = H >> 1; ep.mapW = W >> 1; ep.foldP.init(P2);
        plan_conv(d, 4, ws, ws_bytes, big);
        if (d.S > 1 && (size_t)d.M * d.N * 4 * d.S * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
        if (big) return launch_gemm_cfg<CfgL, ADg2<CfgL>, ConvGeom, BDg2<CfgL>, ConvGeom, true>(d, ga, gb, ep, 4, (hipStream_t)stream);
        return launch_gemm_cfg<CfgS, ADg2<CfgS>, ConvGeom, BDg2<CfgS>, ConvGeom, true>(d, ga, gb, ep, 4, (hipStream_t)stream);
    }
    d.N = B * H * W; d.K = Co * KH * KW;
    ep.ldc = (long)H * W; ep.foldP.init(H * W);
    plan_conv(d, 1, ws, ws_bytes, big);
    if (d.S > 1 && (size_t)d.M * d.N * d.S * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
    if (big) return launch_gemm_cfg<CfgL, ADg<CfgL>, ConvGeom, BDg<CfgL>, ConvGeom, true>(d, ga, gb, ep, 1, (hipStream_t)stream);
    return launch_gemm_cfg<CfgS, ADg<CfgS>, ConvGeom, BDg<CfgS>, ConvGeom, true>(d, ga, gb, ep, 1, (hipStream_t)stream);
}

int rcot_conv2d_wgrad(const float* dY, const float* X, float* dWt, int B, int Ci, int H, int W, int Co, int KH,
                      int KW, int stride, int pad, float beta, float* ws, size_t ws_bytes, void* stream) {
    if (!dY || !X || !dWt || !valid(B, Ci, H, W, Co, KH, KW, stride, pad)) return RCOT_EINVAL;
    if (stride == 1 && KH == KW) {
        const int rc = try_wgrad_few_out(dY, X, dWt, B, Ci, H, W, Co, KH, pad, beta, (hipStream_t)stream);
        if (rc != -100) return rc;
    }
    ConvGeom gb = make_geom(X, B, Ci, Co, H, W, KH, KW, stride, pad);
    ConvGeom ga = gb;
    ga.src = dY;
    GemmDims d{};
    d.M = Co; d.N = Ci * KH * KW; d.K = B * gb.OH * gb.OW; d.Zi = 1;
    bool big;
    plan_conv(d, 1, ws, ws_bytes, big);
    if (d.S > 1 && (size_t)d.M * d.N * d.S * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
    EpiP ep{};
    ep.C = dWt; ep.ldc = d.N;
    ep.alpha = 1.f; ep.beta = beta; ep.lrelu = 1.f;
    if (big) return launch_gemm_cfg<CfgL, AWg<CfgL>, ConvGeom, BWg<CfgL>, ConvGeom, true>(d, ga, gb, ep, 1, (hipStream_t)stream);
    return launch_gemm_cfg<CfgS, AWg<CfgS>, ConvGeom, BWg<CfgS>, ConvGeom, true>(d, ga, gb, ep, 1, (hipStream_t)stream);
}

}  // extern "C"
