// Implicit-GEMM dense convolutions on the MFMA fp32 engine: forward (bias + LeakyReLU, residual,
// PixelShuffle/PixelUnshuffle folded into the store), data gradient, weight gradient (split-K
// over batch*pixels).  Serves the critic's k5s1 / k4s2 / k3s1 stack and the transport map's
// patch-embed / Downsample / Upsample / output 3x3 convolutions.
#include "gemm_core.h"
#include "../../include/rcot_hip.h"

using namespace rcot;

namespace {

using CfgL = TileCfg<128, 128>;
using CfgS = TileCfg<64, 64>;

struct ConvGeom {
    const float* src;      // tensor the gather reads
    int Ci, Co, H, W, OH, OW, KH, KW, stride, pad;
    FastDiv dKHW, dKW, dOW, dW, dP, dHW;
};

// forward: B(k=(ci,ky,kx), n=(oy,ox)) = X[b][ci][oy*s+ky-p][ox*s+kx-p]
struct FwdB {
    typedef ConvGeom P;
    __device__ static __forceinline__ int extent(const P& g) { return g.OH * g.OW; }
    __device__ static __forceinline__ float at(const P& g, int b, int k, int n) {
        uint32_t ci, r, ky, kx, oy, ox;
        g.dKHW.divmod(k, ci, r);
        g.dKW.divmod(r, ky, kx);
        g.dOW.divmod(n, oy, ox);
        const int iy = (int)(oy * g.stride + ky) - g.pad, ix = (int)(ox * g.stride + kx) - g.pad;
        if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) return 0.f;
        return g.src[(((long)b * g.Ci + ci) * g.H + iy) * g.W + ix];
    }
};

// data gradient: A(m=ci, k=(co,ky,kx)) = Wt[co][ci][ky][kx]
struct DgradA {
    typedef ConvGeom P;
    __device__ static __forceinline__ int extent(const P& g) { return g.Ci; }
    __device__ static __forceinline__ float at(const P& g, int, int k, int m) {
        uint32_t co, r;
        g.dKHW.divmod(k, co, r);
        return g.src[((long)co * g.Ci + m) * g.dKHW.d + r];
    }
};
// data gradient: B(k=(co,ky,kx), n=(y,x)) = dY[b][co][(y+p-ky)/s][(x+p-kx)/s] when divisible & in range
struct DgradB {
    typedef ConvGeom P;
    __device__ static __forceinline__ int extent(const P& g) { return g.H * g.W; }
    __device__ static __forceinline__ float at(const P& g, int b, int k, int n) {
        uint32_t co, r, ky, kx, y, x;
        g.dKHW.divmod(k, co, r);
        g.dKW.divmod(r, ky, kx);
        g.dW.divmod(n, y, x);
        const int ty = (int)y + g.pad - (int)ky, tx = (int)x + g.pad - (int)kx;
        if (ty < 0 || tx < 0) return 0.f;
        int oy = ty, ox = tx;
        if (g.stride == 2) {
            if ((ty | tx) & 1) return 0.f;
            oy >>= 1; ox >>= 1;
        }
        if (oy >= g.OH || ox >= g.OW) return 0.f;
        return g.src[(((long)b * g.Co + co) * g.OH + oy) * g.OW + ox];
    }
};

// weight gradient: A(m=co, k=(b,pix)) = dY[b][co][pix]
struct WgradA {
    typedef ConvGeom P;
    __device__ static __forceinline__ int extent(const P& g) { return g.Co; }
    __device__ static __forceinline__ float at(const P& g, int, int k, int m) {
        uint32_t b, pix;
        g.dP.divmod(k, b, pix);
        return g.src[((long)b * g.Co + m) * g.dP.d + pix];
    }
};
// weight gradient: B(k=(b,oy,ox), n=(ci,ky,kx)) = X[b][ci][oy*s+ky-p][ox*s+kx-p]
struct WgradB {
    typedef ConvGeom P;
    __device__ static __forceinline__ int extent(const P& g) { return g.Ci * (int)g.dKHW.d; }
    __device__ static __forceinline__ float at(const P& g, int, int k, int n) {
        uint32_t b, pix, oy, ox, ci, r, ky, kx;
        g.dP.divmod(k, b, pix);
        g.dOW.divmod(pix, oy, ox);
        g.dKHW.divmod(n, ci, r);
        g.dKW.divmod(r, ky, kx);
        const int iy = (int)(oy * g.stride + ky) - g.pad, ix = (int)(ox * g.stride + kx) - g.pad;
        if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) return 0.f;
        return g.src[(((long)b * g.Ci + ci) * g.H + iy) * g.W + ix];
    }
};

ConvGeom make_geom(const float* src, int Ci, int Co, int H, int W, int KH, int KW, int stride, int pad) {
    ConvGeom g{};
    g.src = src;
    g.Ci = Ci; g.Co = Co; g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.OH = (H + 2 * pad - KH) / stride + 1;
    g.OW = (W + 2 * pad - KW) / stride + 1;
    g.dKHW.init(KH * KW); g.dKW.init(KW); g.dOW.init(g.OW); g.dW.init(W);
    g.dP.init(g.OH * g.OW); g.dHW.init(H * W);
    return g;
}

template <class Cfg> using AStrK = StridedLoader<Cfg::BM, Cfg::SA, true>;
template <class Cfg> using BFwd = FunctorLoader<Cfg::BN, Cfg::SB, FwdB, false>;
template <class Cfg> using ADg = FunctorLoader<Cfg::BM, Cfg::SA, DgradA, true>;
template <class Cfg> using BDg = FunctorLoader<Cfg::BN, Cfg::SB, DgradB, false>;
template <class Cfg> using AWg = FunctorLoader<Cfg::BM, Cfg::SA, WgradA, true>;
template <class Cfg> using BWg = FunctorLoader<Cfg::BN, Cfg::SB, WgradB, true>;

bool valid(int B, int Ci, int H, int W, int Co, int KH, int KW, int stride, int pad) {
    return B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && (stride == 1 || stride == 2) &&
           pad >= 0 && H + 2 * pad >= KH && W + 2 * pad >= KW;
}

}  // namespace

extern "C" {

int rcot_conv2d_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int Ci, int H, int W,
                    int Co, int KH, int KW, int stride, int pad, float lrelu, int cmap, const float* R,
                    void* stream) {
    if (!X || !Wt || !Y || !valid(B, Ci, H, W, Co, KH, KW, stride, pad) || cmap < 0 || cmap > 2) return RCOT_EINVAL;
    ConvGeom g = make_geom(X, Ci, Co, H, W, KH, KW, stride, pad);
    if (cmap == 1 && ((g.OH | g.OW) & 1)) return RCOT_EINVAL;
    if (cmap == 2 && (Co & 3)) return RCOT_EINVAL;
    GemmDims d{};
    d.M = Co; d.N = g.OH * g.OW; d.K = Ci * KH * KW; d.Zi = 1; d.S = 1;
    d.kchunk = cdiv(d.K, BK) * BK;
    StridedP ap{Wt, (long)d.K, 1, 0, 0, Co, d.K};
    EpiP ep{};
    ep.C = Y; ep.ldc = d.N; ep.sCo = (long)Co * d.N;       // shuffles permute within the same per-image volume
    ep.R = R; ep.ldr = d.N; ep.sRo = (long)Co * d.N;
    ep.bias = bias;
    ep.alpha = 1.f; ep.beta = 0.f; ep.lrelu = lrelu;
    ep.cmap = cmap; ep.mapW = g.OW; ep.mapH = g.OH;
    if (cmap != 0 && R) return RCOT_EINVAL;
    LaunchPlan pl = plan_gemm(d.M, d.N, d.K, B, false, 0);
    if (pl.big) return launch_gemm_cfg<CfgL, AStrK<CfgL>, StridedP, BFwd<CfgL>, ConvGeom, true>(d, ap, g, ep, B, (hipStream_t)stream);
    return launch_gemm_cfg<CfgS, AStrK<CfgS>, StridedP, BFwd<CfgS>, ConvGeom, true>(d, ap, g, ep, B, (hipStream_t)stream);
}

int rcot_conv2d_dgrad(const float* dY, const float* Wt, float* dX, int B, int Ci, int H, int W, int Co, int KH,
                      int KW, int stride, int pad, float beta, void* stream) {
    if (!dY || !Wt || !dX || !valid(B, Ci, H, W, Co, KH, KW, stride, pad)) return RCOT_EINVAL;
    ConvGeom gb = make_geom(dY, Ci, Co, H, W, KH, KW, stride, pad);
    ConvGeom ga = gb;
    ga.src = Wt;
    GemmDims d{};
    d.M = Ci; d.N = H * W; d.K = Co * KH * KW; d.Zi = 1; d.S = 1;
    d.kchunk = cdiv(d.K, BK) * BK;
    EpiP ep{};
    ep.C = dX; ep.ldc = d.N; ep.sCo = (long)Ci * d.N;
    ep.alpha = 1.f; ep.beta = beta; ep.lrelu = 1.f;
    LaunchPlan pl = plan_gemm(d.M, d.N, d.K, B, false, 0);
    if (pl.big) return launch_gemm_cfg<CfgL, ADg<CfgL>, ConvGeom, BDg<CfgL>, ConvGeom, true>(d, ga, gb, ep, B, (hipStream_t)stream);
    return launch_gemm_cfg<CfgS, ADg<CfgS>, ConvGeom, BDg<CfgS>, ConvGeom, true>(d, ga, gb, ep, B, (hipStream_t)stream);
}

int rcot_conv2d_wgrad(const float* dY, const float* X, float* dWt, int B, int Ci, int H, int W, int Co, int KH,
                      int KW, int stride, int pad, float beta, float* ws, size_t ws_bytes, void* stream) {
    if (!dY || !X || !dWt || !valid(B, Ci, H, W, Co, KH, KW, stride, pad)) return RCOT_EINVAL;
    ConvGeom gb = make_geom(X, Ci, Co, H, W, KH, KW, stride, pad);
    ConvGeom ga = gb;
    ga.src = dY;
    GemmDims d{};
    d.M = Co; d.N = Ci * KH * KW; d.K = B * gb.OH * gb.OW; d.Zi = 1;
    LaunchPlan pl = plan_gemm(d.M, d.N, d.K, 1, ws != nullptr, ws_bytes);
    d.S = pl.S;
    d.kchunk = cdiv(cdiv(d.K, d.S), BK) * BK;
    d.S = cdiv(d.K, d.kchunk);
    d.ws = ws;
    if (d.S > 1 && (size_t)d.M * d.N * d.S * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
    EpiP ep{};
    ep.C = dWt; ep.ldc = d.N;
    ep.alpha = 1.f; ep.beta = beta; ep.lrelu = 1.f;
    if (pl.big) return launch_gemm_cfg<CfgL, AWg<CfgL>, ConvGeom, BWg<CfgL>, ConvGeom, true>(d, ga, gb, ep, 1, (hipStream_t)stream);
    return launch_gemm_cfg<CfgS, AWg<CfgS>, ConvGeom, BWg<CfgS>, ConvGeom, true>(d, ga, gb, ep, 1, (hipStream_t)stream);
}

}  // extern "C"
