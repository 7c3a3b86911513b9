// Dense k3s1 / k4s2 convolutions of the critic (Net_Restormer.py:443-487) and their data gradients as bf16x3 K-major products on
// the producer / consumer kernel (gemm_x3w.hip, CONV mode): the implicit-GEMM engine spends most of its issue slots on gather index
// arithmetic (VALU 60 % busy, MFMA 35 %, exact fp32: 40-50 TFLOP/s), this form has none.
//
// Geometry.  A "padded plane" holds an H x W image at rows 1..H, columns 4..W+3 of an (H + 2) x (W + 8) array (zeros elsewhere: one
// zero row above and below, >= 1 zero column left and right, row pitch a multiple of 4); a padded, channel-major tensor is
// [C][B planes] flat.  For a GEMM column n = position in that flat index space, output pixel (y, x) = (row - 1, column - 4):
//   k3 s1 p1 forward   : out(y, x) = sum_t W_t in(y + ky - 1, x + kx - 1)          = sum_t W_t  Xp[n + (ky - 1) Wp + (kx - 1)]
//   its data gradient  : dX(y, x)  = sum_t W_t^T dZ(y - ky + 1, x - kx + 1)        = sum_t W_t^T dZp[n - (ky - 1) Wp - (kx - 1)]
//   k4 s2 p1 forward   : out(y, x) = sum_{a,b,i,j} W[2a+i][2b+j] Q_ij(y + a, x + b),  Q_ij(y, x) = in(2y + i - 1, 2x + j - 1)
//                        with the four parity planes Q_ij stored in the OUTPUT geometry:   = sum ... Qp_ij[n + a Wp + b]
//   its data gradient  : dQ_ij(y, x) = sum_{a,b} W[2a+i][2b+j]^T dZ(y - a, x - b)    = sum ... dZp[n - a Wp - b]   (M = 4 Ci rows),
//                        then dX(2y + i - 1, 2x + j - 1) = dQ_ij(y, x)  (merge kernel).
// Every shifted operand row is CONTIGUOUS in n, so a reduction slab is 16 channel rows at a per-tap offset: exactly what the LDS-DMA
// producer streams (sources need only 4-byte alignment).  Outputs at padding positions are computed and dropped (colmap = -1).
#include "common.h"
#include "../../include/rcot_hip.h"

namespace rcot {
int conv_pcm_x3w(const void* Apk, int M, int K, const float* Xp, long ldb, int N, const int* tapoff, int ntaps, const float* bias,
                 float lrelu, const int* colmap, float* Y, long ldy, float* ws, size_t ws_bytes, hipStream_t st);
}

using namespace rcot;

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// mode 0: out[c][b][yp][xp] = X[b][c][yp - 1][xp - 4]                        geometry (H + 2) x (W + 8)
// mode 1: out[(ij, c)][b][yp][xp] = X[b][c][2 (yp - 1) + i - 1][2 (xp - 4) + j - 1]   geometry (H/2 + 2) x (W/2 + 8)
// one thread per 4 consecutive xp of the output; every position of the padded planes is written (zeros outside the image)
__global__ __launch_bounds__(256) void pcm_prep_kernel(const float* __restrict__ X, float* __restrict__ out, long ldo, int B, int C, int H,
                                                       int W, int mode, long total4) {
    const int Ho = mode ? H / 2 : H, Wo = mode ? W / 2 : W;
    const int Wp = Wo + 8, Hp = Ho + 2, W4 = Wp / 4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
        const int x4 = (int)(t % W4);
        long r = t / W4;
        const int yp = (int)(r % Hp);
        r /= Hp;
        const int b = (int)(r % B);
        const int ch = (int)(r / B);                                      // mode 1: ch = ij * C + c
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mode == 0) {
            const int y = yp - 1, x0 = 4 * x4 - 4;
            if (y >= 0 && y < H && x0 >= 0 && x0 < W) v = *reinterpret_cast<const float4*>(X + (((long)b * C + ch) * H + y) * W + x0);
        } else {
            const int ij = ch / C, c = ch - ij * C, i = ij >> 1, j = ij & 1;
            const int y = 2 * (yp - 1) + i - 1;
            if (yp >= 1 && y >= 0 && y < H) {
                const float* row = X + (((long)b * C + c) * H + y) * W;
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xq = 4 * x4 - 4 + q;                         // Q column
                    const int x = 2 * xq + j - 1;
                    e[q] = (xq >= 0 && x >= 0 && x < W) ? row[x] : 0.f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        *reinterpret_cast<float4*>(out + (long)ch * ldo + ((long)b * Hp + yp) * Wp + 4 * x4) = v;
    }
}

// dX[b][c][iy][ix] = dQ[(ij, c)][b][y + 1][x + 4],  i = (iy + 1) & 1, y = (iy + 1) >> 1 (same for x): the parity planes of the k4 s2
// data gradient back to a dense tensor; one thread per 4 consecutive ix
__global__ __launch_bounds__(256) void pcm_merge_kernel(const float* __restrict__ dQ, long ldq, float* __restrict__ dX, int B, int C, int H, int W,
                                                        long total4) {
    const int Wp = W / 2 + 8, Hp = H / 2 + 2, W4 = W / 4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
        const int x4 = (int)(t % W4);
        long r = t / W4;
        const int iy = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int i = (iy + 1) & 1, y = (iy + 1) >> 1;
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ix = 4 * x4 + q, j = (ix + 1) & 1, x = (ix + 1) >> 1;
            e[q] = dQ[((long)(i * 2 + j) * C + c) * ldq + ((long)b * Hp + y + 1) * Wp + x + 4];
        }
        *reinterpret_cast<float4*>(dX + (((long)b * C + c) * H + iy) * W + 4 * x4) = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// pre-split fragment pack of A[m][k] = W[rowoff[m] + koff[k]]  (layout: rcot_pack_weight's WTs): one thread per (slab, row tile, lane)
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ W, const int* __restrict__ rowoff, const int* __restrict__ koff,
                                                        int M, int K, int MT, unsigned char* __restrict__ Apk, long total) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int lane = (int)(t & 63);
        const long q = t >> 6;
        const int mt = (int)(q % MT), slab = (int)(q / MT);
        const int lm = lane & 31, kg = lane >> 5, m = 32 * mt + lm;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 16 * slab + 8 * kg + 2 * e + u;
                a[u] = (m < M && k < K) ? W[rowoff[m] + koff[k]] : 0.f;
            }
            const unsigned h = pk_bf16(a[0], a[1]);
            hi[e] = h;
            const float r0 = a[0] - __builtin_bit_cast(float, h << 16), r1 = a[1] - __builtin_bit_cast(float, h & 0xffff0000u);
            lo[e] = pk_bf16(r0, r1);
        }
        unsigned char* dst = Apk + ((long)slab * MT + mt) * 2048 + lane * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

inline int grid_of(long n) {
    long g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int rcot_conv_pcm_prep(const float* X, float* out, long ldo, int B, int C, int H, int W, int mode, void* stream) {
    if (!X || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1) || (W & 3) || (mode && ((H & 1) || (W & 7))) || (ldo & 3) ||
        (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return RCOT_EINVAL;
    const int Ho = mode ? H / 2 : H, Wo = mode ? W / 2 : W;
    const long total4 = (long)(mode ? 4 * C : C) * B * (Ho + 2) * ((Wo + 8) / 4);
    RCOT_LAUNCH(pcm_prep_kernel, dim3(grid_of(total4)), dim3(256), 0, (hipStream_t)stream, X, out, ldo, B, C, H, W, mode, total4);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_conv_pcm_merge(const float* dQ, long ldq, float* dX, int B, int C, int H, int W, void* stream) {
    if (!dQ || !dX || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (W & 7) || (H & 1) || (reinterpret_cast<uintptr_t>(dX) & 15)) return RCOT_EINVAL;
    const long total4 = (long)B * C * H * (W / 4);
    RCOT_LAUNCH(pcm_merge_kernel, dim3(grid_of(total4)), dim3(256), 0, (hipStream_t)stream, dQ, ldq, dX, B, C, H, W, total4);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_conv_pcm_pack(const float* W, const int* rowoff, const int* koff, int M, int K, void* Apk, void* stream) {
    if (!W || !rowoff || !koff || !Apk || M <= 0 || K <= 0 || (K & 15) || (reinterpret_cast<uintptr_t>(Apk) & 15)) return RCOT_EINVAL;
    const int MT = cdiv(M, 32);
    const long total = (long)(K / 16) * MT * 64;
    RCOT_LAUNCH(conv_pack_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, W, rowoff, koff, M, K, MT,
                       (unsigned char*)Apk, total);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_conv_pcm(const void* Apk, int M, int K, const float* Xp, long ldb, int N, const int* tapoff, int ntaps, const float* bias,
                  float lrelu, const int* colmap, float* Y, long ldy, float* ws, size_t ws_bytes, void* stream) {
    return conv_pcm_x3w(Apk, M, K, Xp, ldb, N, tapoff, ntaps, bias, lrelu, colmap, Y, ldy, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
