// Direct kernels for the THIN dense convolutions of the path: a 3-channel side (RGB) on the OUTPUT of the product —
//   the transport map's output conv 96 -> 3 (forward and weight gradient; Net_Restormer.py:326) and the data
//   gradients that end in the image, critic conv1 64 -> 3 (5x5; Net_Restormer.py:478, needed by the gradient
//   penalty and the generator step) and patch-embed 48 -> 3.
// As implicit GEMMs these have M = 3 and fill 3 of 64 tile rows (2-3 TFLOP/s); here they are plain FMA kernels
// bound by reading the wide operand once.
#include "common.h"
#include "../../include/rcot_hip.h"

namespace rcot {

namespace {

constexpr int TILE_H = 16, TILE_W = 64;

// out[b][co][y][x] = act( bias[co] + R + sum_{ci,ky,kx} w(co,ci,ky,kx) * in[b][ci][y+ky-pad][x+kx-pad] ) (+ beta*old)
// for NCO <= 4 output channels, stride 1, "same" padding.  w(co,ci,ky,kx) = wt[wb + co*sco + ci*sci + ky*sky + kx*skx]:
// the forward passes the native OIHW strides, the data gradient the transposed + 180-degree rotated view of the
// same tensor (negative tap strides).  One workgroup = a 16 x 64 output tile of one image; per input channel the
// (16+KS-1) x (64+KS-1) input window is staged in LDS (double buffered), each thread owns 4 pixels of one row.
template <int KS, int NCO>
__global__ __launch_bounds__(256) void conv_few_out_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                           long wb, long sco, long sci, long sky, long skx,
                                                           const float* __restrict__ bias, const float* __restrict__ R,
                                                           float* __restrict__ out, int Cin, int H, int W, int pad,
                                                           float lrelu, float beta) {
    constexpr int TH = TILE_H + KS - 1, TW = TILE_W + KS - 1, LDT = TW + 1;
    constexpr int CB = 4;                                           // input channels staged per barrier
    __shared__ float tile[2][CB][TH * LDT];
    extern __shared__ __attribute__((aligned(16))) float wl[];      // all weights, [ci][ky][kx][4] (one float4 per tap)
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    for (int e = tid; e < Cin * KS * KS * 4; e += 256) {
        const int c = e & 3, tap = e >> 2;
        const int ci = tap / (KS * KS), r = tap - ci * (KS * KS), ky = r / KS, kx = r - ky * KS;
        wl[e] = c < NCO ? wt[wb + c * sco + ci * sci + ky * sky + kx * skx] : 0.f;
    }
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H, b = blockIdx.z;
    const long hw = (long)H * W;
    const float* inb = in + (long)b * Cin * hw;
    float acc[NCO][4];
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
    // software pipeline: the windows of the NEXT four channels are requested (unconditional loads from clamped offsets, zeroed
    // by selects) BEFORE the FMAs of the current four and written to the other LDS buffer after them.  One channel per barrier
    // left one memory round trip per channel exposed (one wavefront per SIMD here): 76 us for 96 channels at 128x128.
    constexpr int NE = (TH * TW + 255) / 256;
    int soff[NE], goff[NE];                                   // LDS offset (-1: unused) / clamped global offset
    bool gok[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        const int r = e / TW, c = e - r * TW;
        const int gy = y0 + r - pad, gx = x0 + c - pad;
        soff[i] = e < TH * TW ? r * LDT + c : -1;
        gok[i] = e < TH * TW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        goff[i] = gok[i] ? gy * W + gx : 0;
    }
    float nxt[CB][NE];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int q = 0; q < CB; ++q) {
            const float* p = inb + (long)min(c0 + q, Cin - 1) * hw;
#pragma unroll
            for (int i = 0; i < NE; ++i) nxt[q][i] = p[goff[i]];
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < CB; ++q)
#pragma unroll
            for (int i = 0; i < NE; ++i)
                if (soff[i] >= 0) tile[buf][q][soff[i]] = gok[i] ? nxt[q][i] : 0.f;
    };
    fetch(0);
    commit(0);
    for (int c0 = 0, it = 0; c0 < Cin; c0 += CB, ++it) {
        __syncthreads();                                     // tile[it&1] complete; tile[(it+1)&1] no longer read
        if (c0 + CB < Cin) fetch(c0 + CB);
#pragma unroll
        for (int q = 0; q < CB; ++q) {
            const int ci = c0 + q;
            if (ci >= Cin) break;
            const float* t = tile[it & 1][q] + ty * LDT + tx * 4;
            const float4* wc = reinterpret_cast<const float4*>(wl) + ci * (KS * KS);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                float v[KS + 3];
#pragma unroll
                for (int u = 0; u < KS + 3; ++u) v[u] = t[ky * LDT + u];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float4 w4 = wc[ky * KS + kx];                           // broadcast read
                    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int c = 0; c < NCO; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(wv[c], v[kx + j], acc[c][j]);
                }
            }
        }
        if (c0 + CB < Cin) commit((it + 1) & 1);
    }
    const int y = y0 + ty, x = x0 + tx * 4;
    if (y < H && x < W) {                                     // W % 4 == 0
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            const long o = ((long)b * NCO + c) * hw + (long)y * W + x;
            float4 r = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
            if (bias) { const float bb = bias[c]; r.x += bb; r.y += bb; r.z += bb; r.w += bb; }
            if (R) { const float4 q = *reinterpret_cast<const float4*>(R + o); r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
            if (beta != 0.f) {
                const float4 q = *reinterpret_cast<const float4*>(out + o);
                r.x += beta * q.x; r.y += beta * q.y; r.z += beta * q.z; r.w += beta * q.w;
            }
            if (lrelu != 1.f) {
                r.x = r.x > 0.f ? r.x : r.x * lrelu; r.y = r.y > 0.f ? r.y : r.y * lrelu;
                r.z = r.z > 0.f ? r.z : r.z * lrelu; r.w = r.w > 0.f ? r.w : r.w * lrelu;
            }
            *reinterpret_cast<float4*>(out + o) = r;
        }
    }
}

// dW[co][ci][3][3] += sum_{b,y,x} dY[b][co][y][x] * X[b][ci][y+i-1][x+j-1]   for NCO <= 4 output channels (3x3, pad 1).
// Rolling-row strips (as the depthwise backward, pointwise.hip): a thread owns 4 pixels x RS rows of one (b, ci) plane
// of X, keeps three rows in registers and 9*NCO sums; G = lanes that share a plane (256 / 64 / gsub).
struct Row6 { float v[6]; };
__device__ __forceinline__ void load_row6(const float* __restrict__ plane, int H, int W, int y, int x0, Row6& r) {
    if (y < 0 || y >= H) {
#pragma unroll
        for (int j = 0; j < 6; ++j) r.v[j] = 0.f;
        return;
    }
    const float* q = plane + (long)y * W + x0;
    const float4 c = *reinterpret_cast<const float4*>(q);
    r.v[0] = (x0 > 0) ? q[-1] : 0.f;
    r.v[1] = c.x; r.v[2] = c.y; r.v[3] = c.z; r.v[4] = c.w;
    r.v[5] = (x0 + 4 < W) ? q[4] : 0.f;
}

template <int NCO, int G, int RS>
__global__ __launch_bounds__(256) void wgrad_few_out_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ dw, long nthreads, int Cin, int H, int W,
                                                            int gsub) {
    __shared__ float red[4][9 * NCO];
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < nthreads;
    if (!live) t = 0;
    const int wq = W >> 2, ns = (H + RS - 1) / RS;
    const long plane = t / ((long)ns * wq);                   // = b * Cin + ci
    const int rem = (int)(t - plane * (long)ns * wq);
    const int st = rem / wq, y0 = st * RS, x0 = (rem - st * wq) * 4;
    const int b = (int)(plane / Cin), ci = (int)(plane - (long)b * Cin);
    const long hw = (long)H * W;
    const float* xp = x + plane * hw;
    const float* gp = dy + (long)b * NCO * hw;
    float s[NCO][9];
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int i = 0; i < 9; ++i) s[c][i] = 0.f;
    if (live) {
        Row6 x3[3];
        load_row6(xp, H, W, y0 - 1, x0, x3[2]);
        load_row6(xp, H, W, y0, x0, x3[0]);
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int y = y0 + i;
            if (y < H) {
                Row6& xu = x3[(i + 2) % 3]; Row6& xm = x3[i % 3]; Row6& xd = x3[(i + 1) % 3];
                load_row6(xp, H, W, y + 1, x0, xd);
#pragma unroll
                for (int c = 0; c < NCO; ++c) {
                    const float4 gq = *reinterpret_cast<const float4*>(gp + c * hw + (long)y * W + x0);
                    const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            s[c][0 + dj] = fmaf(gv[k], xu.v[k + dj], s[c][0 + dj]);
                            s[c][3 + dj] = fmaf(gv[k], xm.v[k + dj], s[c][3 + dj]);
                            s[c][6 + dj] = fmaf(gv[k], xd.v[k + dj], s[c][6 + dj]);
                        }
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = G >= 64 ? 64 : gsub;
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            float v = s[c][i];
            for (int o = gl >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            s[c][i] = v;
        }
    // dW layout [co][ci][3][3]
    if (G == 256) {
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < NCO; ++c)
#pragma unroll
                for (int i = 0; i < 9; ++i) red[wave][c * 9 + i] = s[c][i];
        }
        __syncthreads();
        if (threadIdx.x < 9 * NCO) {
            const int c = threadIdx.x / 9, i = threadIdx.x - c * 9;
            atomicAdd(&dw[((long)c * Cin + ci) * 9 + i],
                      (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
        }
    } else if ((lane & (gl - 1)) == 0 && live) {
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int i = 0; i < 9; ++i) atomicAdd(&dw[((long)c * Cin + ci) * 9 + i], s[c][i]);
    }
}

}  // namespace

// ---- dispatch helpers used by conv_ops.hip; return -100 when the shape is not one of the thin cases -----------
int try_conv_few_out(const float* in, const float* wt, long wb, long sco, long sci, long sky, long skx, const float* bias,
                     const float* R, float* out, int B, int Cin, int H, int W, int Cout, int KS, int pad, float lrelu,
                     float beta, hipStream_t st) {
    if (Cout != 3 || (KS != 3 && KS != 5) || 2 * pad != KS - 1 || (W & 3) || B > 65535) return -100;
    if ((reinterpret_cast<uintptr_t>(out) & 15) || (R && (reinterpret_cast<uintptr_t>(R) & 15))) return -100;
    const dim3 grid(cdiv(W, TILE_W), cdiv(H, TILE_H), B);
    const size_t smem = sizeof(float) * 4 * (size_t)Cin * KS * KS;
    if (smem > 96 * 1024) return -100;
    if (KS == 3)
        RCOT_LAUNCH((conv_few_out_kernel<3, 3>), grid, dim3(256), smem, st, in, wt, wb, sco, sci, sky, skx, bias, R, out, Cin, H,
                           W, pad, lrelu, beta);
    else
        RCOT_LAUNCH((conv_few_out_kernel<5, 3>), grid, dim3(256), smem, st, in, wt, wb, sco, sci, sky, skx, bias, R, out, Cin, H,
                           W, pad, lrelu, beta);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int try_wgrad_few_out(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout, int KS, int pad,
                      float beta, hipStream_t st) {
    if (Cout != 3 || KS != 3 || pad != 1 || beta != 1.0f || (W & 3) || (H & 3)) return -100;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(x) & 15)) return -100;
    constexpr int RS = 16;
    const int tpp = cdiv(H, RS) * (W >> 2);
    const long nt = (long)B * Cin * tpp;
    const dim3 grid(cdiv(nt, 256));
#define RCOT_WF(G, SUB) RCOT_LAUNCH((wgrad_few_out_kernel<3, G, RS>), grid, dim3(256), 0, st, dy, x, dw, nt, Cin, H, W, SUB)
    if (tpp % 256 == 0) { RCOT_WF(256, 64); }
    else if (tpp % 64 == 0) { RCOT_WF(64, 64); }
    else if (tpp < 64 && (tpp & (tpp - 1)) == 0) { RCOT_WF(1, tpp); }
    else return -100;
#undef RCOT_WF
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // namespace rcot
