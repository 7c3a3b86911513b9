// The pointwise pieces of the reference's OLDER transport map (MPRNet-style Net.T_net, Net.py:19-216) around its
// 3x3 / 1x1 convolutions (those run on the convolution engine of conv_ops.hip): the shared-slope PReLU (Net.py:185), the
// squeeze-and-excite gate of a channel-attention block (CALayer, Net.py:36-52; CAB :56-73) and the bilinear x0.5 / x2 resampling of
// DownSample / SkipUpSample (:146-176).  All of it is HBM-bound row or plane work: one pass per tensor, float4 where the row length
// and the pointers allow, per-row scalars (gate, mean gradient) read once per workgroup.  gfx950 only.
#include "../../include/rcot_hip.h"
#include "common.h"

using namespace rcot;

namespace {

inline int grid_for(long n, int bs = 256, int cap = 8192) {
    long g = (n + bs - 1) / bs;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ------------------------------------------------------------------ PReLU with ONE learnable slope (nn.PReLU(), Net.py:185)
// y = x > 0 ? x : a x.  nv float4 groups, then the scalar tail [4 nv, n).
// (y may be x: no __restrict__ on the pair)
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* x, const float* __restrict__ slope, float* y, long nv, long n) {
    const float a = slope[0];
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
    for (long i = t0; i < nv; i += step) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = v.x > 0.f ? v.x : a * v.x;
        v.y = v.y > 0.f ? v.y : a * v.y;
        v.z = v.z > 0.f ? v.z : a * v.z;
        v.w = v.w > 0.f ? v.w : a * v.w;
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (long i = 4 * nv + t0; i < n; i += step) {
        const float v = x[i];
        y[i] = v > 0.f ? v : a * v;
    }
}

// dx = x > 0 ? dy : a dy;  part[block] = sum over the block's elements of (x > 0 ? 0 : x dy)   (d/da of a x)
// (dx may be dy)
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* dy, const float* __restrict__ x, const float* __restrict__ slope,
                                                        float* dx, float* __restrict__ part, long nv, long n) {
    __shared__ float red[4];
    const float a = slope[0];
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long i = t0; i < nv; i += step) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        s += (v.x > 0.f ? 0.f : v.x * g.x) + (v.y > 0.f ? 0.f : v.y * g.y) + (v.z > 0.f ? 0.f : v.z * g.z) +
             (v.w > 0.f ? 0.f : v.w * g.w);
        g.x = v.x > 0.f ? g.x : a * g.x;
        g.y = v.y > 0.f ? g.y : a * g.y;
        g.z = v.z > 0.f ? g.z : a * g.z;
        g.w = v.w > 0.f ? g.w : a * g.w;
        reinterpret_cast<float4*>(dx)[i] = g;
    }
    for (long i = 4 * nv + t0; i < n; i += step) {
        const float v = x[i], g = dy[i];
        s += v > 0.f ? 0.f : v * g;
        dx[i] = v > 0.f ? g : a * g;
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// dst[0] += sum part[0..n) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ dst) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) dst[0] += s;
}

// ------------------------------------------------------------------ per-row reductions and per-row scalars ([rows][N] contiguous)
// out[row] = scale * sum_n a[row][n] * (b ? b[row][n] : 1)
template <bool VEC>
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, int N, float scale) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const float* pa = a + row * N;
    const float* pb = b ? b + row * N : nullptr;
    float s = 0.f;
    if (VEC) {
        for (int n = threadIdx.x * 4; n < N; n += 1024) {
            const float4 u = *reinterpret_cast<const float4*>(pa + n);
            if (pb) {
                const float4 v = *reinterpret_cast<const float4*>(pb + n);
                s += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
            } else {
                s += (u.x + u.y) + (u.z + u.w);
            }
        }
    } else {
        for (int n = threadIdx.x; n < N; n += 256) s += pb ? pa[n] * pb[n] : pa[n];
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) out[row] = s * scale;
}

// out[row][n] = a[row][n] * s[row] + (x ? x[row][n] : 0) + (t ? t[row] * tscale : 0);  blockIdx.y walks the row in 1024-element pieces
template <bool VEC>
// (out may be a or x)
__global__ __launch_bounds__(256) void row_scale_add_kernel(const float* a, const float* __restrict__ s, const float* x,
                                                            const float* __restrict__ t, float tscale, float* out, int N) {
    const long row = blockIdx.x;
    const float sc = s[row], add = t ? t[row] * tscale : 0.f;
    const long base = row * N;
    if (VEC) {
        for (int n = (blockIdx.y * 256 + threadIdx.x) * 4; n < N; n += gridDim.y * 1024) {
            float4 u = *reinterpret_cast<const float4*>(a + base + n);
            u.x = u.x * sc + add; u.y = u.y * sc + add; u.z = u.z * sc + add; u.w = u.w * sc + add;
            if (x) {
                const float4 v = *reinterpret_cast<const float4*>(x + base + n);
                u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
            }
            *reinterpret_cast<float4*>(out + base + n) = u;
        }
    } else {
        for (int n = blockIdx.y * 256 + threadIdx.x; n < N; n += gridDim.y * 256) {
            float u = a[base + n] * sc + add;
            if (x) u += x[base + n];
            out[base + n] = u;
        }
    }
}

// ------------------------------------------------------------------ the gate of a channel-attention layer (CALayer.conv_du, Net.py:42-47)
constexpr int CA_MAXC = 1024, CA_MAXR = 256;

// one workgroup per image: hid = relu(W1 mean), gate = sigmoid(W2 hid);  W1 [Cr][C], W2 [C][Cr]
__global__ __launch_bounds__(256) void ca_gate_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, float* __restrict__ hid,
                                                          float* __restrict__ gate, int C, int Cr) {
    __shared__ float m[CA_MAXC], h[CA_MAXR];
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < C; k += 256) m[k] = mean[(long)b * C + k];
    __syncthreads();
    for (int j = threadIdx.x; j < Cr; j += 256) {
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += W1[j * C + k] * m[k];
        s = s > 0.f ? s : 0.f;
        h[j] = s;
        hid[(long)b * Cr + j] = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) s += W2[c * Cr + j] * h[j];
        gate[(long)b * C + c] = 1.f / (1.f + expf(-s));
    }
}

// Backward of the gate in two small launches (a first form — ONE workgroup walking the images in turn, every weight-gradient element
// read-modify-written per image — took 119 us per call, 9 % of the configs[0] iteration: profiles/r06_mprnet_kstats_first.txt).
// 1) one workgroup per image:  ds = dgate * gate (1 - gate);  dhid = (hid > 0) W2^T ds;  dmean = W1^T dhid;  ds, dhid kept for 2)
__global__ __launch_bounds__(256) void ca_gate_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ hid, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, float* __restrict__ dmean,
                                                          float* __restrict__ ds_out, float* __restrict__ dh_out, int C, int Cr) {
    __shared__ float ds[CA_MAXC], dh[CA_MAXR], part[4][CA_MAXR];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gate[(long)b * C + c];
        const float v = dgate[(long)b * C + c] * (g * (1.f - g));
        ds[c] = v;
        ds_out[(long)b * C + c] = v;
    }
    __syncthreads();
    // dhid[j] = sum_c W2[c][j] ds[c]: lanes along j (contiguous in W2), the four wavefronts take every fourth c
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int j0 = 0; j0 < Cr; j0 += 64) {
        const int j = j0 + l;
        float s = 0.f;
        if (j < Cr)
            for (int c = w; c < C; c += 4) s += W2[c * Cr + j] * ds[c];
        if (j < Cr) part[w][j] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Cr; j += 256) {
        const float s = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
        const float v = hid[(long)b * Cr + j] > 0.f ? s : 0.f;
        dh[j] = v;
        dh_out[(long)b * Cr + j] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C; k += 256) {                    // dmean[k] = sum_j W1[j][k] dhid[j] (lanes along k: contiguous)
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) s += W1[j * C + k] * dh[j];
        dmean[(long)b * C + k] = s;
    }
}

// 2) dW2[c][j] += sum_b ds[b][c] hid[b][j];  dW1[j][k] += sum_b dhid[b][j] mean[b][k]   (one thread per element, images in order)
__global__ __launch_bounds__(256) void ca_gate_wgrad_kernel(const float* __restrict__ ds, const float* __restrict__ dh,
                                                            const float* __restrict__ hid, const float* __restrict__ mean,
                                                            float* __restrict__ dW1, float* __restrict__ dW2, int B, int C, int Cr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * Cr) return;
    const int c = i / Cr, j = i - c * Cr;                            // dW2 [C][Cr]
    const int j1 = i / C, k = i - j1 * C;                            // dW1 [Cr][C]
    float s2 = 0.f, s1 = 0.f;
    for (int b = 0; b < B; ++b) {
        s2 += ds[(long)b * C + c] * hid[(long)b * Cr + j];
        s1 += dh[(long)b * Cr + j1] * mean[(long)b * C + k];
    }
    dW2[i] += s2;
    dW1[i] += s1;
}

// ------------------------------------------------------------------ bilinear resampling, align_corners=False (nn.Upsample, Net.py:149,158,167)
// scale 0.5: source coordinate 2 o + 0.5 -> the mean of the 2 x 2 cell
__global__ __launch_bounds__(256) void down2_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int Ho, int Wo) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const long r = i / Wo;
        const int oy = (int)(r % Ho);
        const long p = r / Ho;
        const float* s = x + (p * (2 * Ho) + 2 * oy) * (long)(2 * Wo) + 2 * ox;
        const float2 a = *reinterpret_cast<const float2*>(s), b = *reinterpret_cast<const float2*>(s + 2 * Wo);
        y[i] = 0.5f * (0.5f * a.x + 0.5f * a.y) + 0.5f * (0.5f * b.x + 0.5f * b.y);
    }
}

// adjoint: dx[2 oy + i][2 ox + j] = beta dx + 0.25 dy[oy][ox]
__global__ __launch_bounds__(256) void down2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long total, int Ho,
                                                        int Wo, float beta) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const long r = i / Wo;
        const int oy = (int)(r % Ho);
        const long p = r / Ho;
        const float g = 0.25f * dy[i];
        float* d = dx + (p * (2 * Ho) + 2 * oy) * (long)(2 * Wo) + 2 * ox;
        float2 a = make_float2(g, g), b = make_float2(g, g);
        if (beta != 0.f) {
            const float2 u = *reinterpret_cast<const float2*>(d), v = *reinterpret_cast<const float2*>(d + 2 * Wo);
            a.x += beta * u.x; a.y += beta * u.y; b.x += beta * v.x; b.y += beta * v.y;
        }
        *reinterpret_cast<float2*>(d) = a;
        *reinterpret_cast<float2*>(d + 2 * Wo) = b;
    }
}

// scale 2: source coordinate max(0.5 (o + 0.5) - 0.5, 0); neighbours i0, i1 = i0 + (i0 < n - 1), weights 1 - l, l (PyTorch's formula)
__device__ __forceinline__ void up2_src(int o, int n, int& i0, int& i1, float& l0, float& l1) {
    float s = 0.5f * ((float)o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// y [2H][2W] = up2(x [H][W]) + (skip ? skip : 0)
__global__ __launch_bounds__(256) void up2_kernel(const float* __restrict__ x, const float* __restrict__ skip, float* __restrict__ y,
                                                  long total, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const long r = i / Wo;
        const int oy = (int)(r % Ho);
        const long p = r / Ho;
        int y0, y1, x0, x1;
        float h0, h1, w0, w1;
        up2_src(oy, H, y0, y1, h0, h1);
        up2_src(ox, W, x0, x1, w0, w1);
        const float* s = x + p * (long)H * W;
        float v = h0 * (w0 * s[(long)y0 * W + x0] + w1 * s[(long)y0 * W + x1]) + h1 * (w0 * s[(long)y1 * W + x0] + w1 * s[(long)y1 * W + x1]);
        if (skip) v += skip[i];
        y[i] = v;
    }
}

// weight with which output o of a x2 axis reads input i
__device__ __forceinline__ float up2_w(int o, int i, int n) {
    int i0, i1;
    float l0, l1;
    up2_src(o, n, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// adjoint as a gather: dx[iy][ix] = sum over the <= 4 x 4 outputs that read it
__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long total, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W);
        const long r = i / W;
        const int iy = (int)(r % H);
        const long p = r / H;
        const float* g = dy + p * (long)Ho * Wo;
        float wx[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ox = 2 * ix - 1 + b;
            wx[b] = (ox >= 0 && ox < Wo) ? up2_w(ox, ix, W) : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = 2 * iy - 1 + a;
            if (oy < 0 || oy >= Ho) continue;
            const float wy = up2_w(oy, iy, H);
            float t = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = 2 * ix - 1 + b;
                const int oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
                t += wx[b] * g[(long)oy * Wo + oxc];
            }
            s += wy * t;
        }
        dx[i] = s;
    }
}

// ------------------------------------------------------------------ transposed, rotated copies of convolution weights
// Wf[ci][co][KH-1-ky][KW-1-kx] = W[co][ci][ky][kx] for n weights of one shape at element offsets table[2 j] (in `src`) / table[2 j + 1]
// (in `dst`): the stride-1 data gradient of a "same" convolution IS the forward convolution of dY with Wf, so it can run on the
// forward kernel (whose 16-row form takes the 80-channel level).  blockIdx.y = weight, one thread per element.
__global__ __launch_bounds__(256) void weight_flip_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const long long* __restrict__ table, int Co, int Ci, int KH, int KW) {
    const int per = Co * Ci * KH * KW, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const float* W = src + table[2 * blockIdx.y];
    float* Wf = dst + table[2 * blockIdx.y + 1];
    const int khw = KH * KW;
    const int t = i % khw, r = i / khw, ci = r % Ci, co = r / Ci;      // i = ((co * Ci + ci) * KH + ky) * KW + kx
    Wf[((long)ci * Co + co) * khw + (khw - 1 - t)] = W[i];
}

}  // namespace

extern "C" {

int rcot_prelu_fwd(const float* x, const float* slope, float* y, long n, void* stream) {
    if (!x || !slope || !y || n <= 0) return RCOT_EINVAL;
    const long nv = (al16(x) && al16(y)) ? n / 4 : 0;
    RCOT_LAUNCH(prelu_fwd_kernel, dim3(grid_for(nv ? nv : n)), dim3(256), 0, (hipStream_t)stream, x, slope, y, nv, n);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* dslope, long n, float* ws,
                   size_t ws_bytes, void* stream) {
    if (!dy || !x || !slope || !dx || !dslope || !ws || n <= 0) return RCOT_EINVAL;
    const long nv = (al16(x) && al16(dy) && al16(dx)) ? n / 4 : 0;
    const int G = grid_for(nv ? nv : n, 256, 1024);
    if ((size_t)G * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
    RCOT_LAUNCH(prelu_bwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, dy, x, slope, dx, ws, nv, n);
    RCOT_LAUNCH_CHECK();
    RCOT_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, G, dslope);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_row_dot(const float* a, const float* b, float* out, long rows, int N, float scale, void* stream) {
    if (!a || !out || rows <= 0 || rows > 0x7fffffffL || N <= 0) return RCOT_EINVAL;
    const bool vec = (N & 3) == 0 && al16(a) && (!b || al16(b));
    if (vec)
        RCOT_LAUNCH(row_dot_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, b, out, N, scale);
    else
        RCOT_LAUNCH(row_dot_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, b, out, N, scale);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_row_scale_add(const float* a, const float* s, const float* x, const float* t, float tscale, float* out, long rows, int N,
                       void* stream) {
    if (!a || !s || !out || rows <= 0 || rows > 0x7fffffffL || N <= 0) return RCOT_EINVAL;
    const bool vec = (N & 3) == 0 && al16(a) && al16(out) && (!x || al16(x));
    const int per = vec ? 1024 : 256;
    int gy = (N + per - 1) / per;
    gy = gy > 64 ? 64 : gy;
    if (vec)
        RCOT_LAUNCH(row_scale_add_kernel<true>, dim3((unsigned)rows, gy), dim3(256), 0, (hipStream_t)stream, a, s, x, t, tscale, out, N);
    else
        RCOT_LAUNCH(row_scale_add_kernel<false>, dim3((unsigned)rows, gy), dim3(256), 0, (hipStream_t)stream, a, s, x, t, tscale, out, N);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_ca_gate_fwd(const float* mean, const float* W1, const float* W2, float* hid, float* gate, int B, int C, int Cr, void* stream) {
    if (!mean || !W1 || !W2 || !hid || !gate || B <= 0 || C <= 0 || Cr <= 0) return RCOT_EINVAL;
    if (C > CA_MAXC || Cr > CA_MAXR) return RCOT_EUNSUPPORTED;
    RCOT_LAUNCH(ca_gate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mean, W1, W2, hid, gate, C, Cr);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_ca_gate_bwd(const float* dgate, const float* gate, const float* hid, const float* mean, const float* W1, const float* W2,
                     float* dW1, float* dW2, float* dmean, int B, int C, int Cr, float* ws, size_t ws_bytes, void* stream) {
    if (!dgate || !gate || !hid || !mean || !W1 || !W2 || !dW1 || !dW2 || !dmean || !ws || B <= 0 || C <= 0 || Cr <= 0) return RCOT_EINVAL;
    if (C > CA_MAXC || Cr > CA_MAXR) return RCOT_EUNSUPPORTED;
    if ((size_t)B * (C + Cr) * sizeof(float) > ws_bytes) return RCOT_EWORKSPACE;
    float* ds = ws;
    float* dh = ws + (size_t)B * C;
    RCOT_LAUNCH(ca_gate_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dgate, gate, hid, W1, W2, dmean, ds, dh, C, Cr);
    RCOT_LAUNCH_CHECK();
    RCOT_LAUNCH(ca_gate_wgrad_kernel, dim3((C * Cr + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)ds, (const float*)dh,
                hid, mean, dW1, dW2, B, C, Cr);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_bilinear_down2(const float* x, float* y, long planes, int H, int W, void* stream) {
    if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (reinterpret_cast<uintptr_t>(x) & 7)) return RCOT_EINVAL;
    const long total = planes * (H / 2) * (W / 2);
    RCOT_LAUNCH(down2_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, x, y, total, H / 2, W / 2);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_bilinear_down2_bwd(const float* dy, float* dx, long planes, int H, int W, float beta, void* stream) {
    if (!dy || !dx || planes <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (reinterpret_cast<uintptr_t>(dx) & 7)) return RCOT_EINVAL;
    const long total = planes * (H / 2) * (W / 2);
    RCOT_LAUNCH(down2_bwd_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, dy, dx, total, H / 2, W / 2, beta);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_bilinear_up2(const float* x, const float* skip, float* y, long planes, int H, int W, void* stream) {
    if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || H > (1 << 14) || W > (1 << 14)) return RCOT_EINVAL;
    const long total = planes * (2L * H) * (2L * W);
    RCOT_LAUNCH(up2_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, x, skip, y, total, H, W);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_bilinear_up2_bwd(const float* dy, float* dx, long planes, int H, int W, void* stream) {
    if (!dy || !dx || planes <= 0 || H <= 0 || W <= 0 || H > (1 << 14) || W > (1 << 14)) return RCOT_EINVAL;
    const long total = planes * (long)H * W;
    RCOT_LAUNCH(up2_bwd_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, dy, dx, total, H, W);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

int rcot_conv_weight_flip(const float* src, float* dst, const long long* table, int n, int Co, int Ci, int KH, int KW, void* stream) {
    if (!src || !dst || !table || n <= 0 || n > 65535 || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0 || (long)Co * Ci * KH * KW > 0x7fffffffL)
        return RCOT_EINVAL;
    const int per = Co * Ci * KH * KW;
    RCOT_LAUNCH(weight_flip_kernel, dim3((per + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, src, dst, table, Co, Ci, KH, KW);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}

}  // extern "C"
