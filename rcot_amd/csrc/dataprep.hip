// Patch preparation on the device: the per-sample work of the reference's TrainDataset.__getitem__
// (util/dataset_utils.py:215-281) after the image file has been decoded —
//   crop a P x P window            (RandomCrop / _crop_patch, :57-60, :188-196)
//   one of the 8 dihedral maps      (util/image_utils.py:133-163 data_augmentation, mode drawn by :177-182)
//   Gaussian noise + uint8 quantise (util/degradation_utils.py:21-27: clip(clean + randn*sigma, 0, 255).astype(uint8))
//   ToTensor                        (HWC uint8 -> CHW float / 255, :264-265)
// in ONE pass from the uint8 HWC image(s) resident in HBM straight into the sample's slot of the batch tensors.
// The host only decodes files and draws (y0, x0, mode, seed); at >= 1000 patches/s the PIL/numpy chain of the reference
// (num_workers = 0, trainer.py:32,134) would be the bottleneck of the training loop.
#include "common.h"
#include "../../include/rcot_hip.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {          // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// standard normal from a counter (Box-Muller on two 24-bit uniforms)
__device__ __forceinline__ float counter_randn(uint64_t seed, uint64_t idx) {
    const uint64_t h = mix64(seed + 0x9e3779b97f4a7c15ull * (idx + 1));
    const float u1 = ((float)((h >> 40) & 0xffffff) + 1.0f) * (1.0f / 16777217.0f);     // (0, 1)
    const float u2 = (float)((h >> 8) & 0xffffff) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// source pixel (sy, sx) inside the crop for output pixel (i, j) under numpy's rot90 / flipud composition
__device__ __forceinline__ void dihedral(int mode, int P, int i, int j, int& sy, int& sx) {
    switch (mode) {
        case 1: sy = P - 1 - i; sx = j; break;               // flipud
        case 2: sy = j; sx = P - 1 - i; break;               // rot90 (counter-clockwise)
        case 3: sy = j; sx = i; break;                       // rot90 + flipud
        case 4: sy = P - 1 - i; sx = P - 1 - j; break;       // rot180
        case 5: sy = i; sx = P - 1 - j; break;               // rot180 + flipud
        case 6: sy = P - 1 - j; sx = i; break;               // rot270
        case 7: sy = P - 1 - j; sx = P - 1 - i; break;       // rot270 + flipud
        default: sy = i; sx = j; break;                      // 0: identity
    }
}

__global__ __launch_bounds__(256) void patch_prep_kernel(const unsigned char* __restrict__ deg, const unsigned char* __restrict__ clean,
                                                         int W, int y0, int x0, int P, int mode, float sigma, uint64_t seed,
                                                         float* __restrict__ deg_out, float* __restrict__ clean_out) {
    const int n = P * P;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const int i = p / P, j = p - i * P;
        int sy, sx;
        dihedral(mode, P, i, j, sy, sx);
        const long src = ((long)(y0 + sy) * W + (x0 + sx)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float cv = (float)clean[src + c];
            float dv;
            if (deg) {
                dv = (float)deg[src + c];
            } else {
                // the reference adds noise AFTER the augmentation, element by element of the HWC patch
                const float v = cv + sigma * counter_randn(seed, ((uint64_t)p) * 3 + c);
                dv = floorf(fminf(fmaxf(v, 0.f), 255.f));                        // clip, then astype(uint8) truncation
            }
            clean_out[(long)c * n + p] = cv / 255.0f;            // ToTensor divides (bit-equal to the reference's values)
            deg_out[(long)c * n + p] = dv / 255.0f;
        }
    }
}

}  // namespace

extern "C" int rcot_patch_prep(const unsigned char* deg_img, const unsigned char* clean_img, int H, int W, int y0, int x0,
                               int P, int mode, float noise_sigma, unsigned long long seed, float* deg_out, float* clean_out,
                               void* stream) {
    if (!clean_img || !deg_out || !clean_out || P <= 0 || y0 < 0 || x0 < 0 || y0 + P > H || x0 + P > W || mode < 0 || mode > 7)
        return RCOT_EINVAL;
    const int nb = (P * P + 255) / 256;
    RCOT_LAUNCH(patch_prep_kernel, dim3(nb > 256 ? 256 : nb), dim3(256), 0, (hipStream_t)stream, deg_img, clean_img, W, y0, x0,
                       P, mode, noise_sigma, (uint64_t)seed, deg_out, clean_out);
    RCOT_LAUNCH_CHECK();
    return RCOT_OK;
}
