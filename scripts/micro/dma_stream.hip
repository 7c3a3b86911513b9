// Per-CU and chip-wide rate of operand streaming into a CU, as a GEMM loader does it: workgroups of 256 threads pull "slabs"
// of P 1-KiB pieces (64 lanes x 16 B, pieces of one slab 64 KiB apart like the rows of a [K][16384] fp32 plane) through an
// NST-stage ring and do nothing else.  MODE 0: global_load_lds_dwordx4 (LDS-DMA); MODE 1: global_load_dwordx4 to registers.
// Source either streams from HBM (every workgroup its own columns) or is a small L2-resident panel every workgroup re-reads.
// Build: hipcc --offload-arch=gfx950 -O3 dma_stream.hip -o dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int PWV, int NST>      // PWV pieces per wave per slab (4 waves), NST ring stages
__global__ __launch_bounds__(256) void k(const float* src, long wg_stride_f, long row_stride_f, int nslab, int wrap, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* base = src + (long)blockIdx.x * wg_stride_f + lane * 4;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 r[NST][PWV];
    auto issue = [&](int s) {
#pragma unroll
        for (int h = 0; h < PWV; ++h) {
            const int q = wave + 4 * h;                                   // piece = one 1-KiB row segment
            const int pi = s * 4 * PWV + q;
            const float* p = base + (long)(wrap ? pi % wrap : pi) * row_stride_f;
            if (MODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + ((s % NST) * 4 * PWV + q) * 256), 16, 0, 0);
            else r[s % NST][h] = *reinterpret_cast<const f32x4*>(p);
        }
    };
    // fully unrolled ring over NST-deep prefetch: process slabs in groups of NST so that r[][] indices are static
    for (int i = 0; i < NST - 1 && i < nslab; ++i) issue(i);
    for (int s0 = 0; s0 < nslab; s0 += NST) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int s = s0 + j;
            if (s >= nslab) break;
            if (MODE == 0) {
                if (s + NST - 1 < nslab) wait_vm<(NST - 2) * PWV>(); else wait_vm<0>();
                __syncthreads();
            }
            if (s + NST - 1 < nslab) issue(s + NST - 1);
            if (MODE == 0) acc += *reinterpret_cast<const f32x4*>(lds + (s % NST) * 4 * PWV * 256 + threadIdx.x * 4);
            else {
#pragma unroll
                for (int h = 0; h < PWV; ++h) acc += r[j][h];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}

template <int MODE, int PWV, int NST>
void run(const char* what, const float* src, long wg_stride_f, long row_stride_f, int nslab, int wrap, int grid, float* sink) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    const size_t smem = MODE == 0 ? (size_t)NST * 4 * PWV * 1024 : 0;
    (void)hipFuncSetAttribute((const void*)k<MODE, PWV, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<MODE, PWV, NST><<<grid, 256, smem>>>(src, wg_stride_f, row_stride_f, nslab, wrap, sink);
    (void)hipEventRecord(s);
    k<MODE, PWV, NST><<<grid, 256, smem>>>(src, wg_stride_f, row_stride_f, nslab, wrap, sink);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double bytes = (double)grid * nslab * 4 * PWV * 1024;
    printf("%-28s %s PWV=%d (slab %2d KiB) NST=%d grid=%4d: %7.1f us  %6.0f GB/s  (%5.1f GB/s per CU)\n", what, MODE ? "regs" : "glds", PWV,
           4 * PWV, NST, grid, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 256);
}

int main() {
    const long ld = 16384, rows = 16384;         // [rows][16384] fp32 = 1 GiB, row stride 64 KiB
    float *p, *sink;
    (void)hipMalloc(&p, rows * ld * 4); (void)hipMalloc(&sink, 4);
    (void)hipMemset(p, 0, rows * ld * 4);
    // workgroup b = (column block b % 64, row block b / 64); a slab = 4*PWV consecutive rows of 1 KiB
    // HBM stream: 64 slabs per workgroup; encoded through wg_stride by launching with the buffer viewed as column blocks:
    // block b reads rows [(b/64)*RB, +RB) -> base offset (b/64)*RB*ld + (b%64)*256: pass as a table-free formula below.
#define RUN(MODE, PWV, NST, G) run<MODE, PWV, NST>(what, p, wgs, rs, nsl, wrap, G, sink)
    for (int pass = 1; pass < 4; ++pass) {
        const char* what = pass == 0 ? "HBM stream (own 1-KiB columns)" : pass == 1 ? "L2 panel (128 KiB, shared)" : pass == 2 ? "L2 panel, rows 64 KiB apart" : "L2 panel, rows 4 KiB apart";
        // pass 0: wg stride 256 floats walks the 64 column blocks, then continues into the next rows because 64*256 = ld:
        //         b*256 = (b/64)*ld + (b%64)*256 -> workgroups b/64 > 0 start ONE row lower only; give each its own row block via
        //         a large row stride instead: slab rows are rs apart with rs = ld * 16 (every 16th row), so 1024 workgroups x 64 slabs
        //         x 16 pieces touch distinct 1-KiB segments:  row = 16*piece + b/64.
        const long wgs = pass == 0 ? 256 : 0;
        const long rs = pass == 0 ? ld * 16 : pass == 1 ? 256 : pass == 2 ? ld : 1024;          // pass 1: pieces are consecutive 1-KiB of one small panel
        const int nsl = 40, wrap = pass == 0 ? 0 : 128;
        for (int G : {256, 512}) {
            RUN(0, 6, 3, G); RUN(0, 6, 4, G); RUN(0, 4, 3, G); RUN(0, 4, 6, G); RUN(0, 2, 8, G);
            RUN(1, 6, 3, G); RUN(1, 4, 4, G);
        }
    }
    return 0;
}
