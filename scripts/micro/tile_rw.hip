// How fast can HBM be written / read in TILES of a row-major [rows][16384] fp32 plane set (row stride 64 KiB), the way a GEMM
// epilogue or an operand loader touches it?  Each workgroup (256 threads) handles tiles of R rows x W bytes; tiles walk the
// column-tile index fastest or the row-tile index fastest.  Compare with a linear sweep of the same bytes.
// Build: hipcc --offload-arch=gfx950 -O3 tile_rw.hip -o tile_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode 0 write, 1 read.  grid-stride over tiles.  rowfast: consecutive tile ids are row-tiles of the same column block.
template <int MODE>
__global__ __launch_bounds__(256) void k(float* p, long rows, long ld, int R, int Wf, int rowfast, float* sink) {
    const long tcols = ld / Wf, trows = rows / R, nt = tcols * trows;
    const int lanes_per_row = Wf / 4;                  // float4 per tile row
    f32x4 acc = {0, 0, 0, 0};
    for (long t = blockIdx.x; t < nt; t += gridDim.x) {
        const long tr = rowfast ? t % trows : t / tcols, tc = rowfast ? t / trows : t % tcols;
        float* base = p + tr * R * ld + tc * Wf;
        for (int i = threadIdx.x; i < R * lanes_per_row; i += 256) {
            const int r = i / lanes_per_row, c = i % lanes_per_row;
            f32x4* q = reinterpret_cast<f32x4*>(base + (long)r * ld) + c;
            if (MODE == 0) *q = f32x4{1.f, 2.f, 3.f, (float)t};
            else acc += *q;
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}
int main(int argc, char** argv) {
    const int grid_arg = argc > 1 ? atoi(argv[1]) : 2048;
    const long ld = 16384, rows = 8 * 2048;            // 8 images x 2048 channel rows x 64 KiB = 1 GiB
    float *p, *sink;
    hipMalloc(&p, rows * ld * 4); hipMalloc(&sink, 4);
    hipMemset(p, 0, rows * ld * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const double gb = rows * ld * 4 / 1e9;
    for (int mode = 0; mode < 2; ++mode)
        for (int rowfast = 0; rowfast < 2; ++rowfast) {
            const int Rs[] = {128, 1};
            const int narrow = argc > 2 ? atoi(argv[2]) : 0;
            const int Ws[] = {narrow ? 16 : 256, narrow ? 32 : 4096, narrow ? 64 : 4096, narrow ? 128 : 4096};
            for (int R : Rs) for (int Wf : Ws) {
                if (R == 1 && Wf != 4096) continue;
                if (R != 1 && (long)R * Wf > 128 * 256) continue;      // same 128 KiB tiles or smaller
                if (!narrow && (long)R * Wf < 64 * 128) continue;
                const int grid = grid_arg;
                auto run = [&]() { if (mode == 0) k<0><<<grid, 256>>>(p, rows, ld, R, Wf, rowfast, sink); else k<1><<<grid, 256>>>(p, rows, ld, R, Wf, rowfast, sink); };
                run();
                hipEventRecord(s); run(); run(); hipEventRecord(e); hipEventSynchronize(e);
                float ms; hipEventElapsedTime(&ms, s, e);
                printf("grid %4d %s rowfast=%d tile %3d rows x %5d B: %6.0f GB/s (%5.1f GB/s per workgroup)\n", grid, mode ? "read " : "write", rowfast, R, Wf * 4, 2 * gb / ms * 1e3, 2 * gb / ms * 1e3 / grid);
            }
        }
    return 0;
}
