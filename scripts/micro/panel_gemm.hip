// Stand-alone check + timing of the column-panel projection kernel (rcot_amd/csrc/gemm_panel.h):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/micro/panel_gemm.hip -o /tmp/panel_gemm && /tmp/panel_gemm
// small case against fp64 on the host (with and without the LayerNorm prologue), then the level-1 shapes at 8 x 128 x 128 timed cold
// (every launch on the next of several operand sets).
#include "gemm_panel.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace rcot_panel;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int check(int M, int Z, int N, bool ln) {
    const int K = 96, lda = (M + 3) & ~3;
    std::vector<float> At((size_t)K * lda, 0.f), B((size_t)Z * K * N), C((size_t)Z * M * N, -777.f), mu((size_t)Z * N), rs((size_t)Z * N), lw(K), lb(K);
    for (int k = 0; k < K; ++k) for (int m = 0; m < M; ++m) At[(size_t)k * lda + m] = frand() * 0.1f;
    for (auto& v : B) v = frand() + 0.5f;
    for (auto& v : mu) v = frand() * 0.2f;
    for (auto& v : rs) v = 1.f + 0.3f * frand();
    for (int k = 0; k < K; ++k) { lw[k] = 1.f + 0.2f * frand(); lb[k] = 0.1f * frand(); }
    float *dA, *dB, *dC, *dmu, *drs, *dlw, *dlb;
    CK(hipMalloc(&dA, At.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4 + 4096));
    CK(hipMalloc(&dmu, mu.size() * 4)); CK(hipMalloc(&drs, rs.size() * 4)); CK(hipMalloc(&dlw, K * 4)); CK(hipMalloc(&dlb, K * 4));
    CK(hipMemcpy(dA, At.data(), At.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmu, mu.data(), mu.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drs, rs.data(), rs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dlw, lw.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlb, lb.data(), K * 4, hipMemcpyHostToDevice));
    std::vector<float> guard(C.size() + 1024, -777.f);
    CK(hipMemcpy(dC, guard.data(), guard.size() * 4, hipMemcpyHostToDevice));
    PanelArgs a{};
    a.M = M; a.N = N; a.K = K; a.Zo = Z; a.Zi = 1; a.At = dA; a.lda = lda; a.B = dB; a.ldb = N; a.sBo = (long)K * N; a.sBi = 0;
    a.C = dC; a.ldc = N; a.sCo = (long)M * N; a.sCi = 0;
    if (ln) { a.mu = dmu; a.rs = drs; a.sLN = N; a.lnw = dlw; a.lnb = dlb; }
    a.nts = 0;
    const int rc = try_gemm_panel(a, 256, 0);
    CK(hipDeviceSynchronize());
    if (rc) { printf("M=%d ln=%d: not taken (rc %d)\n", M, (int)ln, rc); return 1; }
    std::vector<float> out(guard.size());
    CK(hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int z = 0; z < Z; ++z)
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double acc = 0;
                for (int k = 0; k < K; ++k) {
                    double b = B[((size_t)z * K + k) * N + n];
                    if (ln) b = (double)(float)((float)((float)((float)b - mu[(size_t)z * N + n]) * rs[(size_t)z * N + n]) * lw[k]) + lb[k];
                    acc += (double)At[(size_t)k * lda + m] * b;
                }
                const double got = out[((size_t)z * M + m) * N + n];
                maxerr = fmax(maxerr, fabs(got - acc));
                maxref = fmax(maxref, fabs(acc));
            }
    int bad_guard = 0;
    for (size_t i = C.size(); i < out.size(); ++i) bad_guard += out[i] != -777.f;
    printf("M=%3d Z=%d N=%5d ln=%d: max|C - fp64| / max|C| = %.2e   guard words touched %d\n", M, Z, N, (int)ln, maxerr / maxref, bad_guard);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dmu); hipFree(drs); hipFree(dlw); hipFree(dlb);
    return (maxerr / maxref > 2e-6) || bad_guard;
}

void timeit(int M, bool ln) {
    const int K = 96, Z = 8, N = 16384, lda = (M + 3) & ~3, NSET = 4;
    float *dA, *dB[NSET], *dC[NSET], *dmu, *drs, *dlw, *dlb;
    // random operands (the clock of the part depends on what the MFMAs chew on: zeros run faster than data)
    std::vector<float> At((size_t)K * lda), hb((size_t)Z * K * N), hs((size_t)Z * N), hw(K);
    for (auto& v : At) v = frand() * 0.1f;
    CK(hipMalloc(&dA, At.size() * 4)); CK(hipMemcpy(dA, At.data(), At.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < NSET; ++i) {
        for (auto& v : hb) v = frand() + 0.5f;
        CK(hipMalloc(&dB[i], hb.size() * 4)); CK(hipMemcpy(dB[i], hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dC[i], (size_t)Z * M * N * 4));
    }
    CK(hipMalloc(&dmu, hs.size() * 4)); CK(hipMalloc(&drs, hs.size() * 4)); CK(hipMalloc(&dlw, K * 4)); CK(hipMalloc(&dlb, K * 4));
    for (auto& v : hs) v = 0.5f + 0.1f * frand();
    CK(hipMemcpy(dmu, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : hs) v = 1.5f + 0.3f * frand();
    CK(hipMemcpy(drs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : hw) v = 1.f + 0.2f * frand();
    CK(hipMemcpy(dlw, hw.data(), K * 4, hipMemcpyHostToDevice));
    for (auto& v : hw) v = 0.1f * frand();
    CK(hipMemcpy(dlb, hw.data(), K * 4, hipMemcpyHostToDevice));
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int nts = 0; nts < 2; ++nts) {
        auto go = [&](int i) {
            PanelArgs a{};
            a.M = M; a.N = N; a.K = K; a.Zo = Z; a.Zi = 1; a.At = dA; a.lda = lda; a.B = dB[i]; a.ldb = N; a.sBo = (long)K * N;
            a.C = dC[i]; a.ldc = N; a.sCo = (long)M * N;
            if (ln) { a.mu = dmu; a.rs = drs; a.sLN = N; a.lnw = dlw; a.lnb = dlb; }
            a.nts = nts;
            if (try_gemm_panel(a, 256, 0)) { printf("not taken\n"); exit(1); }
        };
        for (int i = 0; i < NSET; ++i) go(i);
        CK(hipDeviceSynchronize());
        const int reps = 12;
        CK(hipEventRecord(s));
        for (int r = 0; r < reps; ++r) go(r % NSET);
        CK(hipEventRecord(e)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, s, e));
        const double us = ms * 1e3 / reps, gf = 2.0 * M * K * (double)Z * N / 1e9, mb = 4.0 * Z * N * (double)(M + K) / 1e6;
        printf("%3d <- 96 at 8 x 128x128 ln=%d nts=%d: %7.1f us  %6.1f TF/s (%.2f of 157.3)  %5.0f GB/s\n", M, (int)ln, nts, us, gf / us / 1e3, gf / us / 1e3 / 157.3, mb / us * 1e-3 * 1e3);
    }
    hipFree(dA); for (int i = 0; i < NSET; ++i) { hipFree(dB[i]); hipFree(dC[i]); } hipFree(dmu); hipFree(drs); hipFree(dlw); hipFree(dlb);
}

int main() {
    int bad = 0;
    for (int M : {288, 510, 255, 384})
        for (int ln = 0; ln < 2; ++ln) bad += check(M, 3, 640, ln);
    if (!getenv("PANEL_SKIP_CHECK")) bad += check(288, 2, 16384, true);
    printf(bad ? "CHECK FAILED\n" : "checks ok\n");
    for (int M : {288, 510, 255})
        for (int ln = 0; ln < 2; ++ln) timeit(M, ln);
    return bad;
}
