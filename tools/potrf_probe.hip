// potrf_probe.hip -- developer probe: where the time of hb3_chain_kernel goes (time stamps of workgroup 0's phases,
// s_memrealtime at 100 MHz) on a random SPD band matrix of the wide-image shape (bw = 1024).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DBCR_PROBE_ON -I bpldenoising_amd/csrc tools/potrf_probe.hip -o tools/_bin/potrf_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hb_band_solver.hpp"
using namespace bpltv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

int main(int argc, char** argv) {
    const int bw = argc > 1 ? atoi(argv[1]) : 1024, n = argc > 2 ? atoi(argv[2]) : 8192, O = argc > 3 ? atoi(argv[3]) : 1;
    const int offs[4] = {0, 1, bw - 1, bw};
    std::vector<double> pl((size_t)4 * O * n);
    for (int t = 0; t < 4; ++t)
        for (size_t i = 0; i < (size_t)O * n; ++i) pl[(size_t)t * O * n + i] = t == 0 ? 8.0 : -1.0 + 0.001 * (double)(i % 97);
    double* d_pl; int* d_fail;
    CK(hipMalloc(&d_pl, pl.size() * 8)); CK(hipMalloc(&d_fail, O * sizeof(int)));
    CK(hipMemcpy(d_pl, pl.data(), pl.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_fail, 0, O * sizeof(int)));
    hipStream_t st; CK(hipStreamCreate(&st));
    BandDiags D; D.planes = d_pl; D.tot = (size_t)O * n; D.nd = 4;
    for (int t = 0; t < 4; ++t) D.off[t] = offs[t];
    HbBandSolver ch;
    if (ch.alloc(bw, n, O, st, false)) { printf("alloc: %s\n", ch.err.c_str()); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // stamp the chain kernel of a panel in the middle of the run (all streams busy), not the last one
        const int mid = argc > 4 ? atoi(argv[4]) : ((n + 127) / 128) / 2;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bcr_probe_panel), &mid, sizeof(int)));
        printf("chain-kernel stamps: panel %d\n", mid);
    }
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        if (ch.factor(D, d_fail)) { printf("factor: %s\n", ch.err.c_str()); return 1; }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("factor bw %d n %d O %d: %.3f ms = %.1f us per panel\n", bw, n, O, ms, 1e3 * ms / ((n + 127) / 128));
    }
    long long h[64];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(bcr_probe_buf), sizeof(h)));
    auto us = [&](int a, int b) { return (h[b] - h[a]) * 0.01; };
    printf("last chain kernel, workgroup 0 (us): loads+stage %.2f | P0 product %.2f | D update %.2f | D to LDS %.2f | potrf body %.2f | stores %.2f\n",
           us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6));
    printf("potrf body per block column (us): update / panel+inverse work\n");
    long long prev = h[4];
    for (int p = 0; p < 8; ++p) {
        printf("  p %d: %.2f / %.2f\n", p, (h[8 + 2 * p] - prev) * 0.01, (h[9 + 2 * p] - h[8 + 2 * p]) * 0.01);
        prev = h[9 + 2 * p];
    }
    printf("  tail: %.2f\n", (h[5] - prev) * 0.01);
    printf("last bulk update launch, workgroup 0 (us): operands requested -> chunk 0 staged %.2f", us(32, 33));
    for (int ch = 0; ch < 4; ++ch) printf(" | mfma %d %.2f", ch, us(33 + 2 * ch, 34 + 2 * ch));
    for (int ch = 1; ch < 4; ++ch) printf(" | stage %d %.2f", ch, us(32 + 2 * ch, 33 + 2 * ch));
    printf(" | old tile requested %.2f | acc to LDS (waits for the tile) %.2f | stores %.2f | total %.2f\n", us(40, 41), us(41, 42),
           us(42, 43), us(32, 43));
    {   // launch spans of the last panels (all workgroups): C = chain, T = trsm, U1 / U2 = update parts
        static unsigned long long sp[4][64][2];
        CK(hipMemcpyFromSymbol(sp, HIP_SYMBOL(bcr_span), sizeof(sp)));
        const int npan = (n + 127) / 128;
        unsigned long long t0 = ~0ull;
        for (int k = npan - 30; k < npan - 20; ++k) if (k >= 0 && sp[0][k & 63][0] < t0) t0 = sp[0][k & 63][0];
        printf("launch spans, panels %d..%d (us from the first chain kernel): begin-end of C | T | U1 | U2\n", npan - 30, npan - 21);
        for (int k = npan - 30; k < npan - 20; ++k) {
            if (k < 0) continue;
            printf("  panel %4d:", k);
            for (int kind = 0; kind < 4; ++kind) {
                const unsigned long long b = sp[kind][k & 63][0], e = sp[kind][k & 63][1];
                if (b == ~0ull || e == 0) printf("        -        |");
                else printf(" %7.1f-%7.1f |", (double)(long long)(b - t0) * 0.01, (double)(long long)(e - t0) * 0.01);
            }
            printf("\n");
        }
    }
    {   // one solve: phases of the last forward-substitution launch (workgroup 1 of problem 0)
        double *d_v, *d_s;
        CK(hipMalloc(&d_v, (size_t)O * n * 8)); CK(hipMalloc(&d_s, (size_t)O * n * 8));
        CK(hipMemset(d_v, 0, (size_t)O * n * 8));
        CK(hipEventRecord(e0, st));
        ch.solve(d_v, nullptr, d_s);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(bcr_probe_buf), sizeof(h)));
        printf("solve %.3f ms = %.2f us per block and direction; last forward launch, a tile workgroup (us): loads issued + v staged %.2f"
               " | Linv product + reduce %.2f | tile product + reduce %.2f | update stored %.2f | total %.2f\n",
               ms, 1e3 * ms / (2.0 * ((n + 127) / 128)), us(50, 51), us(51, 52), us(52, 53), us(53, 54), us(50, 54));
    }
    int f = 0; CK(hipMemcpy(&f, d_fail, sizeof(int), hipMemcpyDeviceToHost));
    printf("fail flag %d\n", f);
    return 0;
}
