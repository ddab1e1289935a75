// nd_unit.hip -- unit checks of the nested-dissection Cholesky (csrc/nd_solver.hpp, nd_kernels.hpp) against the host
// restatement tools/nd_ref.hpp: factor entries (W = L11^-1, L21 of every front) and solutions on random SPD stencil
// matrices, several images per call, both stencils, shapes with small-regime levels only and with large-regime
// levels (multi-panel pivot blocks); the LU variant on random non-symmetric stencil matrices (solutions).
// `nd_unit time M nimg [sr]` times factor and solve on an M x M grid.
// tests/test_gpu_evaluate.py::test_nd_solver_unit_checks runs it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../bpldenoising_amd/csrc/nd_solver.hpp"
#include "nd_ref.hpp"

#define CK(call)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            printf("%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

using namespace bpltv;

static int check(int M, int N, bool sr, int leaf, int nimg, double big) {
    NdSolver S;
    const NdStencil st = sr ? nd_stencil_sr() : nd_stencil_tv();
    if (S.build(M, N, st, leaf)) { printf("build failed: %s\n", S.err.c_str()); return 1; }
    S.skinny2_min = 0;   // every kernel the tree is eligible for, whatever the number of images (the library waits for 256 fronts per level)
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    if (S.alloc(nimg, stream)) { printf("alloc failed: %s\n", S.err.c_str()); return 1; }
    const NdTree& T = S.T;
    const int n = T.n, nd = st.nd;
    const size_t tot = (size_t)nimg * n;
    std::vector<double> planes((size_t)nd * tot), rhs(tot), xt(tot);
    std::vector<std::vector<std::vector<double>>> P(nimg);
    for (int k = 0; k < nimg; ++k) {
        P[k] = ndref::random_spd(T, 100 + 7 * k + M, big);
        for (int t = 0; t < nd; ++t) memcpy(&planes[(size_t)t * tot + (size_t)k * n], P[k][t].data(), n * sizeof(double));
        std::vector<double> x(n), b;
        for (int g = 0; g < n; ++g) x[g] = std::sin(0.37 * g + k) + 0.1 * (g % 7);
        ndref::matvec(T, P[k], x, b);
        memcpy(&xt[(size_t)k * n], x.data(), n * sizeof(double));
        memcpy(&rhs[(size_t)k * n], b.data(), n * sizeof(double));
    }
    double *d_planes, *d_vec, *d_acc;
    int* d_fail;
    CK(hipMalloc((void**)&d_planes, planes.size() * sizeof(double)));
    CK(hipMalloc((void**)&d_vec, tot * sizeof(double)));
    CK(hipMalloc((void**)&d_acc, tot * sizeof(double)));
    CK(hipMalloc((void**)&d_fail, nimg * sizeof(int)));
    CK(hipMemcpy(d_planes, planes.data(), planes.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vec, rhs.data(), tot * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemset(d_acc, 0, tot * sizeof(double)));
    CK(hipMemset(d_fail, 0, nimg * sizeof(int)));
    if (S.factor(d_planes, tot, nimg, d_fail)) { printf("factor failed: %s\n", S.err.c_str()); return 1; }
    if (S.solve(d_vec, d_acc, nimg)) { printf("solve failed: %s\n", S.err.c_str()); return 1; }
    CK(hipStreamSynchronize(stream));
    std::vector<int> fail(nimg);
    std::vector<double> x(tot), acc(tot), fac((size_t)nimg * T.fac_doubles);
    CK(hipMemcpy(fail.data(), d_fail, nimg * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(x.data(), d_vec, tot * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(acc.data(), d_acc, tot * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(fac.data(), S.fac, fac.size() * sizeof(double), hipMemcpyDeviceToHost));
    int bad = 0;
    double worst_f = 0.0, worst_x = 0.0;
    int nsmall = 0, nlarge = 0;
    for (const auto& a : S.lv) (a.small ? nsmall : nlarge)++;
    for (int k = 0; k < nimg; ++k) {
        if (fail[k]) { printf("  image %d: pivot failure at node %d\n", k, fail[k] - 1); ++bad; continue; }
        const ndref::Factor F = ndref::factor(T, P[k]);
        // factor entries, front by front, relative to the front's largest entry
        for (size_t q = 0; q < T.nodes.size(); ++q) {
            const NdNode& v = T.nodes[q];
            const int f = v.p + v.b;
            double mx = 0.0, df = 0.0;
            for (int c = 0; c < v.p; ++c)
                for (int r = 0; r < f; ++r) {
                    if (r < v.p && r < c) continue;
                    // pivot blocks wider than one panel: the library keeps W_kk per 128-column panel and L between
                    // the panels, the restatement the whole inverse -- compare L21 there (it is unique)
                    if (r < v.p && v.p > HB2_NB) continue;
                    const double a = F.fac[(size_t)v.fac_off + r + (size_t)f * c], g = fac[(size_t)k * T.fac_doubles + v.fac_off + r + (size_t)f * c];
                    mx = std::fmax(mx, std::fabs(a));
                    df = std::fmax(df, std::fabs(a - g));
                    if (g != g) df = 1e300;
                }
            if (mx > 0 && df / mx > worst_f) worst_f = df / mx;
        }
        double err = 0.0, nrm = 0.0;
        for (int g = 0; g < n; ++g) {
            err = std::fmax(err, std::fabs(x[(size_t)k * n + g] - xt[(size_t)k * n + g]));
            nrm = std::fmax(nrm, std::fabs(xt[(size_t)k * n + g]));
            if (acc[(size_t)k * n + g] != x[(size_t)k * n + g]) err = 1e300;   // acc started at zero
        }
        worst_x = std::fmax(worst_x, err / nrm);
    }
    const bool ok = bad == 0 && worst_f <= 1e-8 && worst_x <= 1e-8;
    printf("%s %4dx%-4d %s leaf %3d x%d images, %zu fronts, %d levels (%d small, %d large), max front %d: factor %.1e  solution %.1e\n",
           ok ? "ok  " : "FAIL", M, N, sr ? "sr" : "tv", leaf, nimg, T.nodes.size(), T.levels(), nsmall, nlarge, T.max_f, worst_f, worst_x);
    S.release();
    CK(hipFree(d_planes)); CK(hipFree(d_vec)); CK(hipFree(d_acc)); CK(hipFree(d_fail));
    CK(hipStreamDestroy(stream));
    return ok ? 0 : 1;
}

// LU variant: random row-diagonally-dominant NON-symmetric stencil matrices (independent lower and upper diagonals, a few
// strongly coupled pairs), solution against the vector the right-hand side was made from.
static int check_lu(int M, int N, bool sr, int leaf, int nimg, double big) {
    NdSolver S;
    const NdStencil st = sr ? nd_stencil_sr() : nd_stencil_tv();
    if (S.build(M, N, st, leaf, true)) { printf("build failed: %s\n", S.err.c_str()); return 1; }
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    if (S.alloc(nimg, stream)) { printf("alloc failed: %s\n", S.err.c_str()); return 1; }
    const NdTree& T = S.T;
    const int n = T.n, nd = st.nd;
    const size_t tot = (size_t)nimg * n;
    std::vector<double> pl((size_t)nd * tot, 0.0), pu((size_t)nd * tot, 0.0), rhs(tot, 0.0), xt(tot);
    unsigned long long rs = 1234567 + 31 * M + N;
    auto rnd = [&]() {
        rs = rs * 6364136223846793005ull + 1442695040888963407ull;
        return (double)((rs >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53);
    };
    for (int k = 0; k < nimg; ++k) {
        double* L = &pl[(size_t)k * n];
        double* U = &pu[(size_t)k * n];
        std::vector<double> rowsum(n, 0.0);
        for (int t = 1; t < nd; ++t)
            for (int g = 0; g < n; ++g) {
                const int i = g % M, j = g / M, a = i + st.di[t], c = j + st.dj[t];
                if (a < 0 || a >= M || c < 0 || c >= N) continue;
                const int g2 = a + M * c;   // g2 > g
                double vl = -(0.1 + rnd()), vu = -(0.1 + rnd());
                if (big > 0.0 && rnd() < 0.2) { vl *= big; vu *= big * (0.5 + rnd()); }
                L[(size_t)t * tot + g] = vl;   // A[g2][g]
                U[(size_t)t * tot + g] = vu;   // A[g][g2]
                rowsum[g2] += std::fabs(vl);
                rowsum[g] += std::fabs(vu);
            }
        for (int g = 0; g < n; ++g) L[g] = rowsum[g] + 1.0 + rnd();
        for (int g = 0; g < n; ++g) xt[(size_t)k * n + g] = std::sin(0.37 * g + k) + 0.1 * (g % 7);
        const double* x = &xt[(size_t)k * n];
        double* y = &rhs[(size_t)k * n];
        for (int g = 0; g < n; ++g) y[g] = L[g] * x[g];
        for (int t = 1; t < nd; ++t)
            for (int g = 0; g < n; ++g) {
                const int i = g % M, j = g / M, a = i + st.di[t], c = j + st.dj[t];
                if (a < 0 || a >= M || c < 0 || c >= N) continue;
                const int g2 = a + M * c;
                y[g2] += L[(size_t)t * tot + g] * x[g];
                y[g] += U[(size_t)t * tot + g] * x[g2];
            }
    }
    double *d_pl, *d_pu, *d_vec, *d_acc;
    int* d_fail;
    CK(hipMalloc((void**)&d_pl, pl.size() * sizeof(double)));
    CK(hipMalloc((void**)&d_pu, pu.size() * sizeof(double)));
    CK(hipMalloc((void**)&d_vec, tot * sizeof(double)));
    CK(hipMalloc((void**)&d_acc, tot * sizeof(double)));
    CK(hipMalloc((void**)&d_fail, nimg * sizeof(int)));
    CK(hipMemcpy(d_pl, pl.data(), pl.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pu, pu.data(), pu.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vec, rhs.data(), tot * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemset(d_acc, 0, tot * sizeof(double)));
    CK(hipMemset(d_fail, 0, nimg * sizeof(int)));
    if (S.factor_lu(d_pl, d_pu, tot, nimg, d_fail)) { printf("factor failed: %s\n", S.err.c_str()); return 1; }
    if (S.solve(d_vec, d_acc, nimg)) { printf("solve failed: %s\n", S.err.c_str()); return 1; }
    CK(hipStreamSynchronize(stream));
    std::vector<int> fail(nimg);
    std::vector<double> x(tot), acc(tot);
    CK(hipMemcpy(fail.data(), d_fail, nimg * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(x.data(), d_vec, tot * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(acc.data(), d_acc, tot * sizeof(double), hipMemcpyDeviceToHost));
    int bad = 0, nsmall = 0, nlarge = 0;
    for (const auto& a : S.lv) (a.small ? nsmall : nlarge)++;
    double worst = 0.0;
    for (int k = 0; k < nimg; ++k) {
        if (fail[k]) { printf("  image %d: pivot failure at node %d\n", k, fail[k] - 1); ++bad; continue; }
        double err = 0.0, nrm = 0.0;
        for (int g = 0; g < n; ++g) {
            const double xv = x[(size_t)k * n + g];
            err = std::fmax(err, std::fabs(xv - xt[(size_t)k * n + g]));
            nrm = std::fmax(nrm, std::fabs(xt[(size_t)k * n + g]));
            if (xv != xv || acc[(size_t)k * n + g] != xv) err = 1e300;
        }
        worst = std::fmax(worst, err / nrm);
    }
    const bool ok = bad == 0 && worst <= 1e-8;
    printf("%s LU %4dx%-4d %s leaf %3d x%d images, %zu fronts, %d levels (%d small, %d large), max front %d: solution %.1e\n", ok ? "ok  " : "FAIL", M, N,
           sr ? "sr" : "tv", leaf, nimg, T.nodes.size(), T.levels(), nsmall, nlarge, T.max_f, worst);
    S.release();
    CK(hipFree(d_pl)); CK(hipFree(d_pu)); CK(hipFree(d_vec)); CK(hipFree(d_acc)); CK(hipFree(d_fail));
    CK(hipStreamDestroy(stream));
    return ok ? 0 : 1;
}

static int timing(int M, int nimg, bool sr, int leaf) {
    NdSolver S;
    const NdStencil st = sr ? nd_stencil_sr() : nd_stencil_tv();
    auto t0 = std::chrono::steady_clock::now();
    if (S.build(M, M, st, leaf)) { printf("build failed: %s\n", S.err.c_str()); return 1; }
    if (getenv("ND_NO_STAGE")) S.staged_solve = false;  // A/B: the column-loop substitutions on every small level
    if (getenv("ND_NO_SKINNY")) S.skinny_fronts = false;
    if (getenv("ND_NO_WAVE")) S.wave_fronts = false;    // A/B: the workgroup-per-front kernel on every small level (this tool only)
    {
        int nw = 0, ns = 0;
        for (const auto& a : S.lv) { ns += a.small; nw += a.small && S.wave_fronts && a.wave >= 0; }
        printf("small levels %d, of them a wave per front %d\n", ns, nw);
    }
    const double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    if (S.alloc(nimg, stream)) { printf("alloc failed: %s\n", S.err.c_str()); return 1; }
    const NdTree& T = S.T;
    const int n = T.n, nd = st.nd;
    const size_t tot = (size_t)nimg * n;
    printf("%dx%d %s, %d images: build %.2f s, %zu fronts, %d levels, factor %.2f GB, workspace %.2f GB per image set, %.3g flop per image\n", M, M,
           sr ? "sr" : "tv", nimg, tb, T.nodes.size(), T.levels(), nimg * T.fac_doubles * 8e-9, nimg * S.bytes_per_image() * 1e-9, S.factor_flop);
    const auto P = ndref::random_spd(T, 5, 1e6);
    std::vector<double> planes((size_t)nd * tot), rhs(tot), xt(n), b;
    for (int g = 0; g < n; ++g) xt[g] = std::sin(0.37 * g) + 0.1 * (g % 7);
    ndref::matvec(T, P, xt, b);
    for (int k = 0; k < nimg; ++k) {
        for (int t = 0; t < nd; ++t) memcpy(&planes[(size_t)t * tot + (size_t)k * n], P[t].data(), n * sizeof(double));
        memcpy(&rhs[(size_t)k * n], b.data(), n * sizeof(double));
    }
    double *d_planes, *d_vec;
    int* d_fail;
    CK(hipMalloc((void**)&d_planes, planes.size() * sizeof(double)));
    CK(hipMalloc((void**)&d_vec, tot * sizeof(double)));
    CK(hipMalloc((void**)&d_fail, nimg * sizeof(int)));
    CK(hipMemcpy(d_planes, planes.data(), planes.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemset(d_fail, 0, nimg * sizeof(int)));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpyAsync(d_vec, rhs.data(), tot * sizeof(double), hipMemcpyHostToDevice, stream));
        CK(hipEventRecord(e0, stream));
        if (S.factor(d_planes, tot, nimg, d_fail)) { printf("factor failed: %s\n", S.err.c_str()); return 1; }
        CK(hipEventRecord(e1, stream));
        if (S.solve(d_vec, nullptr, nimg)) { printf("solve failed: %s\n", S.err.c_str()); return 1; }
        CK(hipEventRecord(e2, stream));
        CK(hipStreamSynchronize(stream));
        float mf = 0, ms = 0;
        CK(hipEventElapsedTime(&mf, e0, e1));
        CK(hipEventElapsedTime(&ms, e1, e2));
        printf("  rep %d: factor %.2f ms (%.1f TFLOP/s), solve %.2f ms (%.0f GB/s of factor)\n", rep, mf, nimg * S.factor_flop / mf * 1e-9, ms,
               2.0 * nimg * T.fac_doubles * 8 / ms * 1e-6);
    }
#ifdef ND_PROBE_ON
    for (int l = T.levels() - 1; l >= 0 && S.lv[l].small; --l) {   // phases of one mid-grid workgroup per small level
        const int n0 = S.lv[l].n0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(nd_probe_node0), &n0, sizeof(int)));
        if (S.factor(d_planes, tot, nimg, d_fail)) return 1;
        CK(hipStreamSynchronize(stream));
        long long pb[16];
        CK(hipMemcpyFromSymbol(pb, HIP_SYMBOL(nd_probe_buf), sizeof(pb)));
        const NdNode& v = T.nodes[n0 + (S.lv[l].n1 - n0) / 2];
        printf("  level %2d (%d fronts, p %d b %d): descriptor+zero %.2f us, matrix entries %.2f, children %.2f, factor %.2f, write %.2f, total %.2f\n", l,
               S.lv[l].n1 - n0, v.p, v.b, (pb[1] - pb[0]) * 0.01, (pb[2] - pb[1]) * 0.01, (pb[3] - pb[2]) * 0.01, (pb[4] - pb[3]) * 0.01, (pb[5] - pb[4]) * 0.01,
               (pb[5] - pb[0]) * 0.01);
    }
#endif
    std::vector<double> x(tot);
    std::vector<int> fail(nimg);
    CK(hipMemcpy(x.data(), d_vec, tot * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(fail.data(), d_fail, nimg * sizeof(int), hipMemcpyDeviceToHost));
    double err = 0.0, nrm = 0.0;
    for (int k = 0; k < nimg; ++k)
        for (int g = 0; g < n; ++g) { err = std::fmax(err, std::fabs(x[(size_t)k * n + g] - xt[g])); nrm = std::fmax(nrm, std::fabs(xt[g])); }
    printf("  solution error %.2e, fail[0] %d\n", err / nrm, fail[0]);
    return (err / nrm <= 1e-7 && !fail[0]) ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "time"))
        return timing(atoi(argv[2]), atoi(argv[3]), argc > 4 && !strcmp(argv[4], "sr"), argc > 5 ? atoi(argv[5]) : 0);
    int bad = 0;
    bad += check(1, 1, false, 32, 2, 0.0);
    bad += check(5, 4, false, 1, 2, 0.0);
    bad += check(16, 16, false, 8, 3, 1e6);
    bad += check(33, 17, true, 8, 2, 1e6);
    bad += check(40, 64, false, 32, 3, 1e6);        // small regime only
    bad += check(96, 50, true, 32, 2, 1e6);
    bad += check(150, 139, false, 32, 2, 1e6);      // large-regime levels, one pivot panel
    bad += check(300, 260, false, 16, 2, 1e6);      // two pivot panels (root separator 300 > 256: three)
    bad += check(200, 180, true, 32, 2, 1e4);       // separators of width 2: pivot blocks up to 400
    bad += check(7, 500, false, 32, 2, 1e6);
    bad += check_lu(1, 1, true, 32, 2, 0.0);
    bad += check_lu(5, 4, false, 1, 2, 0.0);
    bad += check_lu(33, 17, true, 8, 2, 1e6);
    bad += check_lu(40, 64, false, 32, 3, 1e6);     // small regime only
    bad += check_lu(96, 50, true, 32, 2, 1e6);
    bad += check_lu(128, 128, true, 32, 3, 1e6);    // the shape of the reference's batches
    bad += check_lu(150, 139, false, 32, 2, 1e6);
    bad += check_lu(200, 180, true, 32, 2, 1e4);    // pivot blocks of several panels
    bad += check_lu(7, 500, true, 32, 2, 1e6);
    printf(bad ? "nd_unit: %d FAILED\n" : "nd_unit: all ok\n", bad);
    return bad ? 1 : 0;
}
