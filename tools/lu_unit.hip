// lu_unit.hip -- kernel-level check of the HBM band solvers against host loops (developer tool + GPU test):
//   HbLuSolver  (hb_lu_solver.hpp): random non-symmetric diagonally dominant band matrices
//   HbBandSolver (hb_band_solver.hpp): random SPD band matrices, twisted and not
// given as diagonals (BandDiags).  Prints the max relative solution error; "all ok" when every case passes.
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I bpldenoising_amd/csrc tools/lu_unit.hip -o tools/_bin/lu_unit
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "hb_band_solver.hpp"
#include "hb_lu_solver.hpp"
using namespace bpltv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// dense host solve (Gaussian elimination without pivoting; the matrices are diagonally dominant)
static void host_solve(int n, std::vector<double> A, std::vector<double>& x) {
    for (int k = 0; k < n; ++k) {
        const double p = A[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = A[(size_t)i * n + k] / p;
            if (f == 0.0) continue;
            for (int j = k; j < n; ++j) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
            x[i] -= f * x[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double v = x[k];
        for (int j = k + 1; j < n; ++j) v -= A[(size_t)k * n + j] * x[j];
        x[k] = v / A[(size_t)k * n + k];
    }
}

static bool run_case(int n, int bw, const std::vector<int>& offs, bool sym, int O, bool use_lu) {
    std::mt19937_64 rng(1234 + n + bw);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const int nd = (int)offs.size();
    std::vector<double> pl((size_t)nd * O * n, 0.0), pu((size_t)nd * O * n, 0.0), b((size_t)O * n);
    std::vector<std::vector<double>> dense(O, std::vector<double>((size_t)n * n, 0.0));
    for (int img = 0; img < O; ++img) {
        for (int t = 1; t < nd; ++t)
            for (int c = 0; c + offs[t] < n; ++c) {
                const double lo = U(rng), up = sym ? lo : U(rng);
                pl[((size_t)t * O + img) * n + c] = lo;
                pu[((size_t)t * O + img) * n + c] = up;
                dense[img][(size_t)(c + offs[t]) * n + c] += lo;
                dense[img][(size_t)c * n + c + offs[t]] += up;
            }
        for (int c = 0; c < n; ++c) {
            double s = 1.0;
            for (int j = 0; j < n; ++j) if (j != c) s += std::fabs(dense[img][(size_t)c * n + j]) + (sym ? 0.0 : std::fabs(dense[img][(size_t)j * n + c]));
            pl[((size_t)0 * O + img) * n + c] = s;
            dense[img][(size_t)c * n + c] = s;
        }
        for (int c = 0; c < n; ++c) b[(size_t)img * n + c] = U(rng);
    }
    double *d_pl, *d_pu, *d_v, *d_s; int* d_fail;
    CK(hipMalloc(&d_pl, pl.size() * 8)); CK(hipMalloc(&d_pu, pu.size() * 8));
    CK(hipMalloc(&d_v, b.size() * 8)); CK(hipMalloc(&d_s, b.size() * 8)); CK(hipMalloc(&d_fail, O * sizeof(int)));
    CK(hipMemcpy(d_pl, pl.data(), pl.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pu, pu.data(), pu.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_v, b.data(), b.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_fail, 0, O * sizeof(int)));
    hipStream_t st; CK(hipStreamCreate(&st));
    BandDiags DL, DU;
    DL.planes = d_pl; DL.tot = (size_t)O * n; DL.nd = nd;
    for (int t = 0; t < nd; ++t) DL.off[t] = offs[t];
    DU = DL; DU.planes = d_pu;
    int rc = 0;
    HbLuSolver lu; HbBandSolver ch;
    if (use_lu) {
        rc = lu.alloc(bw, n, O, st);
        if (!rc) rc = lu.factor(DL, DU, d_fail);
        if (!rc) lu.solve(d_v, nullptr, d_s);
    } else {
        ch.options_from_env();
        rc = ch.alloc(bw, n, O, st);
        if (!rc) rc = ch.factor(DL, d_fail);
        if (!rc) ch.solve(d_v, nullptr, d_s);
    }
    if (rc) { printf("solver rc %d: %s\n", rc, use_lu ? lu.err.c_str() : ch.err.c_str()); return false; }
    CK(hipStreamSynchronize(st));
    if (use_lu && getenv("LU_DEBUG") && n > 128 && n <= 256 && O == 1) {   // two panels: compare the pieces with host arithmetic
        const int W = bw + 1;
        std::vector<double> bl((size_t)n * W), bu((size_t)n * W), ai(128 * 128);
        CK(hipMemcpy(bl.data(), lu.bandL, bl.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(bu.data(), lu.bandU, bu.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ai.data(), lu.Ainv(), ai.size() * 8, hipMemcpyDeviceToHost));
        const std::vector<double>& A = dense[0];
        // host inverse of the first block by solving for unit vectors
        std::vector<double> A11(128 * 128), inv(128 * 128);
        for (int r = 0; r < 128; ++r) for (int c = 0; c < 128; ++c) A11[(size_t)r * 128 + c] = A[(size_t)r * n + c];
        for (int c = 0; c < 128; ++c) { std::vector<double> e(128, 0.0); e[c] = 1.0; host_solve(128, A11, e); for (int r = 0; r < 128; ++r) inv[(size_t)r * 128 + c] = e[r]; }
        double eai = 0, ep = 0, esl = 0, esu = 0;
        for (int r = 0; r < 128; ++r) for (int c = 0; c < 128; ++c) eai = std::fmax(eai, std::fabs(ai[r + 128 * c] - inv[(size_t)r * 128 + c]));
        const int m2 = n - 128;
        std::vector<double> P((size_t)m2 * 128, 0.0);
        for (int r = 0; r < m2; ++r) for (int c = 0; c < 128; ++c) { double s2 = 0; for (int k = 0; k < 128; ++k) s2 += A[(size_t)(128 + r) * n + k] * inv[(size_t)k * 128 + c]; P[(size_t)r * 128 + c] = s2; }
        for (int r = 0; r < m2; ++r) for (int c = 0; c < 128; ++c) { const int R = 128 + r, d = R - c; if (d <= bw) ep = std::fmax(ep, std::fabs(bl[(size_t)c * W + d] - P[(size_t)r * 128 + c])); }
        for (int r = 0; r < m2; ++r) for (int c = 0; c < m2; ++c) {
            double s2 = A[(size_t)(128 + r) * n + 128 + c];
            for (int k = 0; k < 128; ++k) s2 -= P[(size_t)r * 128 + k] * A[(size_t)k * n + 128 + c];
            const int R = 128 + r, Cc = 128 + c;
            if (r >= c && r - c <= bw) esl = std::fmax(esl, std::fabs(bl[(size_t)Cc * W + (R - Cc)] - s2));
            if (c >= r && c - r <= bw) esu = std::fmax(esu, std::fabs(bu[(size_t)R * W + (Cc - R)] - s2));
        }
        printf("  debug: |Ainv0 err| %.2e  |L21 err| %.2e  |S lower err| %.2e  |S upper err| %.2e\n", eai, ep, esl, esu);
        // emulate the block substitutions on the host with the DEVICE factors
        std::vector<double> ai1(128 * 128), xd(n), y(b.begin(), b.begin() + n), xe(n, 0.0);
        CK(hipMemcpy(ai1.data(), lu.Ainv() + 128 * 128, ai1.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(xd.data(), d_v, n * 8, hipMemcpyDeviceToHost));
        for (int R = 128; R < n; ++R) for (int c = 0; c < 128; ++c) if (R - c <= bw) y[R] -= bl[(size_t)c * W + (R - c)] * y[c];
        for (int r = 0; r < m2; ++r) { double s2 = 0; for (int k = 0; k < m2; ++k) s2 += ai1[r + 128 * k] * y[128 + k]; xe[128 + r] = s2; }
        for (int k = 0; k < 128; ++k) for (int c = 0; c < m2; ++c) if (128 + c - k <= bw) y[k] -= bu[(size_t)k * W + (128 + c - k)] * xe[128 + c];
        for (int r = 0; r < 128; ++r) { double s2 = 0; for (int k = 0; k < 128; ++k) s2 += ai[r + 128 * k] * y[k]; xe[r] = s2; }
        {   // A12 rows kept in bandU?  Ainv(1) == inverse of the Schur complement?
            double ea12 = 0, eai1 = 0;
            for (int k = 0; k < 128; ++k) for (int c = 0; c < m2; ++c) if (128 + c - k <= bw) ea12 = std::fmax(ea12, std::fabs(bu[(size_t)k * W + (128 + c - k)] - A[(size_t)k * n + 128 + c]));
            std::vector<double> Sd((size_t)m2 * m2, 0.0), invS((size_t)m2 * m2);
            for (int r = 0; r < m2; ++r) for (int c = 0; c < m2; ++c) {
                double s2 = A[(size_t)(128 + r) * n + 128 + c];
                for (int k = 0; k < 128; ++k) s2 -= P[(size_t)r * 128 + k] * A[(size_t)k * n + 128 + c];
                Sd[(size_t)r * m2 + c] = s2;
            }
            for (int c = 0; c < m2; ++c) { std::vector<double> e(m2, 0.0); e[c] = 1.0; host_solve(m2, Sd, e); for (int r = 0; r < m2; ++r) invS[(size_t)r * m2 + c] = e[r]; }
            for (int r = 0; r < m2; ++r) for (int c = 0; c < m2; ++c) eai1 = std::fmax(eai1, std::fabs(ai1[r + 128 * c] - invS[(size_t)r * m2 + c]));
            double smax = 0; int far = 0;
            for (int r = 0; r < m2; ++r) for (int c = 0; c < m2; ++c) if (std::abs(r - c) > bw && std::fabs(Sd[(size_t)r * m2 + c]) > 1e-14) { ++far; smax = std::fmax(smax, std::fabs(Sd[(size_t)r * m2 + c])); }
            printf("  debug: |A12 in bandU err| %.2e  |Ainv1 - inv(S)| %.2e  Schur entries beyond the band: %d (max %.2e)\n", ea12, eai1, far, smax);
        }
        std::vector<double> xr(b.begin(), b.begin() + n);
        host_solve(n, A, xr);
        double e1 = 0, e2 = 0, e2a = 0, e2b = 0;
        for (int c = 0; c < n; ++c) { e1 = std::fmax(e1, std::fabs(xe[c] - xr[c])); e2 = std::fmax(e2, std::fabs(xd[c] - xe[c]));
            if (c < 128) e2a = std::fmax(e2a, std::fabs(xd[c] - xe[c])); else e2b = std::fmax(e2b, std::fabs(xd[c] - xe[c])); }
        printf("  debug: host emulation with device factors vs reference %.2e; device solve vs emulation %.2e (block0 %.2e, block1 %.2e)\n", e1, e2, e2a, e2b);
    }
    std::vector<double> x(b.size());
    std::vector<int> fl(O);
    CK(hipMemcpy(x.data(), d_v, x.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(fl.data(), d_fail, O * sizeof(int), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int img = 0; img < O; ++img) {
        std::vector<double> xr(b.begin() + (size_t)img * n, b.begin() + (size_t)(img + 1) * n);
        host_solve(n, dense[img], xr);
        double num = 0.0, den = 0.0;
        for (int c = 0; c < n; ++c) { num = std::fmax(num, std::fabs(x[(size_t)img * n + c] - xr[c])); den = std::fmax(den, std::fabs(xr[c])); }
        worst = std::fmax(worst, num / den);
        if (fl[img]) { printf("  fail flag %d image %d\n", fl[img], img); worst = 1.0; }
    }
    printf("%-8s n %5d bw %4d sym %d O %d%s: max rel err %.2e %s\n", use_lu ? "LU" : "Cholesky", n, bw, (int)sym, O,
           (!use_lu && ch.twisted) ? " (twisted)" : "", worst, worst < 1e-10 ? "ok" : "FAILED");
    if (use_lu) lu.release(); else ch.release();
    for (void* q : {(void*)d_pl, (void*)d_pu, (void*)d_v, (void*)d_s, (void*)d_fail}) (void)hipFree(q);
    (void)hipStreamDestroy(st);
    return worst < 1e-10;
}

int main() {
    bool ok = true;
    // offsets as the sum-of-regularisers assembly produces them for M = 20, 40, 96
    for (int M : {20, 40, 96}) {
        const std::vector<int> offs = {0, 1, 2, M - 1, M, M + 1, 2 * M};
        const int n = M * (M == 96 ? 30 : 24), bw = 2 * M;
        ok &= run_case(n, bw, offs, true, 2, false);
        ok &= run_case(n, bw, offs, true, 2, true);
        ok &= run_case(n, bw, offs, false, 2, true);
    }
    ok &= run_case(100, 40, {0, 1, 2, 19, 20, 21, 40}, false, 1, true);    // one panel: the block inverse alone
    ok &= run_case(128, 40, {0, 1, 2, 19, 20, 21, 40}, false, 1, true);
    ok &= run_case(200, 40, {0, 1, 2, 19, 20, 21, 40}, false, 1, true);    // two panels
    ok &= run_case(256, 40, {0, 1, 2, 19, 20, 21, 40}, false, 1, true);
    ok &= run_case(300, 150, {0, 1, 149, 150}, true, 1, false);    // TV wide-image shape, not twisted
    ok &= run_case(4000, 200, {0, 1, 199, 200}, true, 2, false);   // twisted
    ok &= run_case(1501, 151, {0, 1, 150, 151}, true, 2, false);   // odd bandwidth and order: 8-byte loads in the blocked substitutions
    ok &= run_case(1100, 150, {0, 1, 149, 150}, true, 3, false);   // not twisted, two groups of four blocks + a ragged tail
    ok &= run_case(5000, 300, {0, 1, 299, 300}, true, 8, false);   // eight problems per launch (one per XCD), three tile rows
    printf(ok ? "all ok\n" : "FAILURES\n");
    return ok ? 0 : 1;
}
