// Dev tool: unit checks of the block-cyclic-reduction kernels against plain host loops.
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I bpldenoising_amd/csrc tools/bcr_unit.hip -o gpurun_out/bcr_unit
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define BCR_DBG 1
#include "adjoint_bcr_kernels.hpp"

using namespace bpltv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

static double rnd() { return (double)rand() / RAND_MAX - 0.5; }
static int nfail = 0;
static void report(const char* what, double err, double tol) {
    printf("%-44s err %.3e  %s\n", what, err, err <= tol ? "ok" : "FAIL");
    if (!(err <= tol)) ++nfail;
}

// dense helpers, column major ld = n
static void matmul(int n, const double* A, bool ta, const double* B, bool tb, double* C) {
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += (ta ? A[k + n * r] : A[r + n * k]) * (tb ? B[c + n * k] : B[k + n * c]);
            C[r + n * c] = s;
        }
}

static void test_potrf(int MP) {
    const int N = 1, O = 2;
    const size_t bsz = (size_t)MP * MP;
    std::vector<double> A(O * bsz), G(bsz);
    for (int k = 0; k < O; ++k) {
        for (auto& g : G) g = rnd();
        matmul(MP, G.data(), false, G.data(), true, A.data() + k * bsz);
        for (int i = 0; i < MP; ++i) A[k * bsz + i + MP * i] += 0.5 + (k ? 1e6 * (i % 3 == 0) : 0.0);
    }
    double *dD, *dDT; int* dfail;
    CK(hipMalloc(&dD, O * bsz * 8)); CK(hipMalloc(&dDT, O * bsz * 8)); CK(hipMalloc(&dfail, O * 4));
    CK(hipMemcpy(dD, A.data(), O * bsz * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dfail, 0, O * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcr_potrf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bcr_potrf_lds(128)));
    hipLaunchKernelGGL(bcr_potrf_kernel, dim3(1, O), dim3(BCR_PT), bcr_potrf_lds(MP), 0, dD, dDT, N, MP, 0, dfail);
    CK(hipDeviceSynchronize());
    std::vector<double> Li(O * bsz), LiT(O * bsz), T1(bsz), T2(bsz);
    int fail[2];
    CK(hipMemcpy(Li.data(), dD, O * bsz * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(LiT.data(), dDT, O * bsz * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(fail, dfail, 8, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, e3 = 0;
    for (int k = 0; k < O; ++k) {
        matmul(MP, Li.data() + k * bsz, false, A.data() + k * bsz, false, T1.data());
        matmul(MP, T1.data(), false, Li.data() + k * bsz, true, T2.data());
        for (int c = 0; c < MP; ++c)
            for (int r = 0; r < MP; ++r) {
                e1 = fmax(e1, fabs(T2[r + MP * c] - (r == c)));
                if (r < c) e2 = fmax(e2, fabs(Li[k * bsz + r + MP * c]));
                e3 = fmax(e3, fabs(Li[k * bsz + r + MP * c] - LiT[k * bsz + c + MP * r]));
            }
    }
    char name[96];
    snprintf(name, sizeof name, "potrf MP=%d  |Linv A Linv^T - I|", MP); report(name, e1, 1e-9);
    snprintf(name, sizeof name, "potrf MP=%d  upper zeros", MP); report(name, e2, 0.0);
    snprintf(name, sizeof name, "potrf MP=%d  transpose copy", MP); report(name, e3, 0.0);
    report("potrf fail flags", (double)(fail[0] + fail[1]), 0.0);
    CK(hipFree(dD)); CK(hipFree(dDT)); CK(hipFree(dfail));
}

static void test_gemm(int MP) {
    const int N = 3, O = 2, s = 1;
    const size_t bsz = (size_t)MP * MP, arr = (size_t)O * N * bsz;
    std::vector<double> h(7 * arr);
    for (auto& v : h) v = rnd();
    // Linv lower triangular
    for (int k = 0; k < O * N; ++k)
        for (int c = 0; c < MP; ++c)
            for (int r = 0; r < c; ++r) h[k * bsz + r + MP * c] = 0.0;
    double* d;
    CK(hipMalloc(&d, BcrArrays::doubles(MP, N, O, MP) * 8));
    CK(hipMemcpy(d, h.data(), 7 * arr * 8, hipMemcpyHostToDevice));
    BcrArrays B = BcrArrays::carve(d, MP, N, O, MP);
    const unsigned nt = (MP + 63) / 64;
    hipLaunchKernelGGL(bcr_x_kernel, dim3(bg_grid(2u * bcr_nelim(N, 0) * O, nt)), dim3(BG_T), 0, 0, B.Li, B.C, B.XA, B.XAT, B.XB, B.XBT, N, O, MP, s);
    CK(hipDeviceSynchronize());
    std::vector<double> g(7 * arr);
    CK(hipMemcpy(g.data(), d, 7 * arr * 8, hipMemcpyDeviceToHost));
    auto blk = [&](std::vector<double>& v, int which, int img, int j) { return v.data() + which * arr + ((size_t)img * N + j) * bsz; };
    std::vector<double> T(bsz);
    double eA = 0, eB = 0, eT = 0;
    for (int img = 0; img < O; ++img) {
        matmul(MP, blk(h, 0, img, 1), false, blk(h, 2, img, 0), false, T.data());  // Linv1 * C0
        for (size_t e = 0; e < bsz; ++e) eA = fmax(eA, fabs(T[e] - blk(g, 3, img, 1)[e]));
        matmul(MP, blk(h, 0, img, 1), false, blk(h, 2, img, 1), true, T.data());   // Linv1 * C1^T
        for (size_t e = 0; e < bsz; ++e) eB = fmax(eB, fabs(T[e] - blk(g, 5, img, 1)[e]));
        for (int c = 0; c < MP; ++c)
            for (int r = 0; r < MP; ++r) {
                eT = fmax(eT, fabs(blk(g, 3, img, 1)[r + MP * c] - blk(g, 4, img, 1)[c + MP * r]));
                eT = fmax(eT, fabs(blk(g, 5, img, 1)[r + MP * c] - blk(g, 6, img, 1)[c + MP * r]));
            }
    }
    char name[96];
    snprintf(name, sizeof name, "x kernel MP=%d  XA", MP); report(name, eA, 1e-12 * MP);
    snprintf(name, sizeof name, "x kernel MP=%d  XB", MP); report(name, eB, 1e-12 * MP);
    snprintf(name, sizeof name, "x kernel MP=%d  transposes", MP); report(name, eT, 0.0);
    // update kernel on the device's own X
    hipLaunchKernelGGL(bcr_upd_kernel, dim3(bg_grid(2u * bcr_nsurv(N, 0) * O, nt)), dim3(BG_T), 0, 0, B.Li, B.C, B.XA, B.XAT, B.XB, B.XBT, N, O, MP, s);
    CK(hipDeviceSynchronize());
    std::vector<double> u(7 * arr);
    CK(hipMemcpy(u.data(), d, 7 * arr * 8, hipMemcpyDeviceToHost));
    double eD0 = 0, eD2 = 0, eC = 0;
    for (int img = 0; img < O; ++img) {
        matmul(MP, blk(g, 3, img, 1), true, blk(g, 3, img, 1), false, T.data());  // XA^T XA
        for (size_t e = 0; e < bsz; ++e) eD0 = fmax(eD0, fabs(blk(g, 0, img, 0)[e] - T[e] - blk(u, 0, img, 0)[e]));
        matmul(MP, blk(g, 5, img, 1), true, blk(g, 5, img, 1), false, T.data());  // XB^T XB
        for (size_t e = 0; e < bsz; ++e) eD2 = fmax(eD2, fabs(blk(g, 0, img, 2)[e] - T[e] - blk(u, 0, img, 2)[e]));
        matmul(MP, blk(g, 5, img, 1), true, blk(g, 3, img, 1), false, T.data());  // XB^T XA
        for (size_t e = 0; e < bsz; ++e) eC = fmax(eC, fabs(-T[e] - blk(u, 2, img, 0)[e]));
    }
    snprintf(name, sizeof name, "upd kernel MP=%d  D_a (lower nb)", MP); report(name, eD0, 1e-12 * MP);
    snprintf(name, sizeof name, "upd kernel MP=%d  D_b (upper nb)", MP); report(name, eD2, 1e-12 * MP);
    snprintf(name, sizeof name, "upd kernel MP=%d  new coupling", MP); report(name, eC, 1e-12 * MP);
    CK(hipFree(d));
}

// full factor + solve of random block tridiagonal SPD systems given by their four diagonals (the
// library's band4 planes); residual against the dense blocks.  op0: level 0 in operator form
// (bcr_factor_band4_launch), otherwise all levels dense from bcr_init_kernel's blocks.
static void test_solve(int M, int N, int O, bool op0) {
    const int MP = (M + 15) / 16 * 16;
    const size_t bsz = (size_t)MP * MP, arr = (size_t)O * N * bsz;
    const size_t npx = (size_t)M * N, tot = npx * O;
    std::vector<double> b4(4 * tot, 0.0);
    for (int img = 0; img < O; ++img)
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                const size_t q = img * npx + (size_t)j * M + i;
                b4[q] = 4.0 + 2.0 * (rnd() + 0.5) + ((i + j + img) % 7 == 0 ? 1e8 : 0.0);
                if (i + 1 < M) b4[tot + q] = rnd();
                if (j + 1 < N) {
                    b4[3 * tot + q] = rnd();
                    if (i >= 1) b4[2 * tot + q] = rnd();
                }
            }
    std::vector<double> D(arr, 0.0), C(arr, 0.0);
    for (int k = 0; k < O * N; ++k) {
        const int j = k % N;
        const size_t q0 = (size_t)k * M;
        for (int c = 0; c < MP; ++c)
            for (int r = 0; r < MP; ++r) {
                double d = 0, cc = 0;
                if (r < M && c < M) {
                    if (r == c) d = b4[q0 + r];
                    else if (r == c + 1) d = b4[tot + q0 + c];
                    else if (c == r + 1) d = b4[tot + q0 + r];
                    if (j + 1 < N) {
                        if (r == c) cc = b4[3 * tot + q0 + c];
                        else if (r == c - 1) cc = b4[2 * tot + q0 + c];
                    }
                } else if (r == c) d = 1.0;
                D[k * bsz + r + MP * c] = d;
                C[k * bsz + r + MP * c] = cc;
            }
    }
    std::vector<double> rhs((size_t)O * N * M), acc0((size_t)O * N * M);
    for (auto& v : rhs) v = rnd();
    for (auto& v : acc0) v = rnd();
    const size_t nd = BcrArrays::doubles(M, N, O, MP);
    double *d, *dv, *da, *db4; int* dfail;
    CK(hipMalloc(&d, nd * 8)); CK(hipMalloc(&dv, rhs.size() * 8)); CK(hipMalloc(&da, rhs.size() * 8)); CK(hipMalloc(&dfail, O * 4));
    CK(hipMalloc(&db4, b4.size() * 8));
    CK(hipMemset(d, 0, nd * 8));
    CK(hipMemset(dfail, 0, O * 4));
    BcrArrays B = BcrArrays::carve(d, M, N, O, MP);
    CK(hipMemcpy(db4, b4.data(), b4.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, rhs.data(), rhs.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, acc0.data(), rhs.size() * 8, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bcr0_schur_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bcr0_schur_lds(128)));
    if (op0) {
        bcr_factor_band4_launch(0, B, db4, M, N, O, MP, dfail);
        bcr_solve_launch(0, B, M, N, O, MP, dv, da, db4);
    } else {
        hipLaunchKernelGGL(bcr_init_kernel, dim3(N, O), dim3(256), 0, 0, db4, M, N, O, MP, B.Li, B.C);
        bcr_factor_launch(0, B, N, O, MP, dfail);
        bcr_solve_launch(0, B, M, N, O, MP, dv, da);
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<double> p(rhs.size()), acc(rhs.size());
    std::vector<int> fail(O);
    CK(hipMemcpy(p.data(), dv, rhs.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(acc.data(), da, rhs.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(fail.data(), dfail, O * 4, hipMemcpyDeviceToHost));
    double eres = 0, eacc = 0, pn = 0;
    for (int img = 0; img < O; ++img)
        for (int j = 0; j < N; ++j)
            for (int r = 0; r < M; ++r) {
                const size_t k = (size_t)img * N + j;
                double s = 0;
                for (int c = 0; c < M; ++c) {
                    s += D[k * bsz + r + MP * c] * p[k * M + c];
                    if (j + 1 < N) s += C[k * bsz + c + MP * r] * p[(k + 1) * M + c];      // A[j, j+1] = C_j^T
                    if (j > 0) s += C[(k - 1) * bsz + r + MP * c] * p[(k - 1) * M + c];    // A[j, j-1] = C_{j-1}
                }
                eres = fmax(eres, fabs(s - rhs[k * M + r]) / (1.0 + fabs(D[k * bsz + r + MP * r]) * fabs(p[k * M + r])));
                eacc = fmax(eacc, fabs(acc[k * M + r] - acc0[k * M + r] - p[k * M + r]));
                pn = fmax(pn, fabs(p[k * M + r]));
            }
    char name[96];
    snprintf(name, sizeof name, "solve%s M=%d N=%d O=%d  scaled residual", op0 ? " op0" : "", M, N, O); report(name, eres, 1e-10);
    snprintf(name, sizeof name, "solve%s M=%d N=%d O=%d  accumulate", op0 ? " op0" : "", M, N, O); report(name, eacc, 1e-15 * (1 + pn));
    int f = 0; for (int v : fail) f += v;
    report("solve fail flags", (double)f, 0.0);
    CK(hipFree(d)); CK(hipFree(dv)); CK(hipFree(da)); CK(hipFree(dfail)); CK(hipFree(db4));
}

// timing of the product kernels on the level-1 shape of a 10 x 128^2 batch (values are irrelevant)
static void bench_products() {
    const int M = 128, N = 128, O = 10, MP = 128, s = 2, l = 1;
    const size_t nd = BcrArrays::doubles(M, N, O, MP);
    double* d;
    CK(hipMalloc(&d, nd * 8));
    CK(hipMemset(d, 0, nd * 8));
    BcrArrays B = BcrArrays::carve(d, M, N, O, MP);
    const unsigned nt = 2;
    const int ne = bcr_nelim(N, l), ns = bcr_nsurv(N, l);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fl_x = 2.0 * ne * O * 2.0 * 128 * 128 * 128 * 0.75, fl_u = (2.0 * ns - 1 + ne) * O * 2.0 * 128 * 128 * 128;
    for (int dbg : {0, 1, 2, 4, 8, 3, 6, 7, 15}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bcr_dbg), &dbg, sizeof(int)));
        float tx = 0, tu = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(bcr_x_kernel, dim3(bg_grid(2u * ne * O, nt)), dim3(BG_T), 0, 0, B.Li, B.C, B.XA, B.XAT, B.XB, B.XBT, N, O, MP, s);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&tx, e0, e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(bcr_upd_kernel, dim3(bg_grid(2u * ns * O, nt)), dim3(BG_T), 0, 0, B.Li, B.C, B.XA, B.XAT, B.XB, B.XBT, N, O, MP, s);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&tu, e0, e1));
        }
        printf("dbg %2d: x %7.1f us (%5.1f TFLOP/s)   upd %7.1f us (%5.1f TFLOP/s)\n", dbg, tx * 1e3, fl_x / tx / 1e9, tu * 1e3, fl_u / tu / 1e9);
    }
    int z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(bcr_dbg), &z, sizeof(int)));
    CK(hipFree(d));
}

int main(int argc, char** argv) {
    if (argc > 1) { bench_products(); return 0; }
    srand(1234);
    test_potrf(16); test_potrf(48); test_potrf(128);
    test_gemm(16); test_gemm(80); test_gemm(128);
    for (int op0 = 0; op0 < 2; ++op0) {
        test_solve(5, 2, 1, op0); test_solve(1, 4, 1, op0); test_solve(20, 5, 2, op0); test_solve(16, 3, 1, op0);
        test_solve(37, 29, 2, op0); test_solve(128, 13, 1, op0); test_solve(100, 64, 2, op0); test_solve(128, 128, 1, op0);
    }
    printf(nfail ? "FAILED %d checks\n" : "all ok\n", nfail);
    return nfail ? 1 : 0;
}
