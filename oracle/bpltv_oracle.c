/*
 * bpltv_oracle.c -- CPU restatement of the TV learning-function path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (as the checker / the timed CPU baseline).  Nothing under bpldenoising_amd/ links or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE.  The reference is Julia; its PDHG loop (`op_denoise_pdps`)
 * lives in the un-vendored, un-pinned package VariationalImaging and its tests hold no expected
 * values.  This file therefore *defines* the arithmetic ("spec v2") that the HIP
 * kernels reproduce bit for bit, and is itself pinned by
 *   - oracle/np_twin.py (numpy restatement of the same recurrence; literal scipy assembly of the
 *     reference's adjoint systems) through tests/golden fixtures,
 *   - mathematical certificates in tests/ (duality gap, adjointness, closed-form cases, finite
 *     differences).
 *
 * Layout: Julia column-major M x N x O, element (i,j,k) at i + M*j + M*N*k.
 * Gradient component 1 differences dimension 1 (i, contiguous), component 2 dimension 2 (j)
 * (/root/reference/src/TVLearningFunctionOp.jl:26-36).
 *
 * Build: see oracle/Makefile (-ffp-contract=off: every fused multiply-add below is an explicit
 * fma(), so that CPU and GPU round identically).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BPLO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Step-size table.  Constants: /root/reference/src/TVLearningFunctionVec.jl:33-43
 * (tau0 = 5, sigma0 = 0.99/5, accel = true); recurrence: SURVEY.md 8(a) row A2
 * (L = sqrt(8) bound on ||grad||, gamma = 1 strong convexity of the fidelity term).
 * Row k = {tau, sigma, omega, 1/(1+tau), 1+omega} used in iteration k.
 * ---------------------------------------------------------------------------------------- */
BPLO_API void bplo_step_table_L(int maxiter, double tau0, double sigma0, int accel, double L, double *tab)
{
    double tau = tau0 / L, sigma = sigma0 / L;
    const double gamma = 1.0;
    for (int k = 0; k < maxiter; ++k) {
        double omega = accel ? 1.0 / sqrt(1.0 + 2.0 * gamma * tau) : 1.0;
        tab[5 * k + 0] = tau;
        tab[5 * k + 1] = sigma;
        tab[5 * k + 2] = omega;
        tab[5 * k + 3] = 1.0 / (1.0 + tau);
        tab[5 * k + 4] = 1.0 + omega;
        if (accel) {
            tau = tau * omega;
            sigma = sigma / omega;
        }
    }
}

BPLO_API void bplo_step_table(int maxiter, double tau0, double sigma0, int accel, double *tab)
{
    bplo_step_table_L(maxiter, tau0, sigma0, accel, sqrt(8.0), tab);
}

/* PatchOp: m x n parameter (column-major, am x an) -> pixel (i,j) uses x[(i*am)/M + am*((j*an)/N)].
 * /root/reference/src/TVLearningFunctionVec.jl:57-60 (PatchOp is external; piecewise-constant
 * upsampling per SURVEY.md 8(a) A3). */
static inline double alpha_at(const double *alpha, int am, int an, int M, int N, int i, int j)
{
    if (am == 1 && an == 1) return alpha[0];
    if (am == M && an == N) return alpha[i + (size_t)M * j];
    return alpha[(int)(((long)i * am) / M) + (size_t)am * (int)(((long)j * an) / N)];
}

BPLO_API void bplo_patch_upsample(const double *alpha, int am, int an, int M, int N, double *amap)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) amap[i + (size_t)M * j] = alpha_at(alpha, am, an, M, N, i, j);
}

/* calc_adjoint(PatchOp, g): sums of a pixel map over each patch (TVLearningFunctionVec.jl:253). */
BPLO_API void bplo_patch_adjoint(const double *g, int M, int N, int am, int an, double *out)
{
    for (int q = 0; q < am * an; ++q) out[q] = 0.0;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i)
            out[(int)(((long)i * am) / M) + (size_t)am * (int)(((long)j * an) / N)] += g[i + (size_t)M * j];
}

/* ------------------------------------------------------------------------------------------
 * FwdGradientOp and its adjoint (A4).  Forward differences, Neumann boundary.
 * ---------------------------------------------------------------------------------------- */
BPLO_API void bplo_grad_fwd(int M, int N, const double *x, double *d1, double *d2)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            d1[k] = (i < M - 1) ? x[k + 1] - x[k] : 0.0;
            d2[k] = (j < N - 1) ? x[k + M] - x[k] : 0.0;
        }
}

/* (G^T y)(i,j) = (y1(i-1,j) - y1(i,j)) + (y2(i,j-1) - y2(i,j)); y1(M-1,.) and y2(.,N-1) are
 * treated as zero (they multiply zero rows of G). */
BPLO_API void bplo_grad_fwd_T(int M, int N, const double *y1, const double *y2, double *r)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            double a = (i < M - 1) ? y1[k] : 0.0, am = (i > 0) ? y1[k - 1] : 0.0;
            double b = (j < N - 1) ? y2[k] : 0.0, bm = (j > 0) ? y2[k - M] : 0.0;
            r[k] = (am - a) + (bm - b);
        }
}

/* ------------------------------------------------------------------------------------------
 * Accelerated PDHG for ROF, fixed iteration count (A2, A3, A9).  "spec v2" arithmetic:
 *
 *   x-step  div = (y1[i-1,j] - y1[i,j]) + (y2[i,j-1] - y2[i,j])        (out of range -> 0)
 *           xn  = fma(-tau, div - f, x) * inv1ptau
 *           xb  = fma(-omega, x, opw * xn)
 *   y-step  d1  = xb[i+1,j] - xb[i,j]  (0 at i = M-1),  d2 likewise in j
 *           y1n = fma(sigma, d1, y1),  y2n = fma(sigma, d2, y2)
 *           rho != 0:  y?n /= (1 + sigma*rho/alpha_ij)
 *           n2  = fma(y2n, y2n, y1n*y1n);  if n2 > alpha_ij^2:  v = alpha_ij * rsqrt_nr(n2); y?n *= v
 *
 * rsqrt_nr (below) is 1/sqrt by four Newton steps from an integer-seeded guess: a fixed sequence of
 * IEEE multiplies and fmas, so CPU and GPU agree bit for bit while the GPU avoids the ~30-instruction
 * correctly-rounded sqrt + divide pair.  It differs from alpha/sqrt(n2) by <= 3 ulp (the projected
 * dual has |y| = alpha(1 +- 4e-16)).  Domain: n2 in the normal range, i.e. alpha > ~1e-150.
 *
 * x starts at f, y at 0.  Outputs: x (primal), optionally y1, y2.
 * ---------------------------------------------------------------------------------------- */
static inline double rsqrt_nr(double n2)
{
    union { double d; uint64_t u; } c;
    c.d = n2;
    c.u = 0x5FE6EB50C7B537A9ull - (c.u >> 1); /* relative error of the seed < 3.5e-2 */
    double r = c.d;
    const double h = 0.5 * n2;
    for (int k = 0; k < 4; ++k) { /* e -> 1.5 e^2: 1.8e-3, 4.6e-6, 3.2e-11, 1.5e-21 */
        double t = r * r;
        double w = fma(-h, t, 1.5);
        r = r * w;
    }
    return r;
}

BPLO_API double bplo_rsqrt_nr(double n2) { return rsqrt_nr(n2); }

/* One primal pass / one dual pass over the columns [j0, j1) of one image.  The primal pass reads y of the
 * previous iteration only, the dual pass reads xb of the current one only, so column blocks of a pass are
 * independent (used by bplo_pdhg_rows). */
static inline void pdhg_x_pass(int M, int j0, int j1, const double *f, const double *y1, const double *y2,
                               double tau, double omega, double inv1ptau, double opw, double *x, double *xb)
{
    for (int j = j0; j < j1; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            double y1m = (i > 0) ? y1[k - 1] : 0.0;
            double y2m = (j > 0) ? y2[k - M] : 0.0;
            double div = (y1m - y1[k]) + (y2m - y2[k]);
            double t = div - f[k];
            double xo = x[k];
            double xn = fma(-tau, t, xo) * inv1ptau;
            xb[k] = fma(-omega, xo, opw * xn);
            x[k] = xn;
        }
}

static inline void pdhg_y_pass(int M, int N, int j0, int j1, const double *alpha, int am, int an,
                               const double *xb, double sigma, double rho, double *y1, double *y2)
{
    for (int j = j0; j < j1; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            double d1 = (i < M - 1) ? xb[k + 1] - xb[k] : 0.0;
            double d2 = (j < N - 1) ? xb[k + M] - xb[k] : 0.0;
            double a = alpha_at(alpha, am, an, M, N, i, j);
            double y1n = fma(sigma, d1, y1[k]);
            double y2n = fma(sigma, d2, y2[k]);
            if (rho != 0.0) {
                double den = 1.0 + sigma * rho / a;
                y1n = y1n / den;
                y2n = y2n / den;
            }
            double n2 = fma(y2n, y2n, y1n * y1n);
            if (n2 > a * a) {
                double v = a * rsqrt_nr(n2);
                y1n = y1n * v;
                y2n = y2n * v;
            }
            y1[k] = y1n;
            y2[k] = y2n;
        }
}

static void pdhg_image(int M, int N, const double *f, const double *alpha, int am, int an,
                       const double *tab, int maxiter, double rho, double *x, double *y1, double *y2,
                       double *xb)
{
    const size_t n = (size_t)M * N;
    memcpy(x, f, n * sizeof(double));
    memset(y1, 0, n * sizeof(double));
    memset(y2, 0, n * sizeof(double));
    for (int it = 0; it < maxiter; ++it) {
        const double tau = tab[5 * it], sigma = tab[5 * it + 1], omega = tab[5 * it + 2];
        const double inv1ptau = tab[5 * it + 3], opw = tab[5 * it + 4];
        pdhg_x_pass(M, 0, N, f, y1, y2, tau, omega, inv1ptau, opw, x, xb);
        pdhg_y_pass(M, N, 0, N, alpha, am, an, xb, sigma, rho, y1, y2);
    }
}

/* ------------------------------------------------------------------------------------------
 * "spec v2f": the same recurrence in single precision, the checker of the opt-in dtype = 32 mode of the library
 * (include/bpltv.h, bpltv_create).  Inputs f and alpha and the f64 step table are rounded to float once; every
 * operation of pdhg_x_pass / pdhg_y_pass is repeated in float with fmaf; the projection uses three Newton steps
 * from the 32-bit seed 0x5F375A86 (e -> 1.5 e^2 from < 3.5e-2: 1.8e-3, 4.7e-6, 3.3e-11 < 2^-24).  The result is
 * widened to double.  Not a restatement of anything in the reference (which is Float64 only).
 * ---------------------------------------------------------------------------------------- */
static inline float rsqrt_nr_f(float n2)
{
    union { float f; uint32_t u; } c;
    c.f = n2;
    c.u = 0x5F375A86u - (c.u >> 1);
    float r = c.f;
    const float h = 0.5f * n2;
    for (int k = 0; k < 3; ++k) {
        float t = r * r;
        float w = fmaf(-h, t, 1.5f);
        r = r * w;
    }
    return r;
}

static inline float alpha_at_f(const float *alpha, int am, int an, int M, int N, int i, int j)
{
    if (am == 1 && an == 1) return alpha[0];
    if (am == M && an == N) return alpha[i + (size_t)M * j];
    return alpha[(int)(((long)i * am) / M) + (size_t)am * (int)(((long)j * an) / N)];
}

BPLO_API int bplo_pdhg_f32(int M, int N, int O, const double *f, const double *alpha, int am, int an,
                           double rho_, double tau0, double sigma0, int accel, int maxiter, double *x_out)
{
    if (M < 1 || N < 1 || O < 0 || maxiter < 0) return 1;
    const size_t n = (size_t)M * N, na = (size_t)am * an;
    double *tab = (double *)malloc(sizeof(double) * 5 * (size_t)(maxiter > 0 ? maxiter : 1));
    float *ff = (float *)malloc(n * sizeof(float)), *x = (float *)malloc(n * sizeof(float));
    float *y1 = (float *)malloc(n * sizeof(float)), *y2 = (float *)malloc(n * sizeof(float));
    float *xb = (float *)malloc(n * sizeof(float)), *al = (float *)malloc(na * sizeof(float));
    int rc = 0;
    if (!tab || !ff || !x || !y1 || !y2 || !xb || !al) {
        rc = 2;
    } else {
        bplo_step_table(maxiter, tau0, sigma0, accel, tab);
        for (size_t e = 0; e < na; ++e) al[e] = (float)alpha[e];
        const float rho = (float)rho_;
        for (int k = 0; k < O; ++k) {
            for (size_t e = 0; e < n; ++e) { ff[e] = (float)f[n * k + e]; x[e] = ff[e]; y1[e] = 0.0f; y2[e] = 0.0f; }
            for (int it = 0; it < maxiter; ++it) {
                const float tau = (float)tab[5 * it], sigma = (float)tab[5 * it + 1], omega = (float)tab[5 * it + 2];
                const float inv1ptau = (float)tab[5 * it + 3], opw = (float)tab[5 * it + 4];
                for (int j = 0; j < N; ++j)
                    for (int i = 0; i < M; ++i) {
                        size_t q = i + (size_t)M * j;
                        float y1m = (i > 0) ? y1[q - 1] : 0.0f;
                        float y2m = (j > 0) ? y2[q - M] : 0.0f;
                        float div = (y1m - y1[q]) + (y2m - y2[q]);
                        float t = div - ff[q];
                        float xo = x[q];
                        float xn = fmaf(-tau, t, xo) * inv1ptau;
                        xb[q] = fmaf(-omega, xo, opw * xn);
                        x[q] = xn;
                    }
                for (int j = 0; j < N; ++j)
                    for (int i = 0; i < M; ++i) {
                        size_t q = i + (size_t)M * j;
                        float d1 = (i < M - 1) ? xb[q + 1] - xb[q] : 0.0f;
                        float d2 = (j < N - 1) ? xb[q + M] - xb[q] : 0.0f;
                        float a = alpha_at_f(al, am, an, M, N, i, j);
                        float y1n = fmaf(sigma, d1, y1[q]);
                        float y2n = fmaf(sigma, d2, y2[q]);
                        if (rho != 0.0f) {
                            float den = 1.0f + sigma * rho / a;
                            y1n = y1n / den;
                            y2n = y2n / den;
                        }
                        float n2 = fmaf(y2n, y2n, y1n * y1n);
                        if (n2 > a * a) {
                            float v = a * rsqrt_nr_f(n2);
                            y1n = y1n * v;
                            y2n = y2n * v;
                        }
                        y1[q] = y1n;
                        y2[q] = y2n;
                    }
            }
            for (size_t e = 0; e < n; ++e) x_out[n * k + e] = (double)x[e];
        }
    }
    free(tab); free(ff); free(x); free(y1); free(y2); free(xb); free(al);
    return rc;
}

/* The same recurrence with the work of ONE iteration spread over images x column blocks ("OpenMP over
 * images then rows"): all threads sweep the primal pass, barrier, the dual pass,
 * barrier.  Same per-pixel operations as pdhg_image, hence the same bits.  Used by bench.py's cpu_baseline
 * leg so that "all cores" can mean the host's cores even for a 10-image batch. */
BPLO_API int bplo_pdhg_rows(int M, int N, int O, const double *f, const double *alpha, int am, int an,
                            double rho, double tau0, double sigma0, int accel, int maxiter, double *x_out,
                            int nthreads, int colblock)
{
    if (M < 1 || N < 1 || O < 1 || maxiter < 0 || colblock < 1) return 1;
    const size_t n = (size_t)M * N;
    double *tab = (double *)malloc(sizeof(double) * 5 * (size_t)(maxiter > 0 ? maxiter : 1));
    double *xb = (double *)malloc(n * O * sizeof(double));
    double *y1 = (double *)calloc(n * O, sizeof(double));
    double *y2 = (double *)calloc(n * O, sizeof(double));
    if (!tab || !xb || !y1 || !y2) { free(tab); free(xb); free(y1); free(y2); return 2; }
    bplo_step_table(maxiter, tau0, sigma0, accel, tab);
    memcpy(x_out, f, n * O * sizeof(double));
    const int nb = (N + colblock - 1) / colblock;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    for (int it = 0; it < maxiter; ++it) {
        const double tau = tab[5 * it], sigma = tab[5 * it + 1], omega = tab[5 * it + 2];
        const double inv1ptau = tab[5 * it + 3], opw = tab[5 * it + 4];
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(static)
#endif
        for (int k = 0; k < O; ++k)
            for (int b = 0; b < nb; ++b) {
                const int j0 = b * colblock, j1 = (j0 + colblock < N) ? j0 + colblock : N;
                pdhg_x_pass(M, j0, j1, f + n * k, y1 + n * k, y2 + n * k, tau, omega, inv1ptau, opw, x_out + n * k, xb + n * k);
            }
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(static)
#endif
        for (int k = 0; k < O; ++k)
            for (int b = 0; b < nb; ++b) {
                const int j0 = b * colblock, j1 = (j0 + colblock < N) ? j0 + colblock : N;
                pdhg_y_pass(M, N, j0, j1, alpha, am, an, xb + n * k, sigma, rho, y1 + n * k, y2 + n * k);
            }
    }
    free(tab); free(xb); free(y1); free(y2);
    return 0;
}

BPLO_API int bplo_pdhg(int M, int N, int O, const double *f, const double *alpha, int am, int an,
                       double rho, double tau0, double sigma0, int accel, int maxiter,
                       double *x_out, double *y1_out, double *y2_out, int nthreads)
{
    if (M < 1 || N < 1 || O < 0 || maxiter < 0) return 1;
    const size_t n = (size_t)M * N;
    double *tab = (double *)malloc(sizeof(double) * 5 * (size_t)(maxiter > 0 ? maxiter : 1));
    if (!tab) return 2;
    bplo_step_table(maxiter, tau0, sigma0, accel, tab);
    int fail = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int k = 0; k < O; ++k) {
        double *xb = (double *)malloc(n * sizeof(double));
        double *ty1 = y1_out ? y1_out + n * k : (double *)malloc(n * sizeof(double));
        double *ty2 = y2_out ? y2_out + n * k : (double *)malloc(n * sizeof(double));
        if (!xb || !ty1 || !ty2) {
            fail = 1;
        } else {
            pdhg_image(M, N, f + n * k, alpha, am, an, tab, maxiter, rho, x_out + n * k, ty1, ty2, xb);
        }
        free(xb);
        if (!y1_out) free(ty1);
        if (!y2_out) free(ty2);
    }
    (void)nthreads;
    free(tab);
    return fail ? 2 : 0;
}

/* ------------------------------------------------------------------------------------------
 * The recurrence in "spec v2" arithmetic (pdhg_x_pass / pdhg_y_pass, bit for bit) with the three run-time choices
 * of include/bpltv.h: bpltv_params.init / order / opnorm -- the degrees of freedom the reference leaves open
 * because op_denoise_pdps (/root/reference/src/TVLearningFunctionVec.jl:52,67) is not in its repository.
 *   init  0: x0 = f, 1: x0 = 0;   order 0: primal step first, 1: dual step first (y from xbar of the previous
 *   iteration -- xbar0 = x0 --, then x, then the over-relaxation);   L: operator-norm estimate (> 0).
 * init = order = 0, L = sqrt(8) is bplo_pdhg.  Checker of the library's non-default starts (tests/test_gpu_pdhg.py).
 * ---------------------------------------------------------------------------------------- */
BPLO_API int bplo_pdhg_opts(int M, int N, int O, const double *f, const double *alpha, int am, int an, double rho,
                            double tau0, double sigma0, int accel, int maxiter, int init, int order, double L,
                            double *x_out, double *y1_out, double *y2_out)
{
    if (M < 1 || N < 1 || O < 0 || maxiter < 0 || !(L > 0.0)) return 1;
    const size_t n = (size_t)M * N;
    double *tab = (double *)malloc(sizeof(double) * 5 * (size_t)(maxiter > 0 ? maxiter : 1));
    double *xb = (double *)malloc(n * sizeof(double));
    double *t1 = (double *)malloc(n * sizeof(double)), *t2 = (double *)malloc(n * sizeof(double));
    if (!tab || !xb || !t1 || !t2) { free(tab); free(xb); free(t1); free(t2); return 2; }
    bplo_step_table_L(maxiter, tau0, sigma0, accel, L, tab);
    for (int k = 0; k < O; ++k) {
        const double *fk = f + n * k;
        double *x = x_out + n * k;
        double *y1 = y1_out ? y1_out + n * k : t1, *y2 = y2_out ? y2_out + n * k : t2;
        for (size_t e = 0; e < n; ++e) { x[e] = init ? 0.0 : fk[e]; xb[e] = x[e]; y1[e] = 0.0; y2[e] = 0.0; }
        for (int it = 0; it < maxiter; ++it) {
            const double tau = tab[5 * it], sigma = tab[5 * it + 1], omega = tab[5 * it + 2];
            const double inv1ptau = tab[5 * it + 3], opw = tab[5 * it + 4];
            if (order) pdhg_y_pass(M, N, 0, N, alpha, am, an, xb, sigma, rho, y1, y2);
            pdhg_x_pass(M, 0, N, fk, y1, y2, tau, omega, inv1ptau, opw, x, xb);
            if (!order) pdhg_y_pass(M, N, 0, N, alpha, am, an, xb, sigma, rho, y1, y2);
        }
    }
    free(tab); free(xb); free(t1); free(t2);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The same ROF solver with each UNPINNED choice of the restatement flipped (tools/unpinned_study.py,
 * tests/test_unpinned.py).  The loop of op_denoise_pdps lives in an absent package, so these are the degrees
 * of freedom a faithful restatement has; the study bounds what they can change after the reference's 5000
 * iterations.  Plain, unfused arithmetic (a*b+c written as such): these runs are compared with tolerances.
 *   flags & 1   x0 = 0 instead of x0 = f
 *   flags & 2   dual step first (y from xbar of the previous iteration, then x, then the over-relaxation)
 *   flags & 4   projection by alpha / sqrt(n2) (IEEE sqrt and divide) instead of the Newton rsqrt
 *   flags & 8   projection written as y / max(1, |y|/alpha)
 *   flags & 16  step sizes updated BEFORE the over-relaxation uses omega (omega of the new tau)
 *   L           operator-norm estimate dividing tau0 and sigma0 (restatement: sqrt(8))
 * ---------------------------------------------------------------------------------------- */
BPLO_API int bplo_pdhg_variant(int M, int N, int O, const double *f, const double *alpha, int am, int an,
                               double tau0, double sigma0, int accel, int maxiter, int flags, double L,
                               double *x_out, double *y1_out, double *y2_out, int nthreads)
{
    if (M < 1 || N < 1 || O < 0 || maxiter < 0 || !(L > 0.0)) return 1;
    const size_t n = (size_t)M * N;
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int img = 0; img < O; ++img) {
        const double *fk = f + n * img;
        double *x = x_out + n * img;
        double *xb = (double *)malloc(n * sizeof(double));
        double *y1 = y1_out ? y1_out + n * img : (double *)malloc(n * sizeof(double));
        double *y2 = y2_out ? y2_out + n * img : (double *)malloc(n * sizeof(double));
        if (!xb || !y1 || !y2) { fail = 1; free(xb); if (!y1_out) free(y1); if (!y2_out) free(y2); continue; }
        for (size_t k = 0; k < n; ++k) {
            x[k] = (flags & 1) ? 0.0 : fk[k];
            xb[k] = x[k];
            y1[k] = 0.0;
            y2[k] = 0.0;
        }
        double tau = tau0 / L, sigma = sigma0 / L;
        const double gamma = 1.0;
        for (int it = 0; it < maxiter; ++it) {
            double omega = accel ? 1.0 / sqrt(1.0 + 2.0 * gamma * tau) : 1.0;
            for (int pass = 0; pass < 2; ++pass) {
                const int do_dual = (flags & 2) ? (pass == 0) : (pass == 1);
                if (!do_dual) {
                    if ((flags & 16) && accel) {   /* new steps first: omega belongs to the updated tau */
                        /* (tau, sigma) of THIS primal step stay; only omega changes */
                        omega = 1.0 / sqrt(1.0 + 2.0 * gamma * (tau * omega));
                    }
                    for (int j = 0; j < N; ++j)
                        for (int i = 0; i < M; ++i) {
                            size_t k = i + (size_t)M * j;
                            double y1m = (i > 0) ? y1[k - 1] : 0.0;
                            double y2m = (j > 0) ? y2[k - M] : 0.0;
                            double a1 = (i < M - 1) ? y1[k] : 0.0, a2 = (j < N - 1) ? y2[k] : 0.0;
                            double div = (y1m - a1) + (y2m - a2);
                            double xo = x[k];
                            double xn = (xo - tau * (div - fk[k])) / (1.0 + tau);
                            xb[k] = xn + omega * (xn - xo);
                            x[k] = xn;
                        }
                } else {
                    for (int j = 0; j < N; ++j)
                        for (int i = 0; i < M; ++i) {
                            size_t k = i + (size_t)M * j;
                            double d1 = (i < M - 1) ? xb[k + 1] - xb[k] : 0.0;
                            double d2 = (j < N - 1) ? xb[k + M] - xb[k] : 0.0;
                            double a = alpha_at(alpha, am, an, M, N, i, j);
                            double y1n = y1[k] + sigma * d1;
                            double y2n = y2[k] + sigma * d2;
                            double n2 = y1n * y1n + y2n * y2n;
                            if (flags & 8) {
                                double m = sqrt(n2) / a;
                                if (m > 1.0) { y1n = y1n / m; y2n = y2n / m; }
                            } else if (n2 > a * a) {
                                double v = (flags & 4) ? a / sqrt(n2) : a * rsqrt_nr(n2);
                                y1n = y1n * v;
                                y2n = y2n * v;
                            }
                            y1[k] = y1n;
                            y2[k] = y2n;
                        }
                }
            }
            if (accel) {
                omega = 1.0 / sqrt(1.0 + 2.0 * gamma * tau);
                tau = tau * omega;
                sigma = sigma / omega;
            }
        }
        free(xb);
        if (!y1_out) free(y1);
        if (!y2_out) free(y2);
    }
    return fail ? 2 : 0;
}

/* cost = 0.5*norm2^2(u - ubar) over all entries (TVLearningFunctionVec.jl:20).  Per-image sums
 * in pixel order, then summed in image order. */
BPLO_API double bplo_cost(int M, int N, int O, const double *u, const double *ubar, double *per_image)
{
    const size_t n = (size_t)M * N;
    double tot = 0.0;
    for (int k = 0; k < O; ++k) {
        double s = 0.0;
        for (size_t q = 0; q < n; ++q) {
            double d = u[n * k + q] - ubar[n * k + q];
            s += d * d;
        }
        if (per_image) per_image[k] = 0.5 * s;
        tot += 0.5 * s;
    }
    return tot;
}

/* Duality gap of image k for a feasible dual (|y_ij| <= alpha_ij):
 * gap = 0.5||u-f||^2 + sum alpha|grad u| - (0.5||f||^2 - 0.5||f - G^T y||^2) >= 0.5||u-u*||^2. */
BPLO_API void bplo_gap(int M, int N, int O, const double *u, const double *y1, const double *y2,
                       const double *f, const double *alpha, int am, int an, double *gap_out)
{
    const size_t n = (size_t)M * N;
    double *w = (double *)malloc(n * sizeof(double));
    for (int k = 0; k < O; ++k) {
        const double *uk = u + n * k, *fk = f + n * k;
        bplo_grad_fwd_T(M, N, y1 + n * k, y2 + n * k, w);
        double pr = 0.0, tv = 0.0, ff = 0.0, fw = 0.0;
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                size_t q = i + (size_t)M * j;
                double d1 = (i < M - 1) ? uk[q + 1] - uk[q] : 0.0;
                double d2 = (j < N - 1) ? uk[q + M] - uk[q] : 0.0;
                double r = uk[q] - fk[q];
                pr += r * r;
                tv += alpha_at(alpha, am, an, M, N, i, j) * sqrt(d1 * d1 + d2 * d2);
                ff += fk[q] * fk[q];
                fw += (fk[q] - w[q]) * (fk[q] - w[q]);
            }
        gap_out[k] = (0.5 * pr + tv) - (0.5 * ff - 0.5 * fw);
    }
    free(w);
}

/* ------------------------------------------------------------------------------------------
 * Adjoint gradients (A6, A7, A8, A10).
 *
 * Reference: per image the saddle system of /root/reference/src/TVLearningFunctionVec.jl:127-131
 * (scalar) / :244-248 (patch)
 *      [ I      -G^T                ] [p  ]   [u - ubar]
 *      [ Act*G + Inact*a*(Den-P)*G   Inact + e*Act ] [lam] = [0       ]
 * Eliminating lam gives the symmetric positive definite system restated here:
 *      (I + sum_k c_k b_k b_k^T + kappa * sum_{k active} (G_k^T G_k)) p = u - ubar
 * with, per inactive pixel k (|g_k| >= 1e-12, g = grad u): b_k^T p = t_k . (G p)_k,
 * t_k = (-g2, g1)/|g_k| and c_k = alpha_k/|g_k| (Den - P is the rank-one matrix t t^T/|g|), and
 * kappa = 1/e on active pixels.  e = eps() (scalar alpha) makes 1/e = 4.5e15 unrepresentable next to
 * the O(1) terms of an assembled matrix; the hard-constraint limit it approximates is reached to
 * < 1e-12 by kappa = 1e12, which is what `kappa_cap` applies (patch alpha: e = sqrt(eps()), kappa
 * = 6.7e7, used literally).  Solved by banded Cholesky (bandwidth M) + iterative refinement with
 * matrix-free residuals.  grad_pixel_k = -[inactive] g_k/|g_k| . (G p)_k  (:133-134, :250).
 *
 * gradient_reg (:137-161, :192-215): gamma = 1e8; "act" = |g| > 1/gamma;
 *      (I + S K S) q = S^-1 (ubar - u),  p = S q,  K = G^T (gamma*Inact + Act*(Den-P)) G,
 *      S = diag(sqrt(alpha)) (patch; the reference's row-scaled I + diag(alpha) K is similar to it)
 *      grad_pixel (scalar)  = (G p)_k . (Act*g/|g| + gamma*Inact*g)_k
 *      grad_pixel (patch)   = p_k * (G^T (Act*g/|g| + gamma*Inact*g))_k
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int M, N;
    double *t1, *t2; /* per pixel direction of the rank-one term (0 if none)          */
    double *c;       /* per pixel weight of the rank-one term                         */
    double *kap;     /* per pixel isotropic weight (active-set penalty / gamma*alpha) */
    double *s;       /* per node scaling S (NULL = identity)                          */
} adj_op;

/* r = A p, A = I + S G^T W G S (matrix free) */
static void adj_apply(const adj_op *op, const double *p, double *out, double *w1, double *w2)
{
    const int M = op->M, N = op->N;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            double pk = op->s ? op->s[k] * p[k] : p[k];
            double d1 = 0.0, d2 = 0.0;
            if (i < M - 1) d1 = (op->s ? op->s[k + 1] * p[k + 1] : p[k + 1]) - pk;
            if (j < N - 1) d2 = (op->s ? op->s[k + M] * p[k + M] : p[k + M]) - pk;
            double bp = op->t1[k] * d1 + op->t2[k] * d2;
            w1[k] = op->c[k] * bp * op->t1[k] + op->kap[k] * d1;
            w2[k] = op->c[k] * bp * op->t2[k] + op->kap[k] * d2;
        }
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            double a = (i < M - 1) ? w1[k] : 0.0, am = (i > 0) ? w1[k - 1] : 0.0;
            double b = (j < N - 1) ? w2[k] : 0.0, bm = (j > 0) ? w2[k - M] : 0.0;
            double gt = (am - a) + (bm - b);
            out[k] = p[k] + (op->s ? op->s[k] * gt : gt);
        }
}

/* Lower band storage: A[(k+d), k] at band[k*(bw+1) + d], d = 0..bw, bw = M. */
static void adj_assemble(const adj_op *op, double *band)
{
    const int M = op->M, N = op->N, bw = M;
    const size_t n = (size_t)M * N, ld = (size_t)bw + 1;
    memset(band, 0, n * ld * sizeof(double));
    for (size_t k = 0; k < n; ++k) band[k * ld] = 1.0;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            size_t a = i + (size_t)M * j;
            const int hb = (i < M - 1), hc = (j < N - 1);
            /* element vector over nodes (a, b = a+1, c = a+M): coefficients of t.(Gp) */
            double e1 = hb ? op->t1[a] : 0.0, e2 = hc ? op->t2[a] : 0.0;
            double va = -(e1 + e2), vb = e1, vc = e2;
            double sa = op->s ? op->s[a] : 1.0;
            double sb = (op->s && hb) ? op->s[a + 1] : 1.0;
            double sc = (op->s && hc) ? op->s[a + M] : 1.0;
            double c = op->c[a], kp = op->kap[a];
            va *= sa; vb *= sb; vc *= sc;
            band[a * ld] += c * va * va;
            if (hb) {
                band[(a + 1) * ld] += c * vb * vb;
                band[a * ld + 1] += c * va * vb;
            }
            if (hc) {
                band[(a + M) * ld] += c * vc * vc;
                band[a * ld + M] += c * va * vc;
            }
            if (hb && hc) band[(a + 1) * ld + (M - 1)] += c * vb * vc;
            if (kp != 0.0) {
                if (hb) {
                    band[a * ld] += kp * sa * sa;
                    band[(a + 1) * ld] += kp * sb * sb;
                    band[a * ld + 1] -= kp * sa * sb;
                }
                if (hc) {
                    band[a * ld] += kp * sa * sa;
                    band[(a + M) * ld] += kp * sc * sc;
                    band[a * ld + M] -= kp * sa * sc;
                }
            }
        }
}

static int band_cholesky(size_t n, int bw, double *band)
{
    const size_t ld = (size_t)bw + 1;
    for (size_t k = 0; k < n; ++k) {
        double *ck = band + k * ld;
        if (!(ck[0] > 0.0)) return 1;
        double d = sqrt(ck[0]);
        ck[0] = d;
        int lim = (int)((n - 1 - k < (size_t)bw) ? (n - 1 - k) : (size_t)bw);
        double inv = 1.0 / d;
        for (int r = 1; r <= lim; ++r) ck[r] *= inv;
        for (int c = 1; c <= lim; ++c) {
            double lc = ck[c];
            if (lc == 0.0) continue;
            double *cc = band + (k + c) * ld;
            for (int r = c; r <= lim; ++r) cc[r - c] -= ck[r] * lc;
        }
    }
    return 0;
}

static void band_solve(size_t n, int bw, const double *band, double *x)
{
    const size_t ld = (size_t)bw + 1;
    for (size_t k = 0; k < n; ++k) {
        const double *ck = band + k * ld;
        double v = x[k] / ck[0];
        x[k] = v;
        int lim = (int)((n - 1 - k < (size_t)bw) ? (n - 1 - k) : (size_t)bw);
        for (int r = 1; r <= lim; ++r) x[k + r] -= ck[r] * v;
    }
    for (size_t kk = n; kk-- > 0;) {
        const double *ck = band + kk * ld;
        int lim = (int)((n - 1 - kk < (size_t)bw) ? (n - 1 - kk) : (size_t)bw);
        double v = x[kk];
        for (int r = 1; r <= lim; ++r) v -= ck[r] * x[kk + r];
        x[kk] = v / ck[0];
    }
}

/* Solve A p = rhs: banded Cholesky + `nref` refinement sweeps.  Returns 0 on success. */
static int adj_solve(const adj_op *op, const double *rhs, double *p, int nref, double *res_norm)
{
    const int M = op->M, N = op->N;
    const size_t n = (size_t)M * N;
    double *band = (double *)malloc(n * ((size_t)M + 1) * sizeof(double));
    double *r = (double *)malloc(n * sizeof(double));
    double *w1 = (double *)malloc(n * sizeof(double));
    double *w2 = (double *)malloc(n * sizeof(double));
    double *ap = (double *)malloc(n * sizeof(double));
    int rc = 2;
    if (band && r && w1 && w2 && ap) {
        adj_assemble(op, band);
        rc = band_cholesky(n, M, band);
        if (rc == 0) {
            memcpy(p, rhs, n * sizeof(double));
            band_solve(n, M, band, p);
            for (int it = 0; it < nref; ++it) {
                adj_apply(op, p, ap, w1, w2);
                for (size_t k = 0; k < n; ++k) r[k] = rhs[k] - ap[k];
                band_solve(n, M, band, r);
                for (size_t k = 0; k < n; ++k) p[k] += r[k];
            }
            if (res_norm) {
                adj_apply(op, p, ap, w1, w2);
                double s = 0.0, t = 0.0;
                for (size_t k = 0; k < n; ++k) {
                    s += (rhs[k] - ap[k]) * (rhs[k] - ap[k]);
                    t += rhs[k] * rhs[k];
                }
                *res_norm = sqrt(s) / (t > 0 ? sqrt(t) : 1.0);
            }
        }
    }
    free(band); free(r); free(w1); free(w2); free(ap);
    return rc;
}

#define BPLO_ACT_TOL 1e-12   /* TVLearningFunctionVec.jl:109,231 */
#define BPLO_GAMMA 1e8       /* TVLearningFunctionVec.jl:142,197 */

/* One image.  gpix (M*N) receives the per-pixel gradient contributions, p_out (nullable) the
 * adjoint state.  patch != 0 selects the array-alpha formulas (:219-252 / :192-213). */
BPLO_API int bplo_gradient_image(int M, int N, const double *u, const double *ubar,
                                 const double *amap, int patch, int reg, double kappa_cap, int nref,
                                 double *gpix, double *p_out, double *res_norm)
{
    const size_t n = (size_t)M * N;
    double *buf = (double *)calloc(9 * n, sizeof(double));
    if (!buf) return 2;
    double *t1 = buf, *t2 = buf + n, *c = buf + 2 * n, *kap = buf + 3 * n, *s = buf + 4 * n;
    double *rhs = buf + 5 * n, *p = buf + 6 * n, *h1 = buf + 7 * n, *h2 = buf + 8 * n;
    adj_op op = {M, N, t1, t2, c, kap, NULL};
    const double eps = DBL_EPSILON;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            size_t k = i + (size_t)M * j;
            double g1 = (i < M - 1) ? u[k + 1] - u[k] : 0.0;
            double g2 = (j < N - 1) ? u[k + M] - u[k] : 0.0;
            double ng = sqrt(g1 * g1 + g2 * g2);
            double a = amap[k];
            if (!reg) {
                if (ng < BPLO_ACT_TOL) { /* active: G p = 0 enforced with weight kappa */
                    double kp = 1.0 / (patch ? sqrt(eps) : eps);
                    kap[k] = (kp > kappa_cap) ? kappa_cap : kp;
                } else {
                    t1[k] = -g2 / ng; t2[k] = g1 / ng;
                    c[k] = a / ng;
                    h1[k] = g1 / ng; h2[k] = g2 / ng;
                }
                rhs[k] = u[k] - ubar[k];
            } else {
                if (ng > 1.0 / BPLO_GAMMA) { /* "act" of gradient_reg */
                    t1[k] = -g2 / ng; t2[k] = g1 / ng;
                    c[k] = patch ? 1.0 / ng : a / ng;
                    h1[k] = g1 / ng; h2[k] = g2 / ng;
                } else {
                    kap[k] = patch ? BPLO_GAMMA : a * BPLO_GAMMA;
                    h1[k] = BPLO_GAMMA * g1; h2[k] = BPLO_GAMMA * g2;
                }
                rhs[k] = ubar[k] - u[k];
            }
        }
    if (reg && patch) { /* symmetrised row scaling: S = diag(sqrt(alpha_node)) */
        for (size_t k = 0; k < n; ++k) {
            s[k] = sqrt(amap[k]);
            rhs[k] = rhs[k] / s[k];
        }
        op.s = s;
    }
    int rc = adj_solve(&op, rhs, p, nref, res_norm);
    if (rc == 0) {
        if (op.s) for (size_t k = 0; k < n; ++k) p[k] *= s[k];
        if (!(reg && patch)) {
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < M; ++i) {
                    size_t k = i + (size_t)M * j;
                    double d1 = (i < M - 1) ? p[k + 1] - p[k] : 0.0;
                    double d2 = (j < N - 1) ? p[k + M] - p[k] : 0.0;
                    double v = d1 * h1[k] + d2 * h2[k];
                    gpix[k] = reg ? v : -v;
                }
        } else {
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < M; ++i) {
                    size_t k = i + (size_t)M * j;
                    double a = (i < M - 1) ? h1[k] : 0.0, am_ = (i > 0) ? h1[k - 1] : 0.0;
                    double b = (j < N - 1) ? h2[k] : 0.0, bm = (j > 0) ? h2[k - M] : 0.0;
                    gpix[k] = p[k] * ((am_ - a) + (bm - b));
                }
        }
        if (p_out) memcpy(p_out, p, n * sizeof(double));
    }
    free(buf);
    return rc;
}

/* Batch wrappers (TVLearningFunctionVec.jl:72-96, :163-190): grad (am*an doubles) summed over the
 * O images; scalar alpha -> sum over pixels, array alpha -> calc_adjoint patch sums. */
BPLO_API int bplo_gradient(int M, int N, int O, const double *u, const double *ubar,
                           const double *alpha, int am, int an, int reg, double kappa_cap, int nref,
                           double *grad_out, double *per_image /* O*am*an or NULL */)
{
    const size_t n = (size_t)M * N;
    const int patch = !(am == 1 && an == 1);
    double *amap = (double *)malloc(n * sizeof(double));
    double *gpix = (double *)malloc(n * sizeof(double));
    double *gi = (double *)malloc(sizeof(double) * am * an);
    if (!amap || !gpix || !gi) { free(amap); free(gpix); free(gi); return 2; }
    bplo_patch_upsample(alpha, am, an, M, N, amap);
    for (int q = 0; q < am * an; ++q) grad_out[q] = 0.0;
    int rc = 0;
    for (int k = 0; k < O && rc == 0; ++k) {
        rc = bplo_gradient_image(M, N, u + n * k, ubar + n * k, amap, patch, reg, kappa_cap, nref,
                                 gpix, NULL, NULL);
        if (rc) break;
        if (!patch) {
            double sgrad = 0.0;
            for (size_t q = 0; q < n; ++q) sgrad += gpix[q];
            gi[0] = sgrad;
        } else {
            bplo_patch_adjoint(gpix, M, N, am, an, gi);
        }
        for (int q = 0; q < am * an; ++q) {
            grad_out[q] += gi[q];
            if (per_image) per_image[(size_t)k * am * an + q] = gi[q];
        }
    }
    free(amap); free(gpix); free(gi);
    return rc;
}

BPLO_API int bplo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
