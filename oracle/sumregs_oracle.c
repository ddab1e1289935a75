/*
 * sumregs_oracle.c -- CPU restatement of the sum-of-regularisers learning function.  TEST INFRASTRUCTURE ONLY
 * (same rules as bpltv_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it).
 *
 * What it restates (/root/reference/src/SumRegsLearningFunction.jl):
 *   :8-36    sumregs_learning_function(x, data, D; Dt = 1e-3) -> (u, cost, grad), x a 3-vector or m x n x 3
 *   :38-85   sumregs_denoise -> sumregs_denoise_pdps: min_u 0.5||u-f||^2 + sum_k alpha_k ||G_k u||_{2,1} with
 *            G_1 = FwdGradientOp, G_2 = BwdGradientOp, G_3 = CenteredGradientOp, three duals; tau0 = 5,
 *            sigma0 = 0.99/5, accel, 5000 iterations
 *   :264-327 sumregs_gradient (vector x), :330-407 (patch x), :112-167 / :195-262 sumregs_gradient_reg
 *
 * PARITY UNPINNED BY THE REFERENCE.  `sumregs_denoise_pdps`, `BwdGradientOp`, `CenteredGradientOp`, `matrix(op,n)`
 * live in the absent package VariationalImaging; the reference holds no fixture for this model.  Choices made
 * here (and reproduced bit for bit by the HIP kernels), all within the Neumann family of the forward operator:
 *   G_1 (fwd):  (u[i+1]-u[i], u[j+1]-u[j]),        0 at the last row / column
 *   G_2 (bwd):  (u[i]-u[i-1], u[j]-u[j-1]),        0 at the first row / column
 *   G_3 (ctr):  0.5*(u[min(i+1,M-1)] - u[max(i-1,0)]) and likewise in j  (mirrored border: one-sided half
 *               differences at the first / last row and column)
 *   PDHG:       accelerated Chambolle-Pock exactly as bpltv_oracle.c ("spec v2": explicit fma, Newton rsqrt
 *               projection, x0 = f, y0 = 0, primal step first), K = [G_1; G_2; G_3],
 *               L = sqrt(8 + 8 + 2) = sqrt(18) >= ||K||
 *   adjoint:    the reference's saddle systems (:318-324, :388-394) reduced to the SPD system
 *                   (I + sum_k G_k^T W_k G_k) p = u - ubar,
 *               W_k per element = alpha_k/|g| t t^T on the inactive set (|g| >= 1e-12; Den - prodKuKu is that
 *               rank-one matrix) and (1/eps()) I on the active set (capped at kappa_cap, as in bpltv_oracle.c);
 *               bandwidth 2M (the centred stencil couples columns j and j+2); banded Cholesky + refinement.
 *               grad_k = -sum_e (G_k p)_e . (Inact Den G_k u)_e  (vector x, :326);
 *               patch x: per pixel -p_q (G_k^T Inact Den G_k u)_q, then calc_adjoint (:397-405).
 *   gradient_reg: gamma = 1e3 (vector x, :117) / 1e8 (patch x, :200); act = |g| > 1/gamma;
 *               (I + sum_k X_k G_k^T (gamma Inact + Act (Den - P)) G_k) p = ubar - u  with X_k = x_k (vector) or
 *               the ROW scaling diag(x_k map) (patch, :250) -- with three different maps that matrix is not
 *               symmetric: banded LU without pivoting (bandwidth 2M both sides).
 *               grad_k = p^T G_k^T (Act Den G_k u + gamma Inact G_k u) (:165); patch: p_q (G_k^T ...)_q (:251-259).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BPLO_API __attribute__((visibility("default")))

/* from bpltv_oracle.c */
void bplo_step_table_L(int maxiter, double tau0, double sigma0, int accel, double L, double *tab);
double bplo_rsqrt_nr(double n2);

static inline double rsqrt_nr(double n2)
{
    union { double d; uint64_t u; } c;
    c.d = n2;
    c.u = 0x5FE6EB50C7B537A9ull - (c.u >> 1);
    double r = c.d;
    const double h = 0.5 * n2;
    for (int k = 0; k < 4; ++k) {
        double t = r * r;
        double w = fma(-h, t, 1.5);
        r = r * w;
    }
    return r;
}

/* parameter slice k (0..2) at pixel (i, j): alpha holds the three am x an slices one after the other */
static inline double sr_alpha_at(const double *alpha, int k, int am, int an, int M, int N, int i, int j)
{
    const double *a = alpha + (size_t)k * am * an;
    if (am == 1 && an == 1) return a[0];
    if (am == M && an == N) return a[i + (size_t)M * j];
    return a[(int)(((long)i * am) / M) + (size_t)am * (int)(((long)j * an) / N)];
}

/* ---- the three operators on one image (component 1 along i, component 2 along j) -------------------- */
BPLO_API void bplo_sr_grad(int k, int M, int N, const double *u, double *d1, double *d2)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            const size_t q = i + (size_t)M * j;
            const double uc = u[q];
            const double up = u[(i < M - 1) ? q + 1 : q], um = u[(i > 0) ? q - 1 : q];
            const double vp = u[(j < N - 1) ? q + M : q], vm = u[(j > 0) ? q - M : q];
            if (k == 0) { d1[q] = up - uc; d2[q] = vp - uc; }
            else if (k == 1) { d1[q] = uc - um; d2[q] = uc - vm; }
            else { d1[q] = 0.5 * (up - um); d2[q] = 0.5 * (vp - vm); }
        }
}

/* (G_k^T y)(q) in gather form -- the summation order the HIP kernel reproduces */
static inline double sr_gradT_at(int k, int M, int N, const double *y1, const double *y2, int i, int j)
{
    const size_t q = i + (size_t)M * j;
    if (k == 0) {
        const double a = (i > 0) ? y1[q - 1] : 0.0, b = (i < M - 1) ? y1[q] : 0.0;
        const double c = (j > 0) ? y2[q - M] : 0.0, d = (j < N - 1) ? y2[q] : 0.0;
        return (a - b) + (c - d);
    }
    if (k == 1) {
        const double a = (i > 0) ? y1[q] : 0.0, b = (i < M - 1) ? y1[q + 1] : 0.0;
        const double c = (j > 0) ? y2[q] : 0.0, d = (j < N - 1) ? y2[q + M] : 0.0;
        return (a - b) + (c - d);
    }
    const double a = (i > 0) ? y1[q - 1] : -y1[q], b = (i < M - 1) ? y1[q + 1] : -y1[q];
    const double c = (j > 0) ? y2[q - M] : -y2[q], d = (j < N - 1) ? y2[q + M] : -y2[q];
    return 0.5 * (a - b) + 0.5 * (c - d);
}

BPLO_API void bplo_sr_gradT(int k, int M, int N, const double *y1, const double *y2, double *out)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) out[i + (size_t)M * j] = sr_gradT_at(k, M, N, y1, y2, i, j);
}

#define SR_L sqrt(18.0) /* ||G_1||^2 + ||G_2||^2 + ||G_3||^2 <= 8 + 8 + 2 */

/* ---- PDHG, three duals ---------------------------------------------------------------------------- */
static void sr_pdhg_image(int M, int N, const double *f, const double *alpha, int am, int an, const double *tab,
                          int maxiter, double rho, double *x, double *y /* 6 planes */, double *xb)
{
    const size_t n = (size_t)M * N;
    memcpy(x, f, n * sizeof(double));
    memset(y, 0, 6 * n * sizeof(double));
    for (int it = 0; it < maxiter; ++it) {
        const double tau = tab[5 * it], sigma = tab[5 * it + 1], omega = tab[5 * it + 2];
        const double inv1ptau = tab[5 * it + 3], opw = tab[5 * it + 4];
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                const size_t q = i + (size_t)M * j;
                const double t0 = sr_gradT_at(0, M, N, y, y + n, i, j);
                const double t1 = sr_gradT_at(1, M, N, y + 2 * n, y + 3 * n, i, j);
                const double t2 = sr_gradT_at(2, M, N, y + 4 * n, y + 5 * n, i, j);
                const double div = (t0 + t1) + t2;
                const double t = div - f[q];
                const double xo = x[q];
                const double xn = fma(-tau, t, xo) * inv1ptau;
                xb[q] = fma(-omega, xo, opw * xn);
                x[q] = xn;
            }
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                const size_t q = i + (size_t)M * j;
                const double bc = xb[q];
                const double bp = xb[(i < M - 1) ? q + 1 : q], bm = xb[(i > 0) ? q - 1 : q];
                const double cp = xb[(j < N - 1) ? q + M : q], cm = xb[(j > 0) ? q - M : q];
                for (int k = 0; k < 3; ++k) {
                    double d1, d2;
                    if (k == 0) { d1 = bp - bc; d2 = cp - bc; }
                    else if (k == 1) { d1 = bc - bm; d2 = bc - cm; }
                    else { d1 = 0.5 * (bp - bm); d2 = 0.5 * (cp - cm); }
                    const double a = sr_alpha_at(alpha, k, am, an, M, N, i, j);
                    double *y1 = y + (size_t)(2 * k) * n, *y2 = y + (size_t)(2 * k + 1) * n;
                    double y1n = fma(sigma, d1, y1[q]);
                    double y2n = fma(sigma, d2, y2[q]);
                    if (rho != 0.0) {
                        const double den = 1.0 + sigma * rho / a;
                        y1n = y1n / den;
                        y2n = y2n / den;
                    }
                    const double n2 = fma(y2n, y2n, y1n * y1n);
                    if (n2 > a * a) {
                        const double v = a * rsqrt_nr(n2);
                        y1n = y1n * v;
                        y2n = y2n * v;
                    }
                    y1[q] = y1n;
                    y2[q] = y2n;
                }
            }
    }
}

/* alpha: 3 * am * an doubles.  y_out (nullable): 6 planes per image, [O][6][N*M]. */
BPLO_API int bplo_sumregs_pdhg(int M, int N, int O, const double *f, const double *alpha, int am, int an, double rho,
                               double tau0, double sigma0, int accel, int maxiter, double *x_out, double *y_out,
                               int nthreads)
{
    if (M < 1 || N < 1 || O < 0 || maxiter < 0) return 1;
    const size_t n = (size_t)M * N;
    double *tab = (double *)malloc(sizeof(double) * 5 * (size_t)(maxiter > 0 ? maxiter : 1));
    if (!tab) return 2;
    bplo_step_table_L(maxiter, tau0, sigma0, accel, SR_L, tab);
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int k = 0; k < O; ++k) {
        double *xb = (double *)malloc(n * sizeof(double));
        double *y = y_out ? y_out + 6 * n * k : (double *)malloc(6 * n * sizeof(double));
        if (!xb || !y) fail = 1;
        else sr_pdhg_image(M, N, f + n * k, alpha, am, an, tab, maxiter, rho, x_out + n * k, y, xb);
        free(xb);
        if (!y_out) free(y);
    }
    free(tab);
    return fail ? 2 : 0;
}

/* Duality gap per image: 0.5||u-f||^2 + sum_k sum alpha_k |G_k u| - (0.5||f||^2 - 0.5||f - K^T y||^2). */
BPLO_API void bplo_sumregs_gap(int M, int N, int O, const double *u, const double *y, const double *f,
                               const double *alpha, int am, int an, double *gap_out)
{
    const size_t n = (size_t)M * N;
    double *d1 = (double *)malloc(n * sizeof(double)), *d2 = (double *)malloc(n * sizeof(double));
    for (int img = 0; img < O; ++img) {
        const double *uk = u + n * img, *fk = f + n * img, *yk = y + 6 * n * img;
        double pr = 0.0, tv = 0.0, ff = 0.0, fw = 0.0;
        for (int k = 0; k < 3; ++k) {
            bplo_sr_grad(k, M, N, uk, d1, d2);
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < M; ++i) {
                    const size_t q = i + (size_t)M * j;
                    tv += sr_alpha_at(alpha, k, am, an, M, N, i, j) * sqrt(d1[q] * d1[q] + d2[q] * d2[q]);
                }
        }
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                const size_t q = i + (size_t)M * j;
                const double w = (sr_gradT_at(0, M, N, yk, yk + n, i, j) + sr_gradT_at(1, M, N, yk + 2 * n, yk + 3 * n, i, j)) +
                                 sr_gradT_at(2, M, N, yk + 4 * n, yk + 5 * n, i, j);
                const double r = uk[q] - fk[q];
                pr += r * r;
                ff += fk[q] * fk[q];
                fw += (fk[q] - w) * (fk[q] - w);
            }
        gap_out[img] = (0.5 * pr + tv) - (0.5 * ff - 0.5 * fw);
    }
    free(d1); free(d2);
}

/* ---- adjoint gradients ------------------------------------------------------------------------------ */
/* Element e of operator k couples, per component, a "plus" node and a "minus" node with scale s:
 * (G_k p)_e,c = s * (p[plus] - p[minus]); s = 0 when the component vanishes at e. */
typedef struct { size_t pl, mi; double s; } sr_comp;

static inline void sr_stencil(int k, int M, int N, int i, int j, sr_comp *c1, sr_comp *c2)
{
    const size_t q = i + (size_t)M * j;
    if (k == 0) {
        c1->pl = (i < M - 1) ? q + 1 : q; c1->mi = q; c1->s = (i < M - 1) ? 1.0 : 0.0;
        c2->pl = (j < N - 1) ? q + M : q; c2->mi = q; c2->s = (j < N - 1) ? 1.0 : 0.0;
    } else if (k == 1) {
        c1->pl = q; c1->mi = (i > 0) ? q - 1 : q; c1->s = (i > 0) ? 1.0 : 0.0;
        c2->pl = q; c2->mi = (j > 0) ? q - M : q; c2->s = (j > 0) ? 1.0 : 0.0;
    } else {
        c1->pl = (i < M - 1) ? q + 1 : q; c1->mi = (i > 0) ? q - 1 : q; c1->s = (c1->pl != c1->mi) ? 0.5 : 0.0;
        c2->pl = (j < N - 1) ? q + M : q; c2->mi = (j > 0) ? q - M : q; c2->s = (c2->pl != c2->mi) ? 0.5 : 0.0;
    }
}

typedef struct {
    int M, N;
    /* per operator k and element: W = c t t^T + kap I, scaled by the parameter (wt) */
    double *t1[3], *t2[3], *c[3], *kap[3];
    double *rowscale[3]; /* reg + patch: diag(x_k map) row scaling of term k (NULL otherwise: folded into c, kap) */
} sr_op;

/* out = A p */
static void sr_apply(const sr_op *op, const double *p, double *out, double *w1, double *w2, double *tmp)
{
    const int M = op->M, N = op->N;
    const size_t n = (size_t)M * N;
    memcpy(out, p, n * sizeof(double));
    for (int k = 0; k < 3; ++k) {
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                const size_t e = i + (size_t)M * j;
                sr_comp c1, c2;
                sr_stencil(k, M, N, i, j, &c1, &c2);
                const double d1 = c1.s * (p[c1.pl] - p[c1.mi]), d2 = c2.s * (p[c2.pl] - p[c2.mi]);
                const double bp = op->t1[k][e] * d1 + op->t2[k][e] * d2;
                w1[e] = op->c[k][e] * bp * op->t1[k][e] + op->kap[k][e] * d1;
                w2[e] = op->c[k][e] * bp * op->t2[k][e] + op->kap[k][e] * d2;
            }
        bplo_sr_gradT(k, M, N, w1, w2, tmp);
        /* NOTE bplo_sr_gradT implements G_k^T exactly only because w vanishes where the component does (s = 0
         * => d = 0 => w = 0 for the rank-one part; kap * d = 0 too). */
        if (op->rowscale[k])
            for (size_t q = 0; q < n; ++q) out[q] += op->rowscale[k][q] * tmp[q];
        else
            for (size_t q = 0; q < n; ++q) out[q] += tmp[q];
    }
}

/* Full band storage (lower and upper, bandwidth bw each side): A[r][c] at band[c*ld + (r - c + bw)], ld = 2bw+1. */
static void sr_assemble(const sr_op *op, int bw, double *band)
{
    const int M = op->M, N = op->N;
    const size_t n = (size_t)M * N, ld = 2 * (size_t)bw + 1;
    memset(band, 0, n * ld * sizeof(double));
    for (size_t q = 0; q < n; ++q) band[q * ld + bw] = 1.0;
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                const size_t e = i + (size_t)M * j;
                sr_comp cc[2];
                sr_stencil(k, M, N, i, j, &cc[0], &cc[1]);
                /* element matrix: sum over (a, b) of V[a]^T W V[b], V rows: comp 1, comp 2 over the nodes */
                size_t nodes[4] = {cc[0].pl, cc[0].mi, cc[1].pl, cc[1].mi};
                double v1[4] = {cc[0].s, -cc[0].s, 0.0, 0.0}, v2[4] = {0.0, 0.0, cc[1].s, -cc[1].s};
                const double t1 = op->t1[k][e], t2 = op->t2[k][e], c = op->c[k][e], kp = op->kap[k][e];
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        const double ta = t1 * v1[a] + t2 * v2[a], tb = t1 * v1[b] + t2 * v2[b];
                        double val = c * ta * tb + kp * (v1[a] * v1[b] + v2[a] * v2[b]);
                        if (val == 0.0) continue;
                        const size_t r = nodes[a], col = nodes[b];
                        if (op->rowscale[k]) val *= op->rowscale[k][r];
                        band[col * ld + (size_t)((long)r - (long)col + bw)] += val;
                    }
            }
}

/* LU without pivoting of a full band, in place (L unit lower below the diagonal, U on and above). */
static int band_lu(size_t n, int bw, double *band)
{
    const size_t ld = 2 * (size_t)bw + 1;
#define AB(r, c) band[(size_t)(c) * ld + (size_t)((long)(r) - (long)(c) + bw)]
    for (size_t k = 0; k < n; ++k) {
        const double piv = AB(k, k);
        if (!(fabs(piv) > 0.0) || piv != piv) return 1;
        const size_t rmax = (k + bw < n - 1) ? k + bw : n - 1;
        for (size_t r = k + 1; r <= rmax; ++r) AB(r, k) /= piv;
        for (size_t c = k + 1; c <= rmax; ++c) {
            const double ukc = AB(k, c);
            if (ukc == 0.0) continue;
            for (size_t r = k + 1; r <= rmax; ++r) AB(r, c) -= AB(r, k) * ukc;
        }
    }
    return 0;
}

static void band_lu_solve(size_t n, int bw, const double *band, double *x)
{
    const size_t ld = 2 * (size_t)bw + 1;
    for (size_t k = 0; k < n; ++k) {
        const size_t rmax = (k + bw < n - 1) ? k + bw : n - 1;
        const double v = x[k];
        for (size_t r = k + 1; r <= rmax; ++r) x[r] -= AB(r, k) * v;
    }
    for (size_t kk = n; kk-- > 0;) {
        const size_t cmax = (kk + bw < n - 1) ? kk + bw : n - 1;
        double v = x[kk];
        for (size_t c = kk + 1; c <= cmax; ++c) v -= AB(kk, c) * x[c];
        x[kk] = v / AB(kk, kk);
    }
#undef AB
}

static int sr_solve(const sr_op *op, const double *rhs, double *p, int nref, double *res_norm)
{
    const int M = op->M, N = op->N, bw = (2 * M < M * N - 1) ? 2 * M : M * N - 1;
    const size_t n = (size_t)M * N;
    double *band = (double *)malloc(n * (2 * (size_t)bw + 1) * sizeof(double));
    double *r = (double *)malloc(n * sizeof(double)), *w1 = (double *)malloc(n * sizeof(double));
    double *w2 = (double *)malloc(n * sizeof(double)), *ap = (double *)malloc(n * sizeof(double));
    double *tmp = (double *)malloc(n * sizeof(double));
    int rc = 2;
    if (band && r && w1 && w2 && ap && tmp) {
        sr_assemble(op, bw, band);
        rc = band_lu(n, bw, band);   /* for the symmetric cases this is the LDL^T of the SPD matrix: no pivoting needed */
        if (rc == 0) {
            memcpy(p, rhs, n * sizeof(double));
            band_lu_solve(n, bw, band, p);
            for (int it = 0; it < nref; ++it) {
                sr_apply(op, p, ap, w1, w2, tmp);
                for (size_t k = 0; k < n; ++k) r[k] = rhs[k] - ap[k];
                band_lu_solve(n, bw, band, r);
                for (size_t k = 0; k < n; ++k) p[k] += r[k];
            }
            if (res_norm) {
                sr_apply(op, p, ap, w1, w2, tmp);
                double s = 0.0, t = 0.0;
                for (size_t k = 0; k < n; ++k) {
                    s += (rhs[k] - ap[k]) * (rhs[k] - ap[k]);
                    t += rhs[k] * rhs[k];
                }
                *res_norm = sqrt(s) / (t > 0 ? sqrt(t) : 1.0);
            }
        }
    }
    free(band); free(r); free(w1); free(w2); free(ap); free(tmp);
    return rc;
}

#define SR_ACT_TOL 1e-12 /* SumRegsLearningFunction.jl:274,290,306 */

/* One image.  amaps: the three upsampled parameter maps (3 planes).  patch: array-x formulas.
 * gpix: 3 planes of per-pixel gradient contributions (vector x: summed over the pixels by the caller). */
BPLO_API int bplo_sumregs_gradient_image(int M, int N, const double *u, const double *ubar, const double *amaps,
                                         int patch, int reg, double kappa_cap, int nref, double *gpix, double *p_out,
                                         double *res_norm)
{
    const size_t n = (size_t)M * N;
    double *buf = (double *)calloc((3 * 6 + 2 + 3) * n, sizeof(double));
    if (!buf) return 2;
    sr_op op;
    op.M = M; op.N = N;
    double *h1[3], *h2[3];
    for (int k = 0; k < 3; ++k) {
        op.t1[k] = buf + (6 * k + 0) * n; op.t2[k] = buf + (6 * k + 1) * n;
        op.c[k] = buf + (6 * k + 2) * n;  op.kap[k] = buf + (6 * k + 3) * n;
        h1[k] = buf + (6 * k + 4) * n;    h2[k] = buf + (6 * k + 5) * n;
        op.rowscale[k] = NULL;
    }
    double *rhs = buf + 18 * n, *p = buf + 19 * n, *tmp = buf + 20 * n, *d1 = buf + 21 * n, *d2 = buf + 22 * n;
    const double gamma = patch ? 1e8 : 1e3;                       /* :200 / :117 */
    double kact = 1.0 / DBL_EPSILON;                              /* eps() in both variants, :319, :389 */
    if (kact > kappa_cap) kact = kappa_cap;
    for (int k = 0; k < 3; ++k) {
        bplo_sr_grad(k, M, N, u, d1, d2);
        const double *am_k = amaps + (size_t)k * n;
        for (size_t e = 0; e < n; ++e) {
            const double g1 = d1[e], g2 = d2[e], ng = sqrt(g1 * g1 + g2 * g2), a = am_k[e];
            const int rowsc = reg && patch;   /* the parameter multiplies ROWS of term k: kept out of c, kap */
            if (!reg) {
                if (ng < SR_ACT_TOL) {
                    op.kap[k][e] = kact;
                } else {
                    op.t1[k][e] = -g2 / ng; op.t2[k][e] = g1 / ng;
                    op.c[k][e] = a / ng;
                    h1[k][e] = g1 / ng; h2[k][e] = g2 / ng;
                }
            } else {
                if (ng > 1.0 / gamma) {
                    op.t1[k][e] = -g2 / ng; op.t2[k][e] = g1 / ng;
                    op.c[k][e] = rowsc ? 1.0 / ng : a / ng;
                    h1[k][e] = g1 / ng; h2[k][e] = g2 / ng;
                } else {
                    op.kap[k][e] = rowsc ? gamma : a * gamma;
                    h1[k][e] = gamma * g1; h2[k][e] = gamma * g2;
                }
            }
        }
        if (reg && patch) op.rowscale[k] = (double *)am_k;
    }
    for (size_t q = 0; q < n; ++q) rhs[q] = reg ? ubar[q] - u[q] : u[q] - ubar[q];
    int rc = sr_solve(&op, rhs, p, nref, res_norm);
    if (rc == 0) {
        for (int k = 0; k < 3; ++k) {
            double *gk = gpix + (size_t)k * n;
            if (!patch) {   /* per element: (G_k p)_e . h_e */
                bplo_sr_grad(k, M, N, p, d1, d2);
                for (size_t e = 0; e < n; ++e) {
                    const double v = d1[e] * h1[k][e] + d2[e] * h2[k][e];
                    gk[e] = reg ? v : -v;
                }
            } else {        /* per node: p_q (G_k^T h)_q */
                bplo_sr_gradT(k, M, N, h1[k], h2[k], tmp);
                for (size_t q = 0; q < n; ++q) gk[q] = reg ? p[q] * tmp[q] : -(p[q] * tmp[q]);
            }
        }
        if (p_out) memcpy(p_out, p, n * sizeof(double));
    }
    free(buf);
    return rc;
}

/* Batch wrapper (:87-110, :169-193): grad (3*am*an doubles, slice-major) summed over the O images. */
BPLO_API int bplo_sumregs_gradient(int M, int N, int O, const double *u, const double *ubar, const double *alpha,
                                   int am, int an, int reg, double kappa_cap, int nref, double *grad_out,
                                   double *per_image /* O*3*am*an or NULL */)
{
    const size_t n = (size_t)M * N, P = (size_t)am * an;
    const int patch = !(am == 1 && an == 1);
    double *amaps = (double *)malloc(3 * n * sizeof(double)), *gpix = (double *)malloc(3 * n * sizeof(double));
    double *gi = (double *)malloc(3 * P * sizeof(double));
    if (!amaps || !gpix || !gi) { free(amaps); free(gpix); free(gi); return 2; }
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) amaps[(size_t)k * n + i + (size_t)M * j] = sr_alpha_at(alpha, k, am, an, M, N, i, j);
    for (size_t q = 0; q < 3 * P; ++q) grad_out[q] = 0.0;
    int rc = 0;
    for (int img = 0; img < O && rc == 0; ++img) {
        rc = bplo_sumregs_gradient_image(M, N, u + n * img, ubar + n * img, amaps, patch, reg, kappa_cap, nref, gpix, NULL, NULL);
        if (rc) break;
        for (int k = 0; k < 3; ++k) {
            const double *gk = gpix + (size_t)k * n;
            double *go = gi + (size_t)k * P;
            for (size_t q = 0; q < P; ++q) go[q] = 0.0;
            if (!patch) {
                double s = 0.0;
                for (size_t q = 0; q < n; ++q) s += gk[q];
                go[0] = s;
            } else {
                for (int j = 0; j < N; ++j)
                    for (int i = 0; i < M; ++i)
                        go[(int)(((long)i * am) / M) + (size_t)am * (int)(((long)j * an) / N)] += gk[i + (size_t)M * j];
            }
        }
        for (size_t q = 0; q < 3 * P; ++q) {
            grad_out[q] += gi[q];
            if (per_image) per_image[(size_t)img * 3 * P + q] = gi[q];
        }
    }
    free(amaps); free(gpix); free(gi);
    return rc;
}
