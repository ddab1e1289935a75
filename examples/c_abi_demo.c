/* c_abi_demo.c -- libbpltv driven from plain C, the way a Julia `ccall` (or any FFI) would.
 *
 *   gcc -I include examples/c_abi_demo.c -o c_abi_demo -L bpldenoising_amd -lbpltv -Wl,-rpath,$PWD/bpldenoising_amd -lm
 *   ./c_abi_demo            (needs an MI355X; prints cost, gradient and timing of one evaluation)
 *
 * Mirrors tv_op_learning_function(x, (ubar, f), D) of /root/reference/src/TVLearningFunctionVec.jl:14-27
 * on a synthetic 4 x 128 x 128 batch: column-major M x N x O doubles in, (u, cost, grad) out.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "bpltv.h"

int main(void)
{
    const int M = 128, N = 128, O = 4;
    const size_t n = (size_t)M * N * O;
    double *ubar = malloc(n * sizeof *ubar), *f = malloc(n * sizeof *f), *u = malloc(n * sizeof *u);
    unsigned r = 20211004u;
    for (size_t k = 0; k < n; ++k) { /* blocky truth + quantised noise */
        const int i = (int)(k % M), j = (int)((k / M) % N);
        const double t = ((i / 32 + j / 32) & 1) ? 0.75 : 0.25;
        r = r * 1664525u + 1013904223u;
        double v = t + 0.2 * ((r >> 8) / 16777216.0 - 0.5);
        v = v < 0 ? 0 : (v > 1 ? 1 : v);
        ubar[k] = t;
        f[k] = floor(255.0 * v + 0.5) / 255.0;
    }
    bpltv_t *h = NULL;
    int rc = bpltv_create(&h, M, N, O, -1, 64);
    if (rc) { fprintf(stderr, "bpltv_create: %d (%s)\n", rc, bpltv_last_error(h)); bpltv_destroy(h); return 1; }
    bpltv_params p;
    bpltv_default_params(&p); /* rho 0, tau0 5, sigma0 0.99/5, accel, maxiter 5000 */
    if ((rc = bpltv_set_data(h, ubar, f))) goto fail;

    double alpha = 0.1, cost = 0, grad = 0;
    if ((rc = bpltv_evaluate(h, &alpha, 1, 1, /*Delta=*/0.1, &p, u, &cost, &grad))) goto fail;
    bpltv_stats_t st;
    bpltv_stats(h, &st);
    printf("scalar alpha %.3f: cost %.10f grad %.8f | PDHG %d iterations in %.3f ms (%d launches, T=%d), adjoint %.3f ms\n",
           alpha, cost, grad, st.iterations, st.pdhg_ms, st.launches, st.tile_iters, st.adjoint_ms);

    double a22[4] = {0.05, 0.1, 0.2, 0.08}, g22[4]; /* 2 x 2 patch parameter, column major */
    if ((rc = bpltv_evaluate(h, a22, 2, 2, 0.1, &p, NULL, &cost, g22))) goto fail;
    printf("2x2 patch alpha: cost %.10f grad [%.6f %.6f; %.6f %.6f]\n", cost, g22[0], g22[2], g22[1], g22[3]);

    double gap[4];
    if ((rc = bpltv_duality_gap(h, gap))) goto fail;
    printf("duality gaps of the last solve: %.3e %.3e %.3e %.3e\n", gap[0], gap[1], gap[2], gap[3]);

    double sweep_a[8] = {0.0, 0.02, 0.04, 0.06, 0.08, 0.1, 0.15, 0.2}, sweep_c[8];
    if ((rc = bpltv_sweep(h, sweep_a, 8, 1, 1, &p, sweep_c, NULL))) goto fail;
    printf("cost curve:");
    for (int k = 0; k < 8; ++k) printf(" (%.2f, %.4f)", sweep_a[k], sweep_c[k]);
    printf("\n");
    bpltv_destroy(h);
    free(ubar); free(f); free(u);
    return 0;
fail:
    fprintf(stderr, "libbpltv error %d: %s\n", rc, bpltv_last_error(h));
    bpltv_destroy(h);
    return 1;
}
