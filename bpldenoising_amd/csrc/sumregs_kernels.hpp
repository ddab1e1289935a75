// sumregs_kernels.hpp -- CDNA4 (gfx950) kernels of the sum-of-regularisers learning function
//     min_u 0.5||u - f||^2 + a1 ||G_f u||_{2,1} + a2 ||G_b u||_{2,1} + a3 ||G_c u||_{2,1}
// (forward, backward and centred differences; /root/reference/src/SumRegsLearningFunction.jl:8-85 call the external
// `sumregs_denoise_pdps`; adjoint gradients :87-407).  Arithmetic = oracle/sumregs_oracle.c, reproduced bit for bit
// for the PDHG recurrence (explicit fma only; build with -ffp-contract=off).  PARITY UNPINNED: operators' border
// conventions and the recurrence are the oracle's choices (see its header).
//
// PDHG kernel: the design of pdhg_tile_kernel (pdhg_kernels.hpp) with three duals.  One workgroup owns one tile of
// one image: region = core + halo, x, f, six dual components and the three parameters in registers, `nit` fused
// iterations on chip, neighbours through seven LDS planes (six duals + xbar), two barriers per iteration.  The
// primal step reads duals at distance one in all four directions (backward and centred stencils), the dual step
// reads xbar at distance one in all four directions, so the validity front shrinks by TWO pixels per iteration on
// every side that is not an image border: halo = 2 * nit.
// Algorithmic traffic per pixel and iteration: read x, 6 y, f and write x, 6 y = 15 words = 120 B (144 B with three
// parameter maps) -- SURVEY.md 8(f) quotes 13/16 for a layout without the f read; the kernel is arithmetic bound.
#pragma once
#include <hip/hip_runtime.h>
#include "pdhg_kernels.hpp"

namespace bpltv {

constexpr size_t sr_lds_bytes(int RI, int RJ) { return sizeof(double) * (7 * (size_t)RI * RJ + 1); }   // seven planes + one zero cell

struct SrArgs {
    const double* in[7];   // x, yf1, yf2, yb1, yb2, yc1, yc2
    double* out[7];
    const double* f;
    const double* alpha;   // 3 slices of am*an doubles
    const double* tab;     // [maxiter][TAB_STRIDE], L = sqrt(18)
    double rho;
    int am, an;
    int it0, nit;
    int M, N, O;
    int nTi, nTj, halo;    // halo = 2 * fused iterations
    int first;
    int img0;              // first image of this launch (launch chains: grid = tiles per image * images of the chain)
};

__device__ __forceinline__ size_t sr_alpha_index(int am, int an, int M, int N, int i, int j) {
    if (am == 1 && an == 1) return 0;
    if (am == M && an == N) return i + (size_t)M * j;
    return (size_t)(((unsigned)i * (unsigned)am) / (unsigned)M) + (size_t)am * (((unsigned)j * (unsigned)an) / (unsigned)N);
}

template <int TI, int TJ>
__global__ __launch_bounds__(TI* TJ) void sr_tile_kernel(SrArgs A) {
    constexpr int RI = TI, RJ = TJ, RN = RI * RJ;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sy = smem;              // six planes [RJ][RI]
    double* sxb = smem + 6 * RN;
    const int tid = threadIdx.x;
    const int li = tid % TI, lj = tid / TI;
    // grid (nTi, nTj, images of the chain): tile and image without an integer division
    const int ta = (int)blockIdx.x, tb = (int)blockIdx.y;
    const int img = A.img0 + (int)blockIdx.z;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    const size_t base = (size_t)img * M * N;
    const int gi = oi + li, gj = oj + lj;
    const bool in = gi < M && gj < N;
    const int ci = min(gi, M - 1), cj = min(gj, N - 1);
    const size_t g = base + ci + (size_t)M * cj;
    const size_t ai = sr_alpha_index(A.am, A.an, M, N, ci, cj), astride = (size_t)A.am * A.an;
    // ---- prologue: all global loads first
    double x, f, y[6], al[3];
    f = A.f[g];
    al[0] = A.alpha[ai]; al[1] = A.alpha[astride + ai]; al[2] = A.alpha[2 * astride + ai];
    if (!A.first) {
        x = A.in[0][g];
#pragma unroll
        for (int c = 0; c < 6; ++c) y[c] = A.in[1 + c][g];
    } else {
        x = f;
#pragma unroll
        for (int c = 0; c < 6; ++c) y[c] = 0.0;
    }
    if (!in) {
        x = 0.0; f = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) y[c] = 0.0;
        al[0] = al[1] = al[2] = 0.0;
    }
    const int l = lj * RI + li;
    // image-border flags (Neumann family) and clamped neighbour offsets inside the region: where the neighbour does
    // not exist in the image the formulas select a constant; where it only leaves the REGION (halo edge) the own cell
    // is read -- those pixels are outside the validity front and are never written back.
    const bool hasL = gi > 0, hasR = gi < M - 1, hasU = gj > 0, hasD = gj < N - 1;
    const int nim = l - ((hasL && li > 0) ? 1 : 0), nip = l + ((hasR && li < RI - 1) ? 1 : 0);
    const int njm = l - ((hasU && lj > 0) ? RI : 0), njp = l + ((hasD && lj < RJ - 1) ? RI : 0);
    // The four one-sided neighbour reads whose border value is the constant 0 address a zero cell behind the planes
    // instead of being selected after the read (zf1m etc. are offsets into their own plane).
    const int zf1m = hasL ? nim : (7 - 0) * RN, zf2m = hasU ? njm : (7 - 1) * RN;
    const int zb1p = hasR ? nip : (7 - 2) * RN, zb2p = hasD ? njp : (7 - 3) * RN;
    if (tid == 0) smem[7 * RN] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) sy[c * RN + l] = y[c];
    __syncthreads();

    const double rho = A.rho;
    const double* __restrict__ row = A.tab + (size_t)TAB_STRIDE * A.it0;
    double tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    for (int it = 0; it < A.nit; ++it) {
        const double* __restrict__ nrow = row + TAB_STRIDE * ((it + 1 < A.nit) ? it + 1 : it);
        const double ntau = nrow[0], nsigma = nrow[1], nomega = nrow[2], ninv1ptau = nrow[3], nopw = nrow[4];
        // ---- primal step: div = (G_f^T y_f + G_b^T y_b) + G_c^T y_c in the oracle's gather order (sr_gradT_at)
        const double f1m = sy[0 * RN + zf1m], f2m = sy[1 * RN + zf2m];   // 0 where the neighbour is outside the image
        const double b1p = sy[2 * RN + zb1p], b2p = sy[3 * RN + zb2p];
        const double c1m = sy[4 * RN + nim], c1p = sy[4 * RN + nip];
        const double c2m = sy[5 * RN + njm], c2p = sy[5 * RN + njp];
        // The oracle's (hasR ? yf1 : 0) etc. need no select: a one-sided difference across the image border is x - x =
        // +0, so yf1 on the last row, yf2 on the last column, yb1 on the first row and yb2 on the first column stay +0
        // through every iteration (fma(sigma, +0, +0) = +0; the projection multiplies by a finite factor).
        const double tf = (f1m - y[0]) + (f2m - y[1]);
        const double tbk = (y[2] - b1p) + (y[3] - b2p);
        const double ca = hasL ? c1m : -y[4], cb = hasR ? c1p : -y[4];
        const double cc = hasU ? c2m : -y[5], cd = hasD ? c2p : -y[5];
        const double tc = 0.5 * (ca - cb) + 0.5 * (cc - cd);
        const double div = (tf + tbk) + tc;
        const double tt = div - f;
        const double xo = x;
        const double xn = __builtin_fma(-tau, tt, xo) * inv1ptau;
        const double bc = __builtin_fma(-omega, xo, opw * xn);
        x = xn;
        sxb[l] = bc;
        __syncthreads();
        // ---- dual steps
        const double bp = sxb[nip], bm = sxb[nim], cp = sxb[njp], cm = sxb[njm];
        double d1[3], d2[3];
        d1[0] = bp - bc; d2[0] = cp - bc;
        d1[1] = bc - bm; d2[1] = bc - cm;
        d1[2] = 0.5 * (bp - bm); d2[2] = 0.5 * (cp - cm);
        double n2v[3];
        bool any_out = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = al[k];
            double y1n = __builtin_fma(sigma, d1[k], y[2 * k]);
            double y2n = __builtin_fma(sigma, d2[k], y[2 * k + 1]);
            if (rho != 0.0) {
                const double den = 1.0 + sigma * rho / a;
                y1n = y1n / den;
                y2n = y2n / den;
            }
            y[2 * k] = y1n;
            y[2 * k + 1] = y2n;
            n2v[k] = __builtin_fma(y2n, y2n, y1n * y1n);
            any_out |= n2v[k] > a * a;
        }
        (void)any_out;
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // per regulariser: lanes outside the ball project (a wave with none skips the chain)
            const double a = al[k];
            if (n2v[k] > a * a) {
                const double v = a * rsqrt_nr(n2v[k]);
                y[2 * k] = y[2 * k] * v;
                y[2 * k + 1] = y[2 * k + 1] * v;
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) sy[c * RN + l] = y[c];
        tau = ntau; sigma = nsigma; omega = nomega; inv1ptau = ninv1ptau; opw = nopw;
        __syncthreads();
    }
    if (gi >= ci0 && gi < ci1 && gj >= cj0 && gj < cj1) {
        const size_t idx = base + gi + (size_t)M * gj;
        // write-through (sc1) stores, as in pdhg_tile_kernel: the next launch reads this state from other XCDs
        __hip_atomic_store(&A.out[0][idx], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int c = 0; c < 6; ++c) __hip_atomic_store(&A.out[1 + c][idx], y[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same iteration with PJ pixels per thread, stacked along j (a thread owns the strip lj = PJ*tj .. PJ*tj + PJ-1 of
// column li): region TI x (PJ*TJ).  48 x 48 with three pixels per thread has a 32 x 32 core at T = 4 -- redundancy 2.25
// instead of the 4.0 of the one-pixel kernel -- and a 128^2 image is 4 x 4 tiles.  j-neighbours inside the strip stay
// in registers: of the 12 LDS reads and 7 writes per pixel and iteration of sr_tile_kernel, 8 and 5.33 remain at
// PJ = 3 (planes yf2 / yb2 are only read across a strip boundary, from the strip's last / first pixel; yc2 from both).
// Same operation sequence per pixel: bit-identical to sr_tile_kernel and to the oracle.
template <int PJ, int TI, int TJ>
__global__ __launch_bounds__(TI* TJ) void sr_strip_kernel(SrArgs A) {
    constexpr int RI = TI, RJ = PJ * TJ, RN = RI * RJ;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sy = smem;              // six planes [RJ][RI]
    double* sxb = smem + 6 * RN;
    const int tid = threadIdx.x;
    const int li = tid % TI, tj = tid / TI, lj0 = PJ * tj;
    // grid (nTi, nTj, images of the chain): tile and image without an integer division
    const int ta = (int)blockIdx.x, tb = (int)blockIdx.y;
    const int img = A.img0 + (int)blockIdx.z;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    const size_t base = (size_t)img * M * N;
    const int gi = oi + li, ci = min(gi, M - 1);
    const size_t astride = (size_t)A.am * A.an;
    double x[PJ], f[PJ], y[PJ][6], al[PJ][3], bc[PJ];
    size_t g[PJ];
    bool hasU[PJ], hasD[PJ];
    // ---- prologue: all global loads first
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = oj + lj0 + pj, cj = min(gj, N - 1);
        g[pj] = base + ci + (size_t)M * cj;
        const size_t ai = sr_alpha_index(A.am, A.an, M, N, ci, cj);
        f[pj] = A.f[g[pj]];
        al[pj][0] = A.alpha[ai]; al[pj][1] = A.alpha[astride + ai]; al[pj][2] = A.alpha[2 * astride + ai];
        if (!A.first) {
            x[pj] = A.in[0][g[pj]];
#pragma unroll
            for (int c = 0; c < 6; ++c) y[pj][c] = A.in[1 + c][g[pj]];
        }
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = oj + lj0 + pj;
        if (A.first) {
            x[pj] = f[pj];
#pragma unroll
            for (int c = 0; c < 6; ++c) y[pj][c] = 0.0;
        }
        if (!(gi < M && gj < N)) {
            x[pj] = 0.0; f[pj] = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) y[pj][c] = 0.0;
            al[pj][0] = al[pj][1] = al[pj][2] = 0.0;
        }
        hasU[pj] = gj > 0;
        hasD[pj] = gj < N - 1;
    }
    const bool hasL = gi > 0, hasR = gi < M - 1;
    const int l0 = lj0 * RI + li, lE = l0 + (PJ - 1) * RI;   // the strip's first and last cell
    const int dm = (hasL && li > 0) ? -1 : 0, dp = (hasR && li < RI - 1) ? 1 : 0;
    // strip ends: the j-neighbour in the region, or a written cell of the same plane where it leaves the region or the
    // image (there the value is either replaced by the border constant or belongs to a pixel outside the validity front)
    const bool upIn = hasU[0] && lj0 > 0, dnIn = hasD[PJ - 1] && lj0 + PJ < RJ;
    const int upC = upIn ? l0 - RI : l0, dnC = dnIn ? lE + RI : lE;   // planes written at both strip ends (yc2, xbar)
    // yf2 is written at the strip's last cell only, yb2 at its first: outside the region a written cell of the own strip,
    // outside the image the zero cell behind the planes (as zf1m / zb1p for the i-neighbours, see sr_tile_kernel)
    const int upF = hasU[0] ? (upIn ? l0 - RI : lE) : (7 - 1) * RN;
    const int dnB = hasD[PJ - 1] ? (dnIn ? lE + RI : l0) : (7 - 3) * RN;
    const int zdm = hasL ? dm : (7 - 0) * RN - l0, zdp = hasR ? dp : (7 - 2) * RN - l0;   // relative to the strip's first cell
    if (tid == 0) smem[7 * RN] = 0.0;
    auto store_duals = [&]() {
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            const int l = l0 + pj * RI;
            sy[0 * RN + l] = y[pj][0];
            sy[2 * RN + l] = y[pj][2];
            sy[4 * RN + l] = y[pj][4];
            if (pj == PJ - 1) sy[1 * RN + l] = y[pj][1];
            if (pj == 0) sy[3 * RN + l] = y[pj][3];
            if (pj == 0 || pj == PJ - 1) sy[5 * RN + l] = y[pj][5];
        }
    };
    store_duals();
    __syncthreads();

    const double rho = A.rho;
    const double* __restrict__ row = A.tab + (size_t)TAB_STRIDE * A.it0;
    double tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    for (int it = 0; it < A.nit; ++it) {
        const double* __restrict__ nrow = row + TAB_STRIDE * ((it + 1 < A.nit) ? it + 1 : it);
        const double ntau = nrow[0], nsigma = nrow[1], nomega = nrow[2], ninv1ptau = nrow[3], nopw = nrow[4];
        // ---- primal step (sr_tile_kernel's expression order)
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            const int l = l0 + pj * RI;
            const double f1m = sy[0 * RN + (hasL ? l : l0) + zdm], b1p = sy[2 * RN + (hasR ? l : l0) + zdp];
            const double c1m = sy[4 * RN + l + dm], c1p = sy[4 * RN + l + dp];
            const double f2m = (pj > 0) ? y[pj > 0 ? pj - 1 : 0][1] : sy[1 * RN + upF];
            const double c2m = (pj > 0) ? y[pj > 0 ? pj - 1 : 0][5] : sy[5 * RN + upC];
            // inside the strip the j+1 neighbour is a register (an out-of-image pixel beyond the last column holds +-0:
            // the select keeps the oracle's +0)
            const bool hU = hasU[pj], hD = hasD[pj];
            const double b2p = (pj < PJ - 1) ? (hD ? y[pj < PJ - 1 ? pj + 1 : pj][3] : 0.0) : sy[3 * RN + dnB];
            const double c2p = (pj < PJ - 1) ? y[pj < PJ - 1 ? pj + 1 : pj][5] : sy[5 * RN + dnC];
            const double tf = (f1m - y[pj][0]) + (f2m - y[pj][1]);   // selects dropped as in sr_tile_kernel
            const double tbk = (y[pj][2] - b1p) + (y[pj][3] - b2p);
            const double ca = hasL ? c1m : -y[pj][4], cb = hasR ? c1p : -y[pj][4];
            const double cc = hU ? c2m : -y[pj][5], cd = hD ? c2p : -y[pj][5];
            const double tc = 0.5 * (ca - cb) + 0.5 * (cc - cd);
            const double div = (tf + tbk) + tc;
            const double tt = div - f[pj];
            const double xo = x[pj];
            const double xn = __builtin_fma(-tau, tt, xo) * inv1ptau;
            bc[pj] = __builtin_fma(-omega, xo, opw * xn);
            x[pj] = xn;
            sxb[l] = bc[pj];
        }
        __syncthreads();
        // ---- dual steps
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            const int l = l0 + pj * RI;
            const double bcc = bc[pj];
            const double bp = sxb[l + dp], bm = sxb[l + dm];
            const double cp = (pj < PJ - 1) ? (hasD[pj] ? bc[pj < PJ - 1 ? pj + 1 : pj] : bcc) : sxb[dnC];
            const double cm = (pj > 0) ? bc[pj > 0 ? pj - 1 : 0] : sxb[upC];
            double d1[3], d2[3];
            d1[0] = bp - bcc; d2[0] = cp - bcc;
            d1[1] = bcc - bm; d2[1] = bcc - cm;
            d1[2] = 0.5 * (bp - bm); d2[2] = 0.5 * (cp - cm);
            double n2v[3];
            bool any_out = false;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double a = al[pj][k];
                double y1n = __builtin_fma(sigma, d1[k], y[pj][2 * k]);
                double y2n = __builtin_fma(sigma, d2[k], y[pj][2 * k + 1]);
                if (rho != 0.0) {
                    const double den = 1.0 + sigma * rho / a;
                    y1n = y1n / den;
                    y2n = y2n / den;
                }
                y[pj][2 * k] = y1n;
                y[pj][2 * k + 1] = y2n;
                n2v[k] = __builtin_fma(y2n, y2n, y1n * y1n);
                any_out |= n2v[k] > a * a;
            }
            (void)any_out;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double a = al[pj][k];
                if (n2v[k] > a * a) {
                    const double v = a * rsqrt_nr(n2v[k]);
                    y[pj][2 * k] = y[pj][2 * k] * v;
                    y[pj][2 * k + 1] = y[pj][2 * k + 1] * v;
                }
            }
        }
        store_duals();
        tau = ntau; sigma = nsigma; omega = nomega; inv1ptau = ninv1ptau; opw = nopw;
        __syncthreads();
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = oj + lj0 + pj;
        if (gi >= ci0 && gi < ci1 && gj >= cj0 && gj < cj1) {
            __hip_atomic_store(&A.out[0][g[pj]], x[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int c = 0; c < 6; ++c) __hip_atomic_store(&A.out[1 + c][g[pj]], y[pj][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Adjoint gradients (SumRegsLearningFunction.jl:87-407): the reduced SPD system of oracle/sumregs_oracle.c
//     (I + sum_k G_k^T W_k G_k) p = rhs,   W_k per element = c t t^T + kap I,
// bandwidth 2M in the column-major pixel order, factored by the HBM band solver (hb_band_solver.hpp).
// ------------------------------------------------------------------------------------------------------------
struct SrCoef {   // planar [3][O][M*N] each, + rhs [O][M*N]
    double *t1, *t2, *c, *kap, *h1, *h2, *rhs;
    size_t tot;   // O*M*N
};

constexpr double SR_ACT_TOL = 1e-12;   // SumRegsLearningFunction.jl:274,290,306

// (G_k v)_e at element (i, j) of one image (v: that image's plane)
__device__ __forceinline__ void sr_grad_at(int k, const double* __restrict__ v, int M, int N, int i, int j, double& d1, double& d2) {
    const size_t q = i + (size_t)M * j;
    const double uc = v[q];
    const double up = v[(i < M - 1) ? q + 1 : q], um = v[(i > 0) ? q - 1 : q];
    const double vp = v[(j < N - 1) ? q + M : q], vm = v[(j > 0) ? q - M : q];
    if (k == 0) { d1 = up - uc; d2 = vp - uc; }
    else if (k == 1) { d1 = uc - um; d2 = uc - vm; }
    else { d1 = 0.5 * (up - um); d2 = 0.5 * (vp - vm); }
}

// (G_k^T y)(q), the oracle's gather form (sr_gradT_at)
__device__ __forceinline__ double sr_gradT_at(int k, const double* __restrict__ y1, const double* __restrict__ y2, int M, int N, int i,
                                              int j) {
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

// Primal-dual gap pieces of the three-dual model per image, the layout of gap_partial_kernel (pdhg_kernels.hpp):
// partial[(k*nblk+b)*4 + {0: ||u-f||^2, 1: sum_k sum alpha_k |G_k u|, 2: ||f||^2, 3: ||f - K^T y||^2}], K^T y =
// (G_f^T y_f + G_b^T y_b) + G_c^T y_c; gap_final_kernel turns them into gap_k >= 0.5 ||u_k - u*_k||^2 for every feasible
// dual.  Checker: bplo_sumregs_gap.  state: the seven planes x, yf1, yf2, yb1, yb2, yc1, yc2.  grid (nblk, O), block 256.
struct SrState { const double* pl[7]; };
__global__ __launch_bounds__(256) void sr_gap_partial_kernel(SrState S, const double* __restrict__ f, const double* __restrict__ alpha,
                                                             int am, int an, int M, int N, double* __restrict__ partial) {
    __shared__ double sh[4];
    const int npx = M * N;
    const size_t base = (size_t)blockIdx.y * npx;
    const double* u = S.pl[0] + base;
    const size_t asl = (size_t)am * an;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < npx; q += gridDim.x * 256) {
        const int i = q % M, j = q / M;
        const double uk = u[q], fk = f[base + q];
        const size_t ai = sr_alpha_index(am, an, M, N, i, j);
        double w = 0.0, tv = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double d1, d2;
            sr_grad_at(k, u, M, N, i, j, d1, d2);
            tv += alpha[(size_t)k * asl + ai] * sqrt(d1 * d1 + d2 * d2);
        }
        w = (sr_gradT_at(0, S.pl[1] + base, S.pl[2] + base, M, N, i, j) + sr_gradT_at(1, S.pl[3] + base, S.pl[4] + base, M, N, i, j)) +
            sr_gradT_at(2, S.pl[5] + base, S.pl[6] + base, M, N, i, j);
        const double r = uk - fk;
        s0 += r * r;
        s1 += tv;
        s2 += fk * fk;
        s3 += (fk - w) * (fk - w);
    }
    s0 = block_sum<256>(s0, sh);
    s1 = block_sum<256>(s1, sh);
    s2 = block_sum<256>(s2, sh);
    s3 = block_sum<256>(s3, sh);
    if (threadIdx.x == 0) {
        double* p = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        p[0] = s0; p[1] = s1; p[2] = s2; p[3] = s3;
    }
}

// per element and operator: coefficients of W_k and of the gradient functional; grid over O*M*N, one thread per
// pixel, all three operators.  reg: gradient_reg (gamma = 1e3 vector / 1e8 patch parameter).
__global__ __launch_bounds__(256) void sr_adj_setup_kernel(const double* __restrict__ u, const double* __restrict__ ubar,
                                                           const double* __restrict__ alpha, int am, int an, int M, int N, int O,
                                                           int patch, int reg, double kappa_act, SrCoef C) {
    // reg && patch: the parameter scales ROWS of term k (SumRegsLearningFunction.jl:250) and stays out of c, kap
    const size_t npx = (size_t)M * N;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npx * O) return;
    const int img = (int)(e / npx), k0 = (int)(e - (size_t)img * npx);
    const int i = k0 % M, j = k0 / M;
    const double* ui = u + (size_t)img * npx;
    const size_t ai = sr_alpha_index(am, an, M, N, i, j), astride = (size_t)am * an;
    const double gamma = patch ? 1e8 : 1e3;   // SumRegsLearningFunction.jl:200 / :117
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double g1, g2;
        sr_grad_at(k, ui, M, N, i, j, g1, g2);
        const double ng = sqrt(g1 * g1 + g2 * g2);
        const double a = alpha[k * astride + ai];
        double t1 = 0.0, t2 = 0.0, c = 0.0, kap = 0.0, h1 = 0.0, h2 = 0.0;
        if (!reg) {
            if (ng < SR_ACT_TOL) {
                kap = kappa_act;
            } else {
                t1 = -g2 / ng; t2 = g1 / ng;
                c = a / ng;
                h1 = g1 / ng; h2 = g2 / ng;
            }
        } else {
            if (ng > 1.0 / gamma) {
                t1 = -g2 / ng; t2 = g1 / ng;
                c = patch ? 1.0 / ng : a / ng;
                h1 = g1 / ng; h2 = g2 / ng;
            } else {
                kap = patch ? gamma : a * gamma;
                h1 = gamma * g1; h2 = gamma * g2;
            }
        }
        const size_t o = (size_t)k * C.tot + e;
        C.t1[o] = t1; C.t2[o] = t2; C.c[o] = c; C.kap[o] = kap; C.h1[o] = h1; C.h2[o] = h2;
    }
    C.rhs[e] = reg ? ubar[e] - u[e] : u[e] - ubar[e];
}

// Element stencil: component c couples node pl (coefficient +s) and node mi (-s); s = 0: absent.
struct SrStencil { int pl[2], mi[2]; double s[2]; };
__device__ __forceinline__ SrStencil sr_stencil(int k, int M, int N, int i, int j) {
    const int q = i + M * j;
    SrStencil S;
    if (k == 0) {
        S.pl[0] = (i < M - 1) ? q + 1 : q; S.mi[0] = q; S.s[0] = (i < M - 1) ? 1.0 : 0.0;
        S.pl[1] = (j < N - 1) ? q + M : q; S.mi[1] = q; S.s[1] = (j < N - 1) ? 1.0 : 0.0;
    } else if (k == 1) {
        S.pl[0] = q; S.mi[0] = (i > 0) ? q - 1 : q; S.s[0] = (i > 0) ? 1.0 : 0.0;
        S.pl[1] = q; S.mi[1] = (j > 0) ? q - M : q; S.s[1] = (j > 0) ? 1.0 : 0.0;
    } else {
        S.pl[0] = (i < M - 1) ? q + 1 : q; S.mi[0] = (i > 0) ? q - 1 : q; S.s[0] = (S.pl[0] != S.mi[0]) ? 0.5 : 0.0;
        S.pl[1] = (j < N - 1) ? q + M : q; S.mi[1] = (j > 0) ? q - M : q; S.s[1] = (S.pl[1] != S.mi[1]) ? 0.5 : 0.0;
    }
    return S;
}

// The seven non-zero diagonals of the lower band, offsets {0, 1, 2, M-1, M, M+1, 2M}: planes[t][O][M*N] holds
// A[q + off_t][q].  Gather form, one thread per column q: every element (operator k, pixel e) that has q among its
// nodes adds V_a^T W V_b for its node slots b on q and a on rows r >= q.  No atomics: the summation order is fixed.
__device__ __forceinline__ int sr_diag_slot(int d, int M) {
    // offsets may coincide for tiny M (M = 1, 2, 3): the first match wins, and hb band_entry adds coinciding planes
    if (d == 0) return 0;
    if (d == 1) return 1;
    if (d == 2) return 2;
    if (d == M - 1) return 3;
    if (d == M) return 4;
    if (d == M + 1) return 5;
    if (d == 2 * M) return 6;
    return -1;
}
// rowscale (nullable; with am, an: the parameter array) and planesU (nullable, zero-initialised by the caller): the
// non-symmetric row-scaled system of sumregs_gradient_reg with a patch parameter -- every entry A(r, q) of term k is
// multiplied by x_k at pixel r, and the strictly upper entries A(q - off, q) go to planesU[t][q - off] (by rows).
__global__ __launch_bounds__(256) void sr_adj_assemble_kernel(SrCoef C, int M, int N, int O, double* __restrict__ planes,
                                                              const double* __restrict__ rowscale, int am, int an,
                                                              double* __restrict__ planesU) {
    const size_t npx = (size_t)M * N;
    const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= npx * O) return;
    const int img = (int)(col / npx), q = (int)(col - (size_t)img * npx);
    const int i = q % M, j = q / M;
    double acc[7] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double accu[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const size_t astride = (size_t)am * an;
    const size_t ib = (size_t)img * npx;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // candidate elements whose stencil can contain q: q itself and its four neighbours
        const int ce[5] = {q, (i > 0) ? q - 1 : -1, (i < M - 1) ? q + 1 : -1, (j > 0) ? q - M : -1, (j < N - 1) ? q + M : -1};
#pragma unroll
        for (int cidx = 0; cidx < 5; ++cidx) {
            const int e = ce[cidx];
            if (e < 0) continue;
            const SrStencil S = sr_stencil(k, M, N, e % M, e / M);
            const int node[4] = {S.pl[0], S.mi[0], S.pl[1], S.mi[1]};
            const double v1[4] = {S.s[0], -S.s[0], 0.0, 0.0}, v2[4] = {0.0, 0.0, S.s[1], -S.s[1]};
            bool has = false;
#pragma unroll
            for (int b = 0; b < 4; ++b) has |= node[b] == q && (v1[b] != 0.0 || v2[b] != 0.0);
            if (!has) continue;
            const size_t o = (size_t)k * C.tot + ib + e;
            const double t1 = C.t1[o], t2 = C.t2[o], c = C.c[o], kp = C.kap[o];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (node[b] != q) continue;
                const double tb = t1 * v1[b] + t2 * v2[b];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int d = node[a] - q;
                    if (d < 0 && !planesU) continue;
                    const double ta = t1 * v1[a] + t2 * v2[a];
                    double val = c * ta * tb + kp * (v1[a] * v1[b] + v2[a] * v2[b]);
                    if (rowscale) val *= rowscale[k * astride + sr_alpha_index(am, an, M, N, node[a] % M, node[a] / M)];
                    const int sl = sr_diag_slot(d < 0 ? -d : d, M);
                    if (sl >= 0) { if (d >= 0) acc[sl] += val; else accu[sl] += val; }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) planes[(size_t)t * C.tot + col] = acc[t];
    if (planesU) {
        const int off[7] = {0, 1, 2, M - 1, M, M + 1, 2 * M};
#pragma unroll
        for (int t = 1; t < 7; ++t)
            if (q - off[t] >= 0 && sr_diag_slot(off[t], M) == t) planesU[(size_t)t * C.tot + ib + (q - off[t])] = accu[t];
    }
}

// residual, pass 1: w_k = W_k (G_k p) per element -> w[(2k + c) * tot + e]
__global__ __launch_bounds__(256) void sr_adj_flux_kernel(SrCoef C, const double* __restrict__ p, int M, int N, int O,
                                                          double* __restrict__ w) {
    const size_t npx = (size_t)M * N;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npx * O) return;
    const int img = (int)(e / npx), k0 = (int)(e - (size_t)img * npx);
    const int i = k0 % M, j = k0 / M;
    const double* pi = p + (size_t)img * npx;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double d1, d2;
        sr_grad_at(k, pi, M, N, i, j, d1, d2);
        const size_t o = (size_t)k * C.tot + e;
        const double t1 = C.t1[o], t2 = C.t2[o], c = C.c[o], kp = C.kap[o];
        const double bp = t1 * d1 + t2 * d2;
        w[(size_t)(2 * k) * C.tot + e] = c * bp * t1 + kp * d1;
        w[(size_t)(2 * k + 1) * C.tot + e] = c * bp * t2 + kp * d2;
    }
}

// residual, pass 2: out = rhs - (p + sum_k G_k^T w_k)
__global__ __launch_bounds__(256) void sr_adj_residual_kernel(SrCoef C, const double* __restrict__ p, const double* __restrict__ w,
                                                              int M, int N, int O, double* __restrict__ out,
                                                              const double* __restrict__ rowscale, int am, int an) {
    const size_t npx = (size_t)M * N;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npx * O) return;
    const int img = (int)(e / npx), k0 = (int)(e - (size_t)img * npx);
    const int i = k0 % M, j = k0 / M;
    const size_t ib = (size_t)img * npx;
    double s = p[e];
    const size_t ai = rowscale ? sr_alpha_index(am, an, M, N, i, j) : 0, astride = (size_t)am * an;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double g = sr_gradT_at(k, w + (size_t)(2 * k) * C.tot + ib, w + (size_t)(2 * k + 1) * C.tot + ib, M, N, i, j);
        s += rowscale ? rowscale[k * astride + ai] * g : g;
    }
    out[e] = C.rhs[e] - s;
}

// per-pixel gradient contributions, three planes gpix[k][O][M*N]
//   vector parameter: (G_k p)_e . h_e per element (:326 / :165);  patch parameter: p_q (G_k^T h_k)_q per node (:397-399 / :251-253)
__global__ __launch_bounds__(256) void sr_adj_gradpix_kernel(SrCoef C, const double* __restrict__ p, int M, int N, int O, int patch,
                                                             int reg, double* __restrict__ gpix) {
    const size_t npx = (size_t)M * N;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npx * O) return;
    const int img = (int)(e / npx), k0 = (int)(e - (size_t)img * npx);
    const int i = k0 % M, j = k0 / M;
    const size_t ib = (size_t)img * npx;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double v;
        if (!patch) {
            double d1, d2;
            sr_grad_at(k, p + ib, M, N, i, j, d1, d2);
            v = d1 * C.h1[(size_t)k * C.tot + e] + d2 * C.h2[(size_t)k * C.tot + e];
        } else {
            v = p[e] * sr_gradT_at(k, C.h1 + (size_t)k * C.tot + ib, C.h2 + (size_t)k * C.tot + ib, M, N, i, j);
        }
        gpix[(size_t)k * C.tot + e] = reg ? v : -v;
    }
}

}  // namespace bpltv
