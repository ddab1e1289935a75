// adjoint_hbm_kernels.hpp -- adjoint solve for images too wide for the LDS window (M > 138).
//
// Same reduced SPD system and same pipeline as adjoint_kernels.hpp, but the band lives in HBM:
// band[O][n][W], W = M+1 (column k of the lower band: entry (k+d, k) at d), factored IN PLACE.
// Grid-wide synchronisation is the kernel boundary: per panel of HB_NB columns one launch of
// hb_panel_kernel (diagonal block + triangular solve of the bw rows below it, rows split over
// workgroups) and one launch of hb_update_kernel (one 64x64 tile of the trailing triangle per
// workgroup, panel slices staged in LDS), all images in the same launches.  The substitutions run
// one 1024-thread workgroup per image with the vector in global memory (L2 resident).
#pragma once
#include <hip/hip_runtime.h>
#include "adjoint_kernels.hpp"

namespace bpltv {

constexpr int HB_NB = 32;   // panel width
constexpr int HB_ROWS = 256; // rows of the panel handled by one workgroup of hb_panel_kernel

// band <- assembled matrix (4 diagonals), zero elsewhere
__global__ __launch_bounds__(256) void hb_init_kernel(const double* __restrict__ band4, int M, int N, int O,
                                                      double* __restrict__ band) {
    const int W = M + 1;
    const size_t n = (size_t)M * N, tot = n * O;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= tot * W) return;
    const size_t col = e / W;  // global column index over images
    const int d = (int)(e - col * W);
    band[e] = band_init(band4, tot, col, d, M);
}

// Factor the HB_NB x HB_NB diagonal block at column k0 (redundantly in every workgroup, in wave 0
// registers with readlane broadcasts) and forward-substitute the rows below it.
// grid (ceil((bw + HB_NB) / HB_ROWS), O), block HB_ROWS.
__global__ __launch_bounds__(HB_ROWS) void hb_panel_kernel(double* __restrict__ band, int M, int N, int k0,
                                                           double* __restrict__ l11buf,
                                                           int* __restrict__ fail) {
    __shared__ double T[HB_NB][HB_NB + 1];  // L11, row major lower
    __shared__ double Dinv[HB_NB];
    __shared__ double LP[HB_ROWS][HB_NB + 1];
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    double* Bi = band + (size_t)img * n * W;
    const int nb = (n - k0 < HB_NB) ? (n - k0) : HB_NB;
    if (tid < 64) {  // wave 0: lane r holds row r of the diagonal block
        double m[HB_NB];
#pragma unroll
        for (int c = 0; c < HB_NB; ++c) {
            double v = (lane == c) ? 1.0 : 0.0;
            if (lane < nb && c < nb && lane >= c) v = Bi[(size_t)(k0 + c) * W + (lane - c)];
            m[c] = v;
        }
        bool bad = false;
#pragma unroll
        for (int c = 0; c < HB_NB; ++c) {
            const double piv = readlane_f64(m[c], c);
            if (!(piv > 0.0)) bad = true;
            double d, di;
            sqrt_rsqrt(piv, d, di);
            m[c] = (lane == c) ? d : m[c] * di;  // column c scaled (rows > c); rows < c hold unused values
#pragma unroll
            for (int q = c + 1; q < HB_NB; ++q) {
                const double lq = readlane_f64(m[c], q);  // L[q][c]
                m[q] = __builtin_fma(-m[c], lq, m[q]);      // A[r][q] -= L[r][c] L[q][c] (used for r >= q)
            }
            if (lane == c) Dinv[c] = di;
        }
        if (bad && lane == 0 && fail[img] == 0) fail[img] = k0 + 1;  // keep the first failing panel
        if (lane < HB_NB) {
#pragma unroll
            for (int c = 0; c < HB_NB; ++c) T[lane][c] = (c <= lane) ? m[c] : 0.0;
        }
    }
    __syncthreads();
    // rows of the panel: r = k0 + rr, rr in [0, bw + HB_NB)
    const int rr = blockIdx.x * HB_ROWS + tid;
    const int r = k0 + rr;
    if (rr >= bw + HB_NB || r >= n) return;
    if (rr < HB_NB) {
        // diagonal block rows.  The other workgroups of this image may still be reading the
        // unfactored block from the band, so L11 goes to a side buffer (every workgroup computes the
        // same values); hb_update_kernel copies it into the band after the kernel boundary.
        if (blockIdx.x == 0)
            for (int c = 0; c < HB_NB; ++c) l11buf[((size_t)img * HB_NB + rr) * HB_NB + c] = T[rr][c];
        return;
    }
    // forward substitution of this row against L11; the row's entries live in LDS (dynamic loops,
    // no register arrays): LP[tid][c]
    double* lrow = &LP[tid][0];
    for (int c = 0; c < HB_NB; ++c) {
        const int d = rr - c;
        double v = (c < nb && d <= bw) ? Bi[(size_t)(k0 + c) * W + d] : 0.0;
        for (int q = 0; q < c; ++q) v = __builtin_fma(-lrow[q], T[c][q], v);
        v *= Dinv[c];
        lrow[c] = v;
        if (c < nb && d <= bw) Bi[(size_t)(k0 + c) * W + d] = v;
    }
}

// Trailing update of one 64x64 tile: A[r][j] -= sum_c L[r][k0+c] L[j][k0+c] for r in tile rows,
// j in tile columns, r >= j.  grid (ntile, O) with ntile = nt(nt+1)/2, nt = ceil(bw/64); block 256.
__global__ __launch_bounds__(256) void hb_update_kernel(double* __restrict__ band, int M, int N, int k0,
                                                        const double* __restrict__ l11buf) {
    __shared__ double PA[64][HB_NB + 1];
    __shared__ double PB[64][HB_NB + 1];
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.y, tid = threadIdx.x;
    double* Bi = band + (size_t)img * n * W;
    if (blockIdx.x == 0) {  // L11 of this panel into the band (see hb_panel_kernel)
        const int nb = (n - k0 < HB_NB) ? (n - k0) : HB_NB;
        for (int e = tid; e < HB_NB * HB_NB; e += 256) {
            const int rr = e / HB_NB, c = e % HB_NB;
            if (c <= rr && rr < nb) Bi[(size_t)(k0 + c) * W + (rr - c)] = l11buf[((size_t)img * HB_NB + rr) * HB_NB + c];
        }
    }
    // decode the lower-triangular tile index
    int ta = 0, t = blockIdx.x;
    while (t > ta) { t -= ta + 1; ++ta; }
    const int tb = t;                       // tb <= ta
    const int base = k0 + HB_NB;            // first trailing row/column
    const int r0 = base + ta * 64, j0 = base + tb * 64;
    if (j0 >= n) return;
    for (int e = tid; e < 64 * HB_NB; e += 256) {
        const int c = e / 64, q = e % 64;   // consecutive q -> consecutive rows: coalesced
        const int ra = r0 + q, rb = j0 + q;
        const int da = ra - (k0 + c), db = rb - (k0 + c);
        PA[q][c] = (ra < n && da <= bw && k0 + c < n) ? Bi[(size_t)(k0 + c) * W + da] : 0.0;
        PB[q][c] = (rb < n && db <= bw && k0 + c < n) ? Bi[(size_t)(k0 + c) * W + db] : 0.0;
    }
    __syncthreads();
    const int tr = (tid & 15) * 4, tc = (tid >> 4) * 4;  // 4x4 elements per thread
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 8
    for (int c = 0; c < HB_NB; ++c) {
        double va[4], vb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { va[a] = PA[tr + a][c]; vb[a] = PB[tc + a][c]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(va[a], vb[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = r0 + tr + a, j = j0 + tc + b;
            const int d = r - j;
            if (r < n && j < n && d >= 0 && d <= bw && r <= k0 + HB_NB - 1 + bw) Bi[(size_t)j * W + d] -= acc[a][b];
        }
}

// Solve L L^T x = b in place, L in the band array; one workgroup of 1024 threads per image, blocks
// of HB_NB columns; x in global memory.  If acc != nullptr the solution is added to it.
__global__ __launch_bounds__(1024) void hb_solve_kernel(const double* __restrict__ band, int M, int N,
                                                        double* __restrict__ x, double* __restrict__ acc) {
    __shared__ double xs[HB_NB];
    __shared__ double part[HB_NB][33];
    const int W = M + 1, bw = M, n = M * N;
    const double* Bi = band + (size_t)blockIdx.x * n * W;
    double* xv = x + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nblk = (n + HB_NB - 1) / HB_NB;
    // ---- forward
    for (int bi = 0; bi < nblk; ++bi) {
        const int k0 = bi * HB_NB;
        const int nb = (n - k0 < HB_NB) ? (n - k0) : HB_NB;
        if (tid < 64) {
            // row `lane` of the 32x32 triangle and the reciprocal diagonal: all loads issued up
            // front, the 32-step chain then runs on registers and readlane broadcasts only
            double t[HB_NB];
#pragma unroll
            for (int c = 0; c < HB_NB; ++c)
                t[c] = (c < lane && lane < nb) ? Bi[(size_t)(k0 + c) * W + (lane - c)] : 0.0;
            const double di = (lane < nb) ? 1.0 / Bi[(size_t)(k0 + lane) * W] : 1.0;
            double val = (lane < nb) ? xv[k0 + lane] : 0.0;
#pragma unroll
            for (int c = 0; c < HB_NB; ++c) {
                const double v = readlane_f64(val, c) * readlane_f64(di, c);
                if (lane == c) val = v;
                if (lane > c) val = __builtin_fma(-t[c], v, val);
            }
            if (lane < nb) { xs[lane] = val; xv[k0 + lane] = val; }
        }
        __syncthreads();
        const int rend = (k0 + nb - 1 + bw < n - 1) ? (k0 + nb - 1 + bw) : (n - 1);
        for (int r = k0 + nb + tid; r <= rend; r += 1024) {
            double s = 0.0;
#pragma unroll 8
            for (int c = 0; c < nb; ++c) {
                const int d = r - (k0 + c);
                if (d <= bw) s = __builtin_fma(Bi[(size_t)(k0 + c) * W + d], xs[c], s);
            }
            xv[r] -= s;
        }
        __syncthreads();
    }
    // ---- backward
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int k0 = bi * HB_NB;
        const int nb = (n - k0 < HB_NB) ? (n - k0) : HB_NB;
        {   // tails: column c handled by 32 threads (tid / 32 = c)
            const int c = tid >> 5, q = tid & 31;
            double s = 0.0;
            if (c < nb) {
                const int kc = k0 + c;
                const int dlo = k0 + nb - kc;
                const int dhi = (n - 1 - kc < bw) ? (n - 1 - kc) : bw;
                for (int d = dlo + q; d <= dhi; d += 32) s = __builtin_fma(Bi[(size_t)kc * W + d], xv[kc + d], s);
            }
            part[c][q] = s;
        }
        __syncthreads();
        if (tid < 64) {
            double val = 0.0;
            if (lane < nb) {
                double s = 0.0;
                for (int q = 0; q < 32; ++q) s += part[lane][q];
                val = xv[k0 + lane] - s;
            }
            double t[HB_NB];  // column `lane` of the triangle: L[k0+cc][k0+lane], cc > lane
#pragma unroll
            for (int cc = 0; cc < HB_NB; ++cc)
                t[cc] = (cc > lane && cc < nb && lane < nb) ? Bi[(size_t)(k0 + lane) * W + (cc - lane)] : 0.0;
            const double di = (lane < nb) ? 1.0 / Bi[(size_t)(k0 + lane) * W] : 1.0;
#pragma unroll
            for (int c = HB_NB - 1; c >= 0; --c) {
                const double v = readlane_f64(val, c) * readlane_f64(di, c);
                if (lane == c) val = v;
                if (lane < c) val = __builtin_fma(-t[c], v, val);
            }
            if (lane < nb) {
                if (acc) acc[(size_t)blockIdx.x * n + k0 + lane] += val;
                xv[k0 + lane] = val;
            }
        }
        __syncthreads();
    }
}

// pixelwise parameter map: out[q] = sum over images of gpix[k][q]
__global__ __launch_bounds__(256) void map_sum_kernel(const double* __restrict__ gpix, size_t npx, int O,
                                                      double* __restrict__ out) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= npx) return;
    double s = 0.0;
    for (int k = 0; k < O; ++k) s += gpix[(size_t)k * npx + q];
    out[q] = s;
}

}  // namespace bpltv
