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
#include "adjoint_bcr_kernels.hpp"

namespace bpltv {

constexpr int HB_NB = 32;   // panel width
constexpr int HB_ROWS = 256; // rows of the panel handled by one workgroup of hb_panel_kernel

// band <- assembled matrix (4 diagonals), zero elsewhere.  grid (nblocks, O), grid-stride over the
// n*W entries of one image (a flat launch over all images would exceed 2^32 work-items at 8 x 1024^2).
__global__ __launch_bounds__(256) void hb_init_kernel(const double* __restrict__ band4, int M, int N, int O,
                                                      double* __restrict__ band) {
    const int W = M + 1;
    const size_t n = (size_t)M * N, tot = n * O;
    const size_t ib = (size_t)blockIdx.y * n;
    const size_t cnt = n * W, stride = (size_t)gridDim.x * 256;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += stride) {
        const size_t col = e / W;
        const int d = (int)(e - col * W);
        band[ib * W + e] = band_init(band4, tot, ib + col, d, M);
    }
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
    // forward substitution of this row against L11.  The row's 32 entries are fetched first (all
    // loads in flight together, coalesced across rows) into LDS, then substituted in place with
    // dynamic loops (no register arrays), then stored.
    double* lrow = &LP[tid][0];
#pragma unroll
    for (int c = 0; c < HB_NB; ++c) {
        const int d = rr - c;
        lrow[c] = (c < nb && d <= bw) ? Bi[(size_t)(k0 + c) * W + d] : 0.0;
    }
    for (int c = 0; c < HB_NB; ++c) {
        // four partial sums + unrolling keep the LDS reads pipelined (a rolled loop waits ~100
        // cycles per read: 512 reads = 20 us per panel)
        double v0 = lrow[c], v1 = 0.0, v2 = 0.0, v3 = 0.0;
        int q = 0;
        for (; q + 4 <= c; q += 4) {
            v0 = __builtin_fma(-lrow[q], T[c][q], v0);
            v1 = __builtin_fma(-lrow[q + 1], T[c][q + 1], v1);
            v2 = __builtin_fma(-lrow[q + 2], T[c][q + 2], v2);
            v3 = __builtin_fma(-lrow[q + 3], T[c][q + 3], v3);
        }
        for (; q < c; ++q) v0 = __builtin_fma(-lrow[q], T[c][q], v0);
        lrow[c] = ((v0 + v1) + (v2 + v3)) * Dinv[c];
    }
#pragma unroll
    for (int c = 0; c < HB_NB; ++c) {
        const int d = rr - c;
        if (c < nb && d <= bw) Bi[(size_t)(k0 + c) * W + d] = lrow[c];
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

// ---- 128-column panels on the f64 MFMA (hb2_*) -------------------------------------------------------
// The 32-column panel kernels above cost ~50 us per panel whatever the arithmetic (kernel boundaries +
// a sequential 32x32 Cholesky + scalar updates): 1.6 us per column.  With 128-column panels the same
// three steps are the dense kernels of the block cyclic reduction, addressed into the band (the lower
// band is a column-major matrix with leading dimension W-1: A(r, c) = band[r + (W-1) c]):
//   hb2_potrf_kernel   diagonal block: Cholesky + inverse in LDS (bcr_potrf_lds_body)
//   hb2_trsm_kernel    P = A21 L11^-T for the bw rows below (MFMA tiles; side panel buffer P)
//   hb2_update_kernel  A22 -= P P^T on the lower 64x64 tiles of the trailing bw x bw block (MFMA),
//                      and L11, P copied into the band
// Side buffers per image: Linv11 (128 x 128), L11 (128 x 128), P (bwp x 128, bwp = bw rounded up to 64).
constexpr int HB2_NB = 128;

// grid (O), block BCR_PT, dynamic LDS bcr_potrf_lds(HB2_NB)
__global__ __launch_bounds__(BCR_PT) void hb2_potrf_kernel(const double* __restrict__ band, int M, int N, int k0,
                                                           double* __restrict__ Linv, double* __restrict__ L11,
                                                           int* __restrict__ fail) {
    extern __shared__ double S[];
    constexpr int MP = HB2_NB, ld = MP + 1;
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const double* Bi = band + (size_t)img * n * W + (size_t)k0 * W;   // A(k0+r, k0+c) = Bi[r + (W-1) c], r >= c
    for (int e = tid; e < MP * MP; e += BCR_PT) {
        const int r = e % MP, c = e / MP;
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        double v = (r == c) ? 1.0 : 0.0;   // identity padding past the end of the matrix
        if (k0 + hi < n) v = (hi - lo <= bw) ? Bi[hi + (size_t)(W - 1) * lo] : 0.0;
        S[r + ld * c] = v;
    }
    __syncthreads();
    double* Lg = L11 + (size_t)img * MP * MP;
    const bool bad = bcr_potrf_lds_body(S, MP, Lg, MP);
    if (bad && lane == 0 && fail[img] == 0) fail[img] = k0 + 1;
    double* Li = Linv + (size_t)img * MP * MP;
    for (int e = tid; e < MP * MP; e += BCR_PT) {
        const int r = e % MP, c = e / MP;
        Li[e] = (r >= c) ? S[(16 * (c >> 4) + (r & 15)) + ld * (16 * (r >> 4) + (c & 15))] : 0.0;
        if ((r >> 4) > (c >> 4)) Lg[e] = S[r + ld * c];          // strictly lower tiles of L
        else if ((r >> 4) < (c >> 4)) Lg[e] = 0.0;               // (diagonal tiles were written by bcr_diag_tile)
    }
}

// 64 x 32 chunk of the band as an MFMA operand: rows R0 + (tid & 63), columns K0 + (tid >> 6) + 4 i.
__device__ __forceinline__ void hb2_fetch_band(const double* __restrict__ Bi, int W, int bw, int n, int R0, int K0,
                                               int tid, double (&v)[8]) {
    const int R = R0 + (tid & 63), kq = K0 + (tid >> 6);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int K = kq + 4 * i;
        v[i] = (R < n && K < n && R >= K && R - K <= bw) ? Bi[(size_t)K * W + (R - K)] : 0.0;
    }
}
// same from a dense column-major array with leading dimension ld (rows < rmax, columns < kmax)
__device__ __forceinline__ void hb2_fetch_dense(const double* __restrict__ base, int ld, int rmax, int kmax, int r0,
                                                int k0, int tid, double (&v)[8]) {
    const int r = r0 + (tid & 63), kq = k0 + (tid >> 6);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kq + 4 * i;
        v[i] = (r < rmax && k < kmax) ? base[r + (size_t)ld * k] : 0.0;
    }
}
// one staged chunk (both operands row-fast): 8 k-steps of 2x2 MFMA tiles per wave
__device__ __forceinline__ void hb2_mma_chunk(const double* __restrict__ As, const double* __restrict__ Bs, BgAcc& acc) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
#pragma unroll
    for (int kk = 0; kk < BG_KC / 4; ++kk) {
        const int ko = (4 * kk + lk) * BG_LD;
        const double a0 = As[ko + wr + lr], a1 = As[ko + wr + 16 + lr];
        const double b0 = Bs[ko + wc + lr], b1 = Bs[ko + wc + 16 + lr];
        acc.c[0][0] = bcr_mfma(a0, b0, acc.c[0][0]);
        acc.c[0][1] = bcr_mfma(a0, b1, acc.c[0][1]);
        acc.c[1][0] = bcr_mfma(a1, b0, acc.c[1][0]);
        acc.c[1][1] = bcr_mfma(a1, b1, acc.c[1][1]);
    }
}

// P(rrel, c) = sum_k A(k0+128+rrel, k0+k) Linv11(c, k).  grid (ceil(bw/64) * 2, O), block BG_T.
__global__ __launch_bounds__(BG_T) void hb2_trsm_kernel(const double* __restrict__ band, int M, int N, int k0,
                                                        const double* __restrict__ Linv, double* __restrict__ P,
                                                        int bwp) {
    __shared__ double lds[BG_LDS];
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.y, tid = threadIdx.x;
    const int rt = blockIdx.x >> 1, c0 = (blockIdx.x & 1) * 64;
    const int R0 = k0 + HB2_NB + 64 * rt;
    if (R0 >= n) return;
    const double* Bi = band + (size_t)img * n * W;
    const double* Li = Linv + (size_t)img * HB2_NB * HB2_NB;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    const int nchunk = (c0 + 64) / BG_KC;   // Linv11(c, k) = 0 for k > c
    double va[8], vb[8];
    hb2_fetch_band(Bi, W, bw, n, R0, k0, tid, va);
    hb2_fetch_dense(Li, HB2_NB, HB2_NB, HB2_NB, c0, 0, tid, vb);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        bg_stage<true>(As, tid, va);
        bg_stage<true>(Bs, tid, vb);
        __syncthreads();
        if (ch + 1 < nchunk) {
            hb2_fetch_band(Bi, W, bw, n, R0, k0 + (ch + 1) * BG_KC, tid, va);
            hb2_fetch_dense(Li, HB2_NB, HB2_NB, HB2_NB, c0, (ch + 1) * BG_KC, tid, vb);
        }
        hb2_mma_chunk(As, Bs, acc);
    }
    bg_to_lds(acc, lds);
    double* Pi = P + (size_t)img * bwp * HB2_NB;
    const int l = tid & 63, hq = tid >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int o = hq + 4 * i;
        Pi[(64 * rt + l) + (size_t)bwp * (c0 + o)] = lds[o * BG_LD + l];
    }
}

// Trailing update A(R, C) -= sum_k P(R, k) P(C, k) on the lower 64x64 tiles (ta >= tb) of the bw x bw block
// behind the panel; the diagonal tiles also copy their 64 rows of P (= L21) into the band and tile 0
// copies L11.  grid (nt (nt+1) / 2, O), nt = ceil(bw/64); block BG_T.
__global__ __launch_bounds__(BG_T) void hb2_update_kernel(double* __restrict__ band, int M, int N, int k0,
                                                          const double* __restrict__ L11, const double* __restrict__ P,
                                                          int bwp) {
    __shared__ double lds[BG_LDS];
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.y, tid = threadIdx.x;
    double* Bi = band + (size_t)img * n * W;
    const double* Pi = P + (size_t)img * bwp * HB2_NB;
    int ta = 0, t = blockIdx.x;
    while (t > ta) { t -= ta + 1; ++ta; }
    const int tb = t;   // tb <= ta
    if (blockIdx.x == 0) {   // L11 of this panel into the band
        const double* Lg = L11 + (size_t)img * HB2_NB * HB2_NB;
        for (int e = tid; e < HB2_NB * HB2_NB; e += BG_T) {
            const int r = e % HB2_NB, c = e / HB2_NB;
            if (r >= c && k0 + r < n && r - c <= bw) Bi[(size_t)(k0 + c) * W + (r - c)] = Lg[e];
        }
    }
    const int base = k0 + HB2_NB;
    if (base + 64 * tb >= n) return;
    if (ta == tb) {          // rows [64 ta, 64 ta + 64) of L21 into the band
        for (int e = tid; e < 64 * HB2_NB; e += BG_T) {
            const int rr = 64 * ta + (e & 63), c = e >> 6;
            const int R = base + rr, K = k0 + c;
            if (R < n && K < n && R - K <= bw) Bi[(size_t)K * W + (R - K)] = Pi[rr + (size_t)bwp * c];
        }
    }
    if (base + 64 * ta >= n) return;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    double va[8], vb[8];
    hb2_fetch_dense(Pi, bwp, bwp, HB2_NB, 64 * ta, 0, tid, va);
    hb2_fetch_dense(Pi, bwp, bwp, HB2_NB, 64 * tb, 0, tid, vb);
    constexpr int nchunk = HB2_NB / BG_KC;
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        bg_stage<true>(As, tid, va);
        bg_stage<true>(Bs, tid, vb);
        __syncthreads();
        if (ch + 1 < nchunk) {
            hb2_fetch_dense(Pi, bwp, bwp, HB2_NB, 64 * ta, (ch + 1) * BG_KC, tid, va);
            hb2_fetch_dense(Pi, bwp, bwp, HB2_NB, 64 * tb, (ch + 1) * BG_KC, tid, vb);
        }
        hb2_mma_chunk(As, Bs, acc);
    }
    bg_to_lds(acc, lds);
    const int l = tid & 63, hq = tid >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int o = hq + 4 * i;
        const int R = base + 64 * ta + l, Cc = base + 64 * tb + o;
        if (R < n && Cc < n && R >= Cc && R - Cc <= bw && 64 * ta + l < bw && 64 * tb + o < bw)
            Bi[(size_t)Cc * W + (R - Cc)] -= lds[o * BG_LD + l];
    }
}

// Substitutions with L in the band array.  A single workgroup streaming the 8.6 GB factor of a
// 1024^2 image is limited to one CU's memory bandwidth (~24 GB/s), so the work of every 64-column
// block is spread over 1 + ceil(bw/64) one-wave workgroups and the kernel boundary is again the
// grid-wide synchronisation: one launch per block.  Every workgroup recomputes the block's solution
// (64x64 matrix-vector product with the inverted diagonal block of adj_invdiag_kernel, operands
// broadcast by v_readlane); workgroup 0 stores it, workgroup 1+j applies it to its 64 rows.
// Forward (L y = b):   in/out `x` = running right-hand side, solution rows -> `y`.
// Backward (L^T x = y): in/out `y` = running right-hand side, solution rows -> `x` (and += acc).
__device__ __forceinline__ double hb_block_matvec(const double* __restrict__ blk, double v, int lane) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        double xr[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) xr[c] = blk[(hh * 32 + c) * SB + lane];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            a0 = __builtin_fma(xr[c], readlane_f64(v, hh * 32 + c), a0);
            a1 = __builtin_fma(xr[c + 1], readlane_f64(v, hh * 32 + c + 1), a1);
            a2 = __builtin_fma(xr[c + 2], readlane_f64(v, hh * 32 + c + 2), a2);
            a3 = __builtin_fma(xr[c + 3], readlane_f64(v, hh * 32 + c + 3), a3);
        }
    }
    return (a0 + a1) + (a2 + a3);
}

// grid (1 + ceil(bw/64), O), block 64
__global__ __launch_bounds__(64) void hb_fwd_block_kernel(const double* __restrict__ band,
                                                          const double* __restrict__ invF, int M, int N, int k0,
                                                          double* __restrict__ x, double* __restrict__ y) {
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.y, lane = threadIdx.x;
    const double* Bi = band + (size_t)img * n * W;
    double* xv = x + (size_t)img * n;
    const int nblk = (n + SB - 1) / SB;
    const double* blk = invF + ((size_t)img * nblk + k0 / SB) * SB * SB;
    const double bv = (k0 + lane < n) ? xv[k0 + lane] : 0.0;
    const double val = hb_block_matvec(blk, bv, lane);  // X[lane][:] . b_blk
    if (blockIdx.x == 0) {
        if (k0 + lane < n) y[(size_t)img * n + k0 + lane] = val;
        return;
    }
    const int r = k0 + SB + (blockIdx.x - 1) * SB + lane;
    if (r >= n || r > k0 + SB - 1 + bw) return;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 16
    for (int c = 0; c < SB; c += 2) {
        const int d0 = r - (k0 + c), d1 = d0 - 1;
        const double l0 = (d0 <= bw && k0 + c < n) ? Bi[(size_t)(k0 + c) * W + d0] : 0.0;
        const double l1 = (d1 <= bw && k0 + c + 1 < n) ? Bi[(size_t)(k0 + c + 1) * W + d1] : 0.0;
        s0 = __builtin_fma(l0, readlane_f64(val, c), s0);
        s1 = __builtin_fma(l1, readlane_f64(val, c + 1), s1);
    }
    xv[r] -= s0 + s1;
}

__global__ __launch_bounds__(64) void hb_bwd_block_kernel(const double* __restrict__ band,
                                                          const double* __restrict__ invB, int M, int N, int k0,
                                                          double* __restrict__ y, double* __restrict__ x,
                                                          double* __restrict__ acc) {
    const int W = M + 1, bw = M, n = M * N;
    const int img = blockIdx.y, lane = threadIdx.x;
    const double* Bi = band + (size_t)img * n * W;
    double* yv = y + (size_t)img * n;
    const int nblk = (n + SB - 1) / SB;
    const double* blk = invB + ((size_t)img * nblk + k0 / SB) * SB * SB;
    const double zv = (k0 + lane < n) ? yv[k0 + lane] : 0.0;
    const double val = hb_block_matvec(blk, zv, lane);  // X[:][lane] . z_blk
    if (blockIdx.x == 0) {
        if (k0 + lane < n) {
            x[(size_t)img * n + k0 + lane] = val;
            if (acc) acc[(size_t)img * n + k0 + lane] += val;
        }
        return;
    }
    // earlier equation k: y_k -= sum_c L[k0+c][k] x_{k0+c},  L[k0+c][k] = band[k][k0 + c - k]
    const int k = k0 - 1 - ((blockIdx.x - 1) * SB + lane);
    if (k < 0 || k0 - k > bw) return;
    const double* col = Bi + (size_t)k * W + (k0 - k);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 16
    for (int c = 0; c < SB; c += 2) {
        const double l0 = (k0 - k + c <= bw && k0 + c < n) ? col[c] : 0.0;
        const double l1 = (k0 - k + c + 1 <= bw && k0 + c + 1 < n) ? col[c + 1] : 0.0;
        s0 = __builtin_fma(l0, readlane_f64(val, c), s0);
        s1 = __builtin_fma(l1, readlane_f64(val, c + 1), s1);
    }
    yv[k] -= s0 + s1;
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
