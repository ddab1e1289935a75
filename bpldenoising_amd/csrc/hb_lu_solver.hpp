// hb_lu_solver.hpp -- banded LU without pivoting in HBM, for the one non-symmetric system of the path:
// sumregs_gradient_reg with a patch parameter (/root/reference/src/SumRegsLearningFunction.jl:250),
//     (I + x1[:] .* G1'(B1 - C1)G1 + x2[:] .* G2'(B2 - C2)G2 + x3[:] .* G3'(B3 - C3)G3) \ (ubar - u),
// whose three different ROW scalings cannot be symmetrised by one diagonal similarity (the TV model has a single
// term and is solved symmetrised).  Reached only when the trust-region radius falls below Delta_t = 1e-3.
//
// Block right-looking LU over panels of 128 columns, built from the kernels of the Cholesky path:
//     A = [ I             0 ] [ A11  A12 ]      A11^-1: explicit inverse of the diagonal block (Gauss-Jordan in LDS)
//         [ A21 A11^-1    I ] [ 0    S   ]      L21 = A21 A11^-1 (hb2_trsm_kernel, full right factor)
//                                              S = A22 - L21 A12 (hb2_update_kernel with two operand panels)
// The lower band (with the diagonal) is stored by columns, the upper band by rows ("the lower band of A^T"), so the
// update of the upper part is the same kernel with the operand panels exchanged.  L21 = A21 A11^-1 is DENSE across
// the 128 columns of its panel (the full inverse fills what a triangular factor would keep banded), so the panels
// P are kept as they are (bwp x 128 each, as much memory as the band) and the forward substitution reads them;
// its diagonal blocks are identities.  Backward substitution applies A11^-1 and reads U12 = A12, which stays
// banded, from the row-stored upper band.
// No twisting, one stream: the path is rare and small (128 panels for a 128 x 128 image).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "adjoint_hbm_kernels.hpp"

namespace bpltv {

// bandL[img][c][d] = A[c+d][c], bandU[img][r][d] = A[r][r+d] (d = 0..bw; both carry the diagonal).
__global__ __launch_bounds__(256) void hb_lu_init_kernel(BandDiags DL, BandDiags DU, int bw, int n, double* __restrict__ bandL,
                                                         double* __restrict__ bandU) {
    const int W = bw + 1;
    const int img = blockIdx.y;
    const size_t ib = (size_t)img * n;
    const size_t cnt = (size_t)n * W, stride = (size_t)gridDim.x * 256;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += stride) {
        const long col = (long)(e / W);
        const int d = (int)(e - (size_t)col * W);
        bandL[ib * W + e] = band_entry(DL, ib, n, col, d);
        bandU[ib * W + e] = (d == 0) ? band_entry(DL, ib, n, col, 0) : band_entry(DU, ib, n, col, d);
    }
}

// Inverse of the 128 x 128 diagonal block at k0 by Gauss-Jordan elimination without pivoting, in LDS.
// grid (O), block 1024, dynamic LDS 128 * 129 doubles.  Ainv(r, c) at r + 128 c; AinvT its transpose.
constexpr int LU_T = 1024;
inline size_t hb_getri_lds() { return sizeof(double) * HB2_NB * (HB2_NB + 1); }
__global__ __launch_bounds__(LU_T) void hb_getri_kernel(const double* __restrict__ bandL, const double* __restrict__ bandU, int bw,
                                                        int n, int k0, int npanel, double* __restrict__ Ainv,
                                                        double* __restrict__ AinvT, int* __restrict__ fail) {
    extern __shared__ double S[];
    __shared__ double d0[HB2_NB];   // |diagonal| of the block as loaded: the scale a pivot is compared with
    constexpr int MP = HB2_NB, ld = MP + 1;
    const int W = bw + 1;
    const int img = blockIdx.x, tid = threadIdx.x;
    const double* BL = bandL + (size_t)img * n * W;
    const double* BU = bandU + (size_t)img * n * W;
    for (int e = tid; e < MP * MP; e += LU_T) {
        const int r = e % MP, c = e / MP;
        double x = (r == c) ? 1.0 : 0.0;   // identity padding past the end of the matrix
        if (k0 + r < n && k0 + c < n) {
            const int d = r >= c ? r - c : c - r;
            x = (d <= bw) ? (r >= c ? BL[(size_t)(k0 + c) * W + d] : BU[(size_t)(k0 + r) * W + d]) : 0.0;
        }
        S[r + ld * c] = x;
        if (r == c) d0[r] = fabs(x);
    }
    __syncthreads();
    const int j = tid & (MP - 1), i0 = tid >> 7;   // thread: column j, rows i0 + 8 m
    int badk = -1;
    for (int k = 0; k < MP; ++k) {
        const double piv = S[k + ld * k];
        // no pivoting: a pivot that is zero, not finite, or has lost every digit against the diagonal entry it
        // started from (|piv| < eps |a_kk|) means the elimination has broken down at this column
        if (badk < 0 && !(fabs(piv) > 2.220446049250313e-16 * d0[k] && fabs(piv) < 1.7e308)) badk = k;
        const double pinv = 1.0 / piv;
        const double rk = (j == k) ? pinv : S[k + ld * j] * pinv;
        double nv[16];
#pragma unroll
        for (int mq = 0; mq < 16; ++mq) {
            const int i = i0 + 8 * mq;
            const double f = S[i + ld * k];
            const double old = (j == k) ? 0.0 : S[i + ld * j];
            nv[mq] = (i == k) ? rk : old - f * rk;
        }
        __syncthreads();
#pragma unroll
        for (int mq = 0; mq < 16; ++mq) S[i0 + 8 * mq + ld * j] = nv[mq];
        __syncthreads();
    }
    if (badk >= 0 && tid == 0 && fail[img] == 0) fail[img] = k0 + badk + 1;
    double* Ai = Ainv + ((size_t)img * npanel + k0 / MP) * MP * MP;
    double* AiT = AinvT + ((size_t)img * npanel + k0 / MP) * MP * MP;
    for (int e = tid; e < MP * MP; e += LU_T) {
        const int r = e % MP, c = e / MP;
        Ai[e] = S[r + ld * c];
        AiT[e] = S[c + ld * r];
    }
}

// Q(crel, k) = A(k0 + k, k0 + 128 + crel): the block row right of the panel (U12), transposed into the panel layout
// of P (leading dimension bwp).  grid (ceil(bwp*128/256), O).
__global__ __launch_bounds__(256) void hb_lu_q_kernel(const double* __restrict__ bandU, int bw, int n, int k0, double* __restrict__ Q,
                                                      int bwp) {
    const int W = bw + 1;
    const int img = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= bwp * HB2_NB) return;
    const int c = e % bwp, k = e / bwp;
    const int row = k0 + k, colg = k0 + HB2_NB + c, d = colg - row;
    double v = 0.0;
    if (c < bw && row < n && colg < n && d <= bw) v = bandU[(size_t)img * n * W + (size_t)row * W + d];
    Q[(size_t)img * bwp * HB2_NB + e] = v;
}

// Forward substitution of the block LU: y_blk = x_blk (identity diagonal block), then the bw rows below lose
// P(:, blk) y_blk with the dense panel P of this block.  grid (1 + ceil(bw/128), O), block 1024.
__global__ __launch_bounds__(1024) void hb_lu_fwd_kernel(const double* __restrict__ P, int bw, int n, int k0, int bwp,
                                                         double* __restrict__ x, double* __restrict__ y) {
    __shared__ double yb[HB2_NB];
    __shared__ double red[8 * HB2_NB];
    const int img = blockIdx.y, tid = threadIdx.x;
    double* xv = x + (size_t)img * n;
    if (tid < HB2_NB) yb[tid] = (k0 + tid < n) ? xv[k0 + tid] : 0.0;
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < HB2_NB && k0 + tid < n) y[(size_t)img * n + k0 + tid] = yb[tid];
        return;
    }
    const double* Pi = P + (size_t)img * bwp * HB2_NB;
    const int rr = tid & 127, part = tid >> 7;
    const int rrel = (blockIdx.x - 1) * HB2_NB + rr, R = k0 + HB2_NB + rrel;
    double s = 0.0;
    if (R < n && rrel < bw) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 16 * part + i;
            s = __builtin_fma(Pi[rrel + (size_t)bwp * c], yb[c], s);
        }
    }
    red[part * HB2_NB + rr] = s;
    __syncthreads();
    if (tid < HB2_NB && R < n && rrel < bw) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q * HB2_NB + tid];
        xv[R] -= t;
    }
}

struct HbLuSolver {
    int bw = 0, n = 0, O = 0, npanel = 0, bwp = 0;
    double *bandL = nullptr, *bandU = nullptr, *buf = nullptr;   // buf: Ainv | AinvT ([O][npanel][128^2]) | Q ([O][bwp*128]) | P ([npanel][O][bwp*128])
    int* fail = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    size_t bytes_needed(int bw_, int n_, int O_) const {
        const size_t W = (size_t)bw_ + 1, np_ = (size_t)(n_ + HB2_NB - 1) / HB2_NB, bp = (size_t)(bw_ + 63) / 64 * 64;
        return ((size_t)2 * O_ * n_ * W + (size_t)O_ * (2 * np_ * HB2_NB * HB2_NB + (1 + np_) * bp * HB2_NB)) * sizeof(double);
    }
#define LUCHK(call)                                                                               \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            err = std::string(#call) + " failed: " + hipGetErrorString(e_);                       \
            return e_ == hipErrorOutOfMemory ? 5 : 2;                                             \
        }                                                                                         \
    } while (0)
    int alloc(int bw_, int n_, int O_, hipStream_t st) {
        bw = bw_; n = n_; O = O_; stream = st;
        npanel = (n + HB2_NB - 1) / HB2_NB;
        bwp = (bw + 63) / 64 * 64;
        const size_t W = (size_t)bw + 1;
        LUCHK(hipMalloc((void**)&bandL, (size_t)O * n * W * sizeof(double)));
        LUCHK(hipMalloc((void**)&bandU, (size_t)O * n * W * sizeof(double)));
        LUCHK(hipMalloc((void**)&buf, (size_t)O * (2 * (size_t)npanel * HB2_NB * HB2_NB + (1 + (size_t)npanel) * bwp * HB2_NB) * sizeof(double)));
        LUCHK(hipMalloc((void**)&fail, (size_t)O * sizeof(int)));
        LUCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hb_getri_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)hb_getri_lds()));
        return 0;
    }
    void release() {
        for (void* p : {(void*)bandL, (void*)bandU, (void*)buf, (void*)fail})
            if (p) (void)hipFree(p);
        bandL = bandU = buf = nullptr; fail = nullptr;
    }
    double* Ainv() const { return buf; }
    double* AinvT() const { return buf + (size_t)O * npanel * HB2_NB * HB2_NB; }
    double* Q() const { return AinvT() + (size_t)O * npanel * HB2_NB * HB2_NB; }
    double* P(int k0) const { return Q() + (size_t)O * bwp * HB2_NB * (1 + (size_t)(k0 / HB2_NB)); }   // panel of block k0, [O][bwp*128]

    // DL: diagonals of the lower band (A[c+off][c] at plane[c]); DU: of the upper band (A[r][r+off] at plane[r]).
    int factor(const BandDiags& DL, const BandDiags& DU, int* d_fail_out) {
        const size_t W = (size_t)bw + 1;
        LUCHK(hipMemsetAsync(fail, 0, (size_t)O * sizeof(int), stream));
        const unsigned ib = (unsigned)std::min<size_t>(((size_t)n * W + 255) / 256, 65536);
        hipLaunchKernelGGL(hb_lu_init_kernel, dim3(ib, O), dim3(256), 0, stream, DL, DU, bw, n, bandL, bandU);
        const int nt = (bw + 63) / 64;
        const int g0 = hb2_update_tiles(nt, 0), g1 = hb2_update_tiles(nt, 1), g2 = hb2_update_tiles(nt, 2);
        for (int k0 = 0; k0 < n; k0 += HB2_NB) {
            hipLaunchKernelGGL(hb_getri_kernel, dim3(O), dim3(LU_T), hb_getri_lds(), stream, bandL, bandU, bw, n, k0, npanel, Ainv(),
                               AinvT(), fail);
            if (k0 + HB2_NB >= n) break;
            hipLaunchKernelGGL(hb_lu_q_kernel, dim3((unsigned)((bwp * HB2_NB + 255) / 256), O), dim3(256), 0, stream, bandU, bw, n, k0, Q(),
                               bwp);
            // P = A21 A11^-1: the trsm kernel computes sum_k A21(r, k) X(c, k), so X = (A11^-1)^T
            hipLaunchKernelGGL(hb2_trsm_kernel, dim3(2 * nt * O), dim3(BG_T), 0, stream, bandL, bw, n, k0, npanel, AinvT(), P(k0), bwp, O, 1);
            const int gs[3] = {g0, g1, g2};
            for (int part = 0; part < 3; ++part) {
                if (gs[part] <= 0) continue;
                // lower band (by columns): A(R, C) -= sum_k P(R, k) Q(C, k); no copies (the dense panels are kept)
                hipLaunchKernelGGL(hb2_update_kernel, dim3(gs[part] * O), dim3(BG_T), 0, stream, bandL, bw, n, k0,
                                   (const double*)P(k0), bwp, part == 1 ? 3 : part, O, (const double*)Q(), (const double*)nullptr, 0);
            }
            for (int part = 0; part < 3; ++part) {
                if (gs[part] <= 0) continue;
                // upper band (by rows = lower band of A^T): A^T(C, R) -= sum_k Q(C, k) P(R, k); no copies (U12 = A12 stays)
                hipLaunchKernelGGL(hb2_update_kernel, dim3(gs[part] * O), dim3(BG_T), 0, stream, bandU, bw, n, k0,
                                   (const double*)Q(), bwp, part == 1 ? 3 : part, O, (const double*)P(k0), (const double*)nullptr, 0);
            }
        }
        hipLaunchKernelGGL(hb_fail_merge_kernel, dim3((O + 63) / 64), dim3(64), 0, stream, fail, (const int*)nullptr, 1, O, d_fail_out);
        LUCHK(hipGetLastError());
        return 0;
    }

    // v <- A^-1 v, accv += solution.  scratch: [O][n] doubles.
    void solve(double* v, double* accv, double* scratch) {
        const unsigned chunks = 1 + (unsigned)((bw + HB2_NB - 1) / HB2_NB);
        for (int k0 = 0; k0 < n; k0 += HB2_NB)
            hipLaunchKernelGGL(hb_lu_fwd_kernel, dim3(chunks, O), dim3(1024), 0, stream, (const double*)P(k0), bw, n, k0, bwp, v, scratch);
        for (int k0 = ((n - 1) / HB2_NB) * HB2_NB; k0 >= 0; k0 -= HB2_NB)
            hipLaunchKernelGGL(hb2_bwd_kernel<HB2_NB>, dim3(chunks, O), dim3(BS_T), 0, stream, bandU, Ainv(), bw, n, k0, npanel, scratch, v, accv,
                               0, 0);
    }
#undef LUCHK
};

}  // namespace bpltv
