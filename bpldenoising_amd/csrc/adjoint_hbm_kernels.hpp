// adjoint_hbm_kernels.hpp -- adjoint solve for images too wide for the LDS window (M > 138).
//
// Same reduced SPD system and same pipeline as adjoint_kernels.hpp, but the band lives in HBM:
// band[O][n][W], W = M+1 (column k of the lower band: entry (k+d, k) at d), factored IN PLACE.  The
// lower band is also a column-major matrix with leading dimension W-1, A(r, c) = band[r + (W-1) c], so
// dense tile kernels address it directly.  Grid-wide synchronisation is the kernel boundary: per panel
// of 128 columns a few launches (all images in the same launches), built from the dense kernels of the
// block cyclic reduction (adjoint_bcr_kernels.hpp):
//   hb3_chain_kernel   diagonal block: the previous panel's contribution (two 128^3 products on the MFMA, in
//                      LDS), Cholesky + inverse in LDS (bcr_potrf_lds_body); L11^-1 and its transpose kept per
//                      panel (the panel solve and the substitutions use them; L11 itself is not stored)
//   hb2_trsm_kernel    P = A21 L11^-T for the bw rows below (MFMA tiles; side panel buffer P)
//   hb2_update_kernel  A22 -= P P^T on the lower 64x64 tiles of the trailing bw x bw block (MFMA),
//                      and P = L21 copied into the band
// and per panel one launch of hb2_fwd_kernel / hb2_bwd_kernel per substitution.  (A first version with
// 32-column panels, a register Cholesky, scalar updates and 64-column substitutions cost 3.5 s per
// 8 x 1024^2 gradient; these kernels 1.25 s launched one after the other, 0.54 s as the twisted three-stream
// pipeline of hb_band_solver.hpp)
#pragma once
#include <hip/hip_runtime.h>
#include "adjoint_kernels.hpp"
#include "adjoint_bcr_kernels.hpp"

namespace bpltv {

// The assembled matrix as a few diagonals of its lower band: plane t holds A[c + off[t]][c] at
// planes[t * tot + img * n + c] (TV: offsets 0, 1, M-1, M = adj_assemble_kernel's band4; the sum-of-regularisers
// model has seven).  Offsets may coincide for tiny M: coinciding planes add.
struct BandDiags {
    const double* planes;
    size_t tot;   // doubles per plane (= O * n)
    int nd;
    int off[8];
};
__device__ __forceinline__ double band_entry(const BandDiags& D, size_t ib, int n, long c, int d) {
    if (c < 0 || c + d >= n) return 0.0;
    double v = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < D.nd && d == D.off[t]) v += D.planes[(size_t)t * D.tot + ib + c];
    return v;
}

// Twisted (two-sided) factorisation of the band: the column range splits into top [0, m), middle [m, m + nm)
// and bottom [m + nm, n).  Problem (img, side) of `sides` per image is a banded matrix of np = m + bw rows whose
// first m columns are eliminated: side 0 = the leading principal submatrix of A, side 1 = the trailing one with
// its index reversed (local k' = global n-1-k'); both leave their Schur complement on the middle in their last
// bw columns.  sides = 1: the whole matrix (np = n).
// band[p][np][W] <- that problem's matrix, zero elsewhere.  grid (nblocks, O * sides), grid-stride over np * W
// entries (a flat launch over all images would exceed 2^32 work-items at 8 x 1024^2).
__global__ __launch_bounds__(256) void hb_init_kernel(BandDiags D, int bw, int n, int sides, int np,
                                                      double* __restrict__ band) {
    const int W = bw + 1;
    const int p = blockIdx.y, img = p / sides, side = p - img * sides;
    const size_t ib = (size_t)img * n;
    double* Bp = band + (size_t)p * np * W;
    const size_t cnt = (size_t)np * W, stride = (size_t)gridDim.x * 256;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += stride) {
        const long col = (long)(e / W);
        const int d = (int)(e - (size_t)col * W);
        double v = 0.0;
        if (col + d < np) v = (side == 0) ? band_entry(D, ib, n, col, d) : band_entry(D, ib, n, (long)n - 1 - col - d, d);
        Bp[e] = v;
    }
}

// The same on a band that is already zero: only the nd diagonals of every column.  grid (nblocks, O * sides).
__global__ __launch_bounds__(256) void hb_init_diag_kernel(BandDiags D, int bw, int n, int sides, int np,
                                                           double* __restrict__ band) {
    const int W = bw + 1;
    const int p = blockIdx.y, img = p / sides, side = p - img * sides;
    const size_t ib = (size_t)img * n;
    double* Bp = band + (size_t)p * np * W;
    for (size_t col = (size_t)blockIdx.x * 256 + threadIdx.x; col < (size_t)np; col += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= D.nd) break;
            const int d = D.off[t];
            bool dup = false;   // coinciding offsets (tiny M): band_entry already adds them; write once
            for (int t2 = 0; t2 < t; ++t2) dup |= D.off[t2] == d;
            if (dup || d > bw || (long)col + d >= np) continue;
            Bp[col * W + d] = (side == 0) ? band_entry(D, ib, n, (long)col, d) : band_entry(D, ib, n, (long)n - 1 - (long)col - d, d);
        }
    }
}

// Middle block of the twisted factorisation: S = A_mid - (A_mid - S_top) - (A_mid - S_bot) with the two side
// problems' trailing windows.  mid[img][nm][W] (a banded problem of nm rows, bandwidth min(bw, nm-1)).
__global__ __launch_bounds__(256) void hb_mid_gather_kernel(BandDiags D, int bw, int n, int m, int nm, int np,
                                                            const double* __restrict__ side, double* __restrict__ mid) {
    const int W = bw + 1;
    const int img = blockIdx.y;
    const size_t ib = (size_t)img * n;
    const double* top = side + (size_t)(2 * img) * np * W;
    const double* bot = side + (size_t)(2 * img + 1) * np * W;
    double* Mi = mid + (size_t)img * nm * W;
    const size_t cnt = (size_t)nm * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e / W), d = (int)(e - (size_t)c * W), r = c + d;
        double v = 0.0;
        if (r < nm) {
            const double a = band_entry(D, ib, n, (long)m + c, d);
            v = a;
            if (r < bw) v -= a - top[(size_t)(m + c) * W + d];
            if (c >= nm - bw) v -= a - bot[(size_t)(n - 1 - m - r) * W + d];
        }
        Mi[e] = v;
    }
}

// Right-hand side / solution vectors of the twisted solve.  vs[p][np]: side problems, vm[img][nm]: middle.
// mode 0: vs <- vec (top: vec[k'], bottom: vec[n-1-k']).
// mode 1: vm <- middle right-hand side from the side vectors' windows after the forward sweeps:
//         b_m - (b_m - vs_top[m + r]) - (b_m - vs_bot[n-1-m-r]) on their supports.
// mode 2: side windows <- middle solution xm (rows m .. np-1 of xs), before the backward sweeps.
// mode 3: vec <- solution (top xs | xm | bottom xs reversed); acc += solution when acc != nullptr.
__global__ __launch_bounds__(256) void hb_tw_vec_kernel(int mode, int bw, int n, int m, int nm, int np,
                                                        double* __restrict__ vec, double* __restrict__ vs,
                                                        double* __restrict__ xs, double* __restrict__ vm,
                                                        double* __restrict__ acc) {
    const int img = blockIdx.y;
    double* v = vec + (size_t)img * n;
    double* st = vs + (size_t)(2 * img) * np;
    double* sb = vs + (size_t)(2 * img + 1) * np;
    double* xt = xs + (size_t)(2 * img) * np;
    double* xb = xs + (size_t)(2 * img + 1) * np;
    double* mm = vm + (size_t)img * nm;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (mode == 0) {
        if (g < np) { st[g] = v[g]; sb[g] = v[n - 1 - g]; }
    } else if (mode == 1) {
        if (g < nm) {
            const double b = v[m + g];
            double r = b;
            if (g < bw) r -= b - st[m + g];
            if (g >= nm - bw) r -= b - sb[n - 1 - m - g];
            mm[g] = r;
        }
    } else if (mode == 2) {
        if (g < nm) {
            if (g < bw) xt[m + g] = mm[g];
            if (g >= nm - bw) xb[n - 1 - m - g] = mm[g];
        }
    } else {
        if (g < n) {
            const double x = (g < m) ? xt[g] : ((g < m + nm) ? mm[g - m] : xb[n - 1 - g]);
            v[g] = x;
            if (acc) acc[(size_t)img * n + g] += x;
        }
    }
}

constexpr int HB2_NB = 128;

// Cholesky factor + inverse of the block in S (bcr_potrf_lds_body); L11^-1 and its transpose are kept per panel:
// the triangular solve of the panel, the next diagonal block and the substitutions all work with them, so the
// Cholesky tile L11 itself is never stored (no kernel reads the diagonal blocks of L from the band).  Only the
// nonzero triangle of each is written: Linv / LinvT are zero-filled once when the solver is allocated.
__device__ __forceinline__ void hb2_potrf_finish(double* __restrict__ S, int img, int k0, int npanel,
                                                 double* __restrict__ Linv, double* __restrict__ LinvT,
                                                 int* __restrict__ fail) {
    constexpr int MP = HB2_NB, ld = MP + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    BCR_PROBE(4);
    const bool bad = bcr_potrf_lds_body(S, MP);
    BCR_PROBE(5);
    if (bad && lane == 0 && fail[img] == 0) fail[img] = k0 + 1;
    double* Li = Linv + ((size_t)img * npanel + k0 / MP) * MP * MP;
    double* LiT = LinvT + ((size_t)img * npanel + k0 / MP) * MP * MP;
    // entry e = r + 128 c of either array is W(hi, lo), hi = max(r, c): one LDS read per entry, in batches of 8 reads
    // followed by their stores (a rolled loop pays the LDS latency 32 times)
    for (int e0 = tid; e0 < MP * MP; e0 += 8 * BCR_PT) {
        double w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * BCR_PT, r = e % MP, c = e / MP;
            const int hi = r > c ? r : c, lo = r > c ? c : r;
            w[i] = S[(16 * (lo >> 4) + (hi & 15)) + ld * (16 * (hi >> 4) + (lo & 15))];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * BCR_PT, r = e % MP, c = e / MP;
            if (r >= c) Li[e] = w[i];
            if (c >= r) LiT[e] = w[i];
        }
    }
    BCR_PROBE(6);
    BCR_SPAN_END(0, k0 / MP);
}

#ifdef BCR_PROBE_ON
__device__ int bcr_probe_panel = -1;   // tools/potrf_probe.hip: stamp the chain kernel of this panel only (-1: every one)
#endif
// The dependent chain of the banded Cholesky in ONE launch per panel (one workgroup per problem): the diagonal
// block k receives the contribution of panel k-1 here, in LDS, and is factored at once --
//     P0 = A(k, k-1) L(k-1,k-1)^-T            (the first 128 rows of panel k-1; its band entries are final once
//                                              panel k-2's update is done)
//     D  = A(k, k) - P0 P0^T                  (band entries updated through panel k-2)
//     L(k,k) = chol(D), L(k,k)^-1
// so that neither the triangular solve of the whole panel (hb2_trsm_kernel) nor a separate update launch sits
// between two diagonal blocks: those run behind, on other streams (HbBandSolver::factor_problems).  The band copy of
// A(k, k) is NOT updated with panel k-1's part (nothing else reads it before L(k,k) overwrites it).
// k0 = 0: plain diagonal block.  grid (nprob), block BCR_PT, dynamic LDS bcr_potrf_lds(HB2_NB) -- the same 129 x 128
// array holds A(k, k-1), then P0, then D.
__global__ __launch_bounds__(BCR_PT) void hb3_chain_kernel(const double* __restrict__ band, int bw, int n, int k0,
                                                           int npanel, double* __restrict__ Linv,
                                                           double* __restrict__ LinvT, int* __restrict__ fail) {
    extern __shared__ double S[];
    constexpr int MP = HB2_NB, ld = MP + 1;
    const int W = bw + 1;
    const int img = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    BCR_SPAN_RESET(0, k0 / MP);
    BCR_SPAN_BEGIN(0, k0 / MP);
#ifdef BCR_PROBE_ON
    if (tid == 0 && blockIdx.x == 0) bcr_probe_on = (bcr_probe_panel < 0 || k0 / MP == bcr_probe_panel) ? 1 : 0;
    __syncthreads();
#endif
    const double* Bi = band + (size_t)img * n * W;
    const double* Bd = Bi + (size_t)k0 * W;                   // A(k0+r, k0+c) = Bd[r + (W-1) c], r >= c
    if (k0 == 0) {
        for (int e0 = tid; e0 < MP * MP; e0 += 8 * BCR_PT) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * BCR_PT, r = e % MP, c = e / MP;
                const int lo = r < c ? r : c, hi = r < c ? c : r;
                const bool in = (hi < n) && (hi - lo <= bw);
                v[i] = Bd[in ? hi + (size_t)(W - 1) * lo : 0];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * BCR_PT, r = e % MP, c = e / MP;
                const int lo = r < c ? r : c, hi = r < c ? c : r;
                double x = (r == c) ? 1.0 : 0.0;
                if (hi < n) x = (hi - lo <= bw) ? v[i] : 0.0;
                S[r + ld * c] = x;
            }
        }
        __syncthreads();
        hb2_potrf_finish(S, img, k0, npanel, Linv, LinvT, fail);
        return;
    }
    BCR_PROBE(0);
    // Every global operand is requested here, before anything waits: A(k, k-1) (32 entries per thread), this wave's
    // tiles of D and its rows of Linv(k-1) land together after one memory latency.
    // A(k0 + r, k0 - 128 + c), e = r + 128 c
    double av[MP * MP / BCR_PT];
    {
        const double* Bp = Bi + (size_t)(k0 - MP) * W;        // column k0-128+c at Bp + c W, row k0+r at offset 128 + r - c
#pragma unroll
        for (int i = 0; i < MP * MP / BCR_PT; ++i) {
            const int e = tid + i * BCR_PT, r = e % MP, c = e / MP, off = MP + r - c;
            const bool in = (k0 + r < n) && (off <= bw);
            av[i] = in ? Bp[(size_t)c * W + off] : 0.0;
        }
    }
    // P0 column tile jw of this wave (waves w and w+4 share a SIMD: jw and 7-jw balance its MFMA work); its B
    // operands Linv(16 jw + lr, k), k < 16 (jw + 1), in MFMA order: 16 consecutive doubles per k
    constexpr int NT = MP / 16;
    const int jw = wave < 4 ? wave : 11 - wave;
    double lb[4 * NT];
    {
        const double* Lp = Linv + ((size_t)img * npanel + (k0 / MP - 1)) * MP * MP;
#pragma unroll
        for (int k4 = 0; k4 < 4 * NT; ++k4) {
            const int kq = 4 * k4 + lk;
            lb[k4] = (k4 < 4 * (jw + 1)) ? Lp[(16 * jw + lr) + (size_t)MP * kq] : 0.0;
        }
    }
    // D(r, c), r >= c, e = r + 128 c: whole columns of the band (the MFMA accumulator layout would gather 16 lines
    // per load); added to -P0 P0^T in LDS afterwards
    double dl[MP * MP / BCR_PT];
#pragma unroll
    for (int i = 0; i < MP * MP / BCR_PT; ++i) {
        const int e = tid + i * BCR_PT, r = e % MP, c = e / MP;
        double x = (r == c) ? 1.0 : 0.0;   // identity padding past the end of the matrix
        if (r >= c && k0 + r < n) x = (r - c <= bw) ? Bd[r + (size_t)(W - 1) * c] : 0.0;
        dl[i] = x;
    }
    // lower 16x16 tiles of D, dealt round-robin over the 8 waves: tile t -> (I, J), I >= J
    constexpr int NTRI = NT * (NT + 1) / 2, TPW = (NTRI + BCR_PT / 64 - 1) / (BCR_PT / 64);
    int tI[TPW], tJ[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        int t = wave + (BCR_PT / 64) * q, a = 0;
        const bool have = t < NTRI;
        if (!have) t = 0;
        while (t > a) { t -= a + 1; ++a; }
        tI[q] = have ? a : -1; tJ[q] = t;
    }
#pragma unroll
    for (int i = 0; i < MP * MP / BCR_PT; ++i) {
        const int e = tid + i * BCR_PT;
        S[(e % MP) + ld * (e / MP)] = av[i];
    }
    __syncthreads();
    BCR_PROBE(1);
    // P0 tile (i, jw) = sum over k-blocks kb <= jw of A(i, kb) Linv(jw, kb)^T for the 8 row tiles i
    {
        bcr_d4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = bcr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k4 = 0; k4 < 4 * NT; ++k4) {
            if (k4 < 4 * (jw + 1)) {   // wave-uniform
                const int kq = 4 * k4 + lk;
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = bcr_mfma(S[(16 * i + lr) + ld * kq], lb[k4], acc[i]);
            }
        }
        __syncthreads();   // every wave is done with A(k, k-1)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * jw + lr)] = acc[i][g];
    }
    __syncthreads();
    BCR_PROBE(2);
    // -P0 P0^T, tile (I, J): sum_k P0(16 I + r, k) P0(16 J + c, k)
    bcr_d4 dacc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        dacc[q] = bcr_d4{0.0, 0.0, 0.0, 0.0};
        if (tI[q] < 0) continue;
        const int ra = 16 * tI[q] + lr, rb = 16 * tJ[q] + lr;
        bcr_d4 acc = dacc[q];
#pragma unroll 8
        for (int k4 = 0; k4 < MP / 4; ++k4) {
            const int kq = 4 * k4 + lk;
            acc = bcr_mfma(-S[ra + ld * kq], S[rb + ld * kq], acc);
        }
        dacc[q] = acc;
    }
    __syncthreads();       // every wave is done with P0
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        if (tI[q] < 0) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) S[(16 * tI[q] + lk + 4 * g) + ld * (16 * tJ[q] + lr)] = dacc[q][g];
    }
    __syncthreads();
    BCR_PROBE(3);
    // D = A(k, k) - P0 P0^T: the owner of (r, c), r >= c, adds the band entry
#pragma unroll
    for (int i = 0; i < MP * MP / BCR_PT; ++i) {
        const int e = tid + i * BCR_PT, r = e % MP, c = e / MP;
        if (r >= c) {
            S[r + ld * c] += dl[i];   // (the Cholesky reads the lower triangle only)
        }
    }
    __syncthreads();
    hb2_potrf_finish(S, img, k0, npanel, Linv, LinvT, fail);
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

// P(rrel, c) = sum_k A(k0+128+rrel, k0+k) Linv11(c, k)   (full != 0: Linv11 is a full matrix -- the LU path passes
// the transposed inverse of the diagonal block, so that P = A21 A11^-1).  grid (ceil(bw/64) * 2 * nprob), id = tile * nprob + problem; block BG_T.
__global__ __launch_bounds__(BG_T) __attribute__((amdgpu_waves_per_eu(4))) void hb2_trsm_kernel(const double* __restrict__ band, int bw, int n, int k0,
                                                        int npanel, const double* __restrict__ Linv,
                                                        double* __restrict__ P, int bwp, int nprob, int full) {
    __shared__ double lds[BG_LDS];
    const int W = bw + 1;
    const int img = (int)(blockIdx.x % (unsigned)nprob), tid = threadIdx.x;
    const int bxt = (int)(blockIdx.x / (unsigned)nprob);
    const int rt = bxt >> 1, c0 = (bxt & 1) * 64;
    const int R0 = k0 + HB2_NB + 64 * rt;
    BCR_SPAN_RESET(1, k0 / HB2_NB);
    BCR_SPAN_BEGIN(1, k0 / HB2_NB);
    if (R0 >= n) return;
    const double* Bi = band + (size_t)img * n * W;
    const double* Li = Linv + ((size_t)img * npanel + k0 / HB2_NB) * HB2_NB * HB2_NB;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    const int nchunk = full ? HB2_NB / BG_KC : (c0 + 64) / BG_KC;   // Cholesky: Linv11(c, k) = 0 for k > c
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
    BCR_SPAN_END(1, k0 / HB2_NB);
}

// Trailing update A(R, C) -= sum_k P(R, k) P(C, k) on the lower 64x64 tiles (ta >= tb) of the bw x bw block
// behind the panel, in three launches per panel (nt = ceil(bw/64) tile rows):
//   part 0  the three tiles of the next panel's diagonal block;
//   part 1  what the next panel's hb2_trsm_kernel and first update read or overwrite: the rest of the first
//           128-column block (tiles (ta, 0), (ta, 1), ta >= 2) and the tiles (2,2), (3,2), (3,3) of the diagonal
//           block after next; its workgroups also copy a finished panel P = L21 into the band, which only the
//           substitutions read;
//   part 2  everything else (tb >= 2), which only has to precede the same tiles' update by the next panel;
//   part 5  no tile: the copies only.
// Which launches a panel gets and on which streams: HbBandSolver::factor_problems (Cholesky; part 0 only for the last
// panel, the others' diagonal blocks are updated inside hb3_chain_kernel) and HbLuSolver::factor (all parts in order).
// grid (hb2_update_tiles(nt, part) * nprob), workgroup id = tile * nprob + problem: ids of equal residue mod 8
// share an XCD (observed round-robin placement; speed only), so with nprob a multiple of 8 every L2 serves the
// 1 MB panels P of nprob / 8 problems instead of all of them; block BG_T.
__host__ __device__ inline int hb2_update_tiles(int nt, int part) {
    const int nA0 = nt > 2 ? nt - 2 : 0, ntr = nt * (nt + 1) / 2;
    const int first = ntr < 3 ? ntr : 3;
    const int nd = (nt >= 4) ? 3 : ((nt == 3) ? 1 : 0);            // (2,2), (3,2), (3,3) that exist
    if (part == 0) return first;
    if (part == 1) return 2 * nA0 + nd > 1 ? 2 * nA0 + nd : 1;     // at least one workgroup: the copies
    const int m2 = nt > 2 ? nt - 2 : 0;
    const int rest = m2 * (m2 + 1) / 2 - nd;
    return rest > 0 ? rest : 0;
}
__global__ __launch_bounds__(BG_T) __attribute__((amdgpu_waves_per_eu(3))) void hb2_update_kernel(double* __restrict__ band, int bw, int n, int k0,
                                                          const double* __restrict__ P, int bwp, int part, int nprob,
                                                          const double* __restrict__ PB, const double* __restrict__ cpP,
                                                          int cpk0) {
    __shared__ double lds[BG_LDS];
    const int W = bw + 1;
    const int img = (int)(blockIdx.x % (unsigned)nprob), tid = threadIdx.x;
    double* Bi = band + (size_t)img * n * W;
    const double* Pi = P + (size_t)img * bwp * HB2_NB;
    // LU path: A(R, C) -= sum_k PA(R, k) PB(C, k) with two different panels (L21 and U12^T); Cholesky: PB = PA
    const double* Pj = PB ? PB + (size_t)img * bwp * HB2_NB : Pi;
    const int nt = (bw + 63) / 64;
    const int bx = (int)(blockIdx.x / (unsigned)nprob), gx = (int)(gridDim.x / (unsigned)nprob);
    if (part == 1 || part == 2) {
        BCR_SPAN_RESET(1 + part, k0 / HB2_NB);
        BCR_SPAN_BEGIN(1 + part, k0 / HB2_NB);
    }
    int ta = -1, tb = -1;   // tile of this workgroup (none: copies only)
    if (part == 0) {
        int t = bx; ta = 0;
        while (t > ta) { t -= ta + 1; ++ta; }
        tb = t;
    } else if (part == 1 || part == 3) {   // 3: the tiles of part 1 without the copies (LU path, upper band)
        const int nA0 = nt > 2 ? nt - 2 : 0;
        if (bx < nA0) { ta = 2 + bx; tb = 0; }
        else if (bx < 2 * nA0) { ta = 2 + bx - nA0; tb = 1; }
        else {
            const int j = bx - 2 * nA0;
            if (j == 0 && nt >= 3) { ta = 2; tb = 2; }
            else if (j == 1 && nt >= 4) { ta = 3; tb = 2; }
            else if (j == 2 && nt >= 4) { ta = 3; tb = 3; }
        }
    } else if (part == 2) {
        int t = bx + ((nt >= 4) ? 3 : ((nt == 3) ? 1 : 0)), a = 0;
        while (t > a) { t -= a + 1; ++a; }
        ta = a + 2; tb = t + 2;
    }
    const int base = k0 + HB2_NB;
    // Copy of a finished panel P = L21 (the one that starts at column cpk0: this panel, or an earlier one whose band
    // entries were still being read) into the band -- only the substitutions read it -- off the critical path of the
    // next panel's Cholesky: rows [64 t, 64 t + 64) of the panel are item t, dealt round-robin over the workgroups
    // of the launch.  Explicit batches of 16 loads, then the stores: a rolled copy loop pays one memory latency per
    // iteration.  (The diagonal blocks of L are not stored: no kernel reads them.)
    if (cpP) {
        const int cbase = cpk0 + HB2_NB;
        const double* Pc = cpP + (size_t)img * bwp * HB2_NB;
        // a part-1 launch with more workgroups than tiles: the extra ones do the copies and the tile workgroups start
        // their products at once (the launch is as long as its slowest workgroup, and it sits on the stream that
        // bounds the pipeline)
        const int ntile = (part == 1) ? hb2_update_tiles(nt, 1) : gx;
        const int c0 = gx > ntile ? bx - ntile : bx, cs = gx > ntile ? gx - ntile : gx;
        for (int tc = c0; tc >= 0 && tc < nt; tc += cs) {
            if (cbase + 64 * tc >= n) continue;
            for (int e0 = tid; e0 < 64 * HB2_NB; e0 += 16 * BG_T) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = e0 + i * BG_T;
                    v[i] = Pc[64 * tc + (e & 63) + (size_t)bwp * (e >> 6)];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = e0 + i * BG_T, rr = 64 * tc + (e & 63), c = e >> 6;
                    const int R = cbase + rr, K = cpk0 + c;
                    if (R < n && K < n && R - K <= bw) Bi[(size_t)K * W + (R - K)] = v[i];
                }
            }
        }
    }
    if (ta < 0 || ta >= nt) return;
    if (base + 64 * tb >= n) return;
    if (base + 64 * ta >= n) return;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    double va[8], vb[8];
    if (part == 2) BCR_PROBE(32);
    hb2_fetch_dense(Pi, bwp, bwp, HB2_NB, 64 * ta, 0, tid, va);
    hb2_fetch_dense(Pj, bwp, bwp, HB2_NB, 64 * tb, 0, tid, vb);
    constexpr int nchunk = HB2_NB / BG_KC;
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        bg_stage<true>(As, tid, va);
        bg_stage<true>(Bs, tid, vb);
        __syncthreads();
        if (part == 2) BCR_PROBE(33 + 2 * ch);
        if (ch + 1 < nchunk) {
            hb2_fetch_dense(Pi, bwp, bwp, HB2_NB, 64 * ta, (ch + 1) * BG_KC, tid, va);
            hb2_fetch_dense(Pj, bwp, bwp, HB2_NB, 64 * tb, (ch + 1) * BG_KC, tid, vb);
        }
        hb2_mma_chunk(As, Bs, acc);
        if (part == 2) BCR_PROBE(34 + 2 * ch);
    }
    const int l = tid & 63, hq = tid >> 6;
    double old[16];   // requested together, before the accumulators go through LDS
    // Interior tiles (strictly below the diagonal, inside the band and the matrix: nearly all of them) need none of
    // the six tests per entry: entry (l, hq + 4 i) of the tile sits at a constant stride of 4 (W - 1) doubles.
    const bool interior = ta > tb && base + 64 * ta + 63 < n && 64 * (ta - tb) + 63 <= bw && 64 * ta + 63 < bw;
    double* t0 = Bi + (size_t)(base + 64 * tb + hq) * W + (64 * (ta - tb) + l - hq);
    const size_t tstep = (size_t)4 * (W - 1);
    if (interior) {
#pragma unroll
        for (int i = 0; i < 16; ++i) old[i] = t0[i * tstep];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = hq + 4 * i;
            const int R = base + 64 * ta + l, Cc = base + 64 * tb + o;
            const bool in = R < n && Cc < n && R >= Cc && R - Cc <= bw && 64 * ta + l < bw && 64 * tb + o < bw;
            old[i] = in ? Bi[(size_t)Cc * W + (R - Cc)] : 0.0;
        }
    }
    if (part == 2) BCR_PROBE(41);
    bg_to_lds(acc, lds);
    if (part == 2) BCR_PROBE(42);
    if (interior) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t0[i * tstep] = old[i] - lds[(hq + 4 * i) * BG_LD + l];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = hq + 4 * i;
            const int R = base + 64 * ta + l, Cc = base + 64 * tb + o;
            if (R < n && Cc < n && R >= Cc && R - Cc <= bw && 64 * ta + l < bw && 64 * tb + o < bw)
                Bi[(size_t)Cc * W + (R - Cc)] = old[i] - lds[o * BG_LD + l];
        }
    }
    if (part == 2) BCR_PROBE(43);
    if (part == 1 || part == 2) BCR_SPAN_END(1 + part, k0 / HB2_NB);
}

// Substitutions with L in the band array, one launch per 128-column block (a single workgroup streaming
// the 8.6 GB factor of a 1024^2 image would be limited to one CU's bandwidth; the kernel boundary is the
// grid-wide synchronisation).  Every 1024-thread workgroup recomputes the block's solution with the
// inverted diagonal block of hb3_chain_kernel (the whole 128 KB block requested at once, as in the
// substitutions of the block cyclic reduction); workgroup 0 stores it, workgroup 1+j applies it to its
// 128 rows of the band (thread (row, part): 16 consecutive columns each, partial sums meet in LDS).
// Forward (L y = b):   in/out `x` = running right-hand side, solution rows -> `y`.
// Backward (L^T x = y): in/out `y` = running right-hand side, solution rows -> `x` (and += acc).
// RW = rows (forward) / earlier equations (backward) per tile workgroup: 128 = one workgroup per 128 x 128 tile; 32 =
// four workgroups per tile, each streaming the 64 KB inverse block and a quarter of the tile -- a launch is as long as
// one CU needs for its workgroup's bytes, and the inverse block is the part every workgroup has to read.
// grid (1 + ceil(bw / RW), O), block BS_T.
__host__ __device__ inline unsigned hb2_subst_grid(int bw, int RW) { return 1u + (unsigned)((bw + RW - 1) / RW); }

template <int RW>
__global__ __launch_bounds__(BS_T) void hb2_fwd_kernel(const double* __restrict__ band, const double* __restrict__ Linv,
                                                       int bw, int n, int k0, int npanel, double* __restrict__ x,
                                                       double* __restrict__ y, int unit_diag = 0) {
    constexpr int NPART = BS_T / RW, CP = HB2_NB / NPART;   // parts of a row, columns per part
    __shared__ double v[HB2_NB], yb[HB2_NB];
    __shared__ __attribute__((aligned(16))) double red[BS_W * BS_MP];
    static_assert(NPART * RW <= BS_W * BS_MP, "partial sums fit the reduction buffer");
    const int W = bw + 1;
    const int img = blockIdx.y, tid = threadIdx.x;
    const double* Bi = band + (size_t)img * n * W;
    double* xv = x + (size_t)img * n;
#ifdef BCR_PROBE_ON
#define HB2_FP(i) do { if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 0) bcr_probe_buf[i] = (long long)wall_clock64(); } while (0)
#else
#define HB2_FP(i) do { } while (0)
#endif
    HB2_FP(50);
    if (tid < HB2_NB) v[tid] = (k0 + tid < n) ? xv[k0 + tid] : 0.0;
    // this workgroup's rows of L are requested BEFORE the diagonal-block product: their addresses do not
    // depend on the solution, so their memory latency overlaps with that of the 128 KB inverse block
    const int rr = tid % RW, part = tid / RW;
    const int R = k0 + HB2_NB + ((int)blockIdx.x - 1) * RW + rr;
    const bool upd = blockIdx.x > 0 && R < n && R <= k0 + HB2_NB - 1 + bw;
    double lt[CP];
#pragma unroll
    for (int i = 0; i < CP; ++i) {
        const int c = CP * part + i, d0 = R - (k0 + c);
        lt[i] = (upd && d0 <= bw) ? Bi[(size_t)(k0 + c) * W + d0] : 0.0;
    }
    __syncthreads();
    HB2_FP(51);
    double val;
    if (unit_diag) {   // LU path: the diagonal block of L is the identity
        val = (tid < HB2_NB) ? v[tid] : 0.0;
    } else {
        double s0 = 0.0, s1 = 0.0;
        bcr_mv_partial(Linv + ((size_t)img * npanel + k0 / HB2_NB) * HB2_NB * HB2_NB, HB2_NB, v, 1, s0, s1);
        val = bcr_mv_reduce(red, HB2_NB, s0, s1);
    }
    if (blockIdx.x == 0) {
        if (tid < HB2_NB && k0 + tid < n) y[(size_t)img * n + k0 + tid] = val;
        return;
    }
    HB2_FP(52);
    if (tid < HB2_NB) yb[tid] = val;
    __syncthreads();
    double s = 0.0;
    if (upd) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < CP; i += 2) {
            const int c = CP * part + i;
            a0 = __builtin_fma(lt[i], yb[c], a0);
            a1 = __builtin_fma(lt[i + 1], yb[c + 1], a1);
        }
        s = a0 + a1;
    }
    red[part * RW + rr] = s;
    __syncthreads();
    HB2_FP(53);
    if (tid < RW && upd) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NPART; ++q) t += red[q * RW + tid];
        xv[R] -= t;
    }
    HB2_FP(54);
}

// apply_only (twisted solve): block k0 lies in the trailing window of a partially factored problem; its
// solution is already in `x` (the middle block's), only the earlier equations are updated.
// mvmode: 2 = `LinvT` is upper triangular (Cholesky); 0 = a full matrix (LU path: the inverse of the diagonal block
// of U, with `band` = the upper band stored by rows).
template <int RW>
__global__ __launch_bounds__(BS_T) void hb2_bwd_kernel(const double* __restrict__ band, const double* __restrict__ LinvT,
                                                       int bw, int n, int k0, int npanel, double* __restrict__ y,
                                                       double* __restrict__ x, double* __restrict__ acc, int apply_only,
                                                       int mvmode = 2) {
    constexpr int NPART = BS_T / RW, CP = HB2_NB / NPART;   // parts of a column, rows per part
    __shared__ double v[HB2_NB], xb[HB2_NB];
    __shared__ __attribute__((aligned(16))) double red[BS_W * BS_MP];
    const int W = bw + 1;
    const int img = blockIdx.y, tid = threadIdx.x;
    const double* Bi = band + (size_t)img * n * W;
    double* yv = y + (size_t)img * n;
    // this workgroup's rows of L^T (earlier equation k: y_k -= sum_c L[k0+c][k] x_{k0+c}, L[k0+c][k] = band[k*W + (k0+c-k)])
    // are requested BEFORE the diagonal-block product: their addresses do not depend on the solution
    const int rr = tid % RW, part = tid / RW;
    const int k = k0 - 1 - (((int)blockIdx.x - 1) * RW + rr);
    const bool upd = blockIdx.x > 0 && k >= 0 && k0 - k <= bw;
    const int cb = CP * part;
    double lt[CP];
    {
        const double* col = Bi + (size_t)(upd ? k : 0) * W + (upd ? k0 - k : 0);
        if (upd && (bw & 1) == 0 && (((size_t)img * n) & 1) == 0 && k0 - k + cb + CP - 1 <= bw && k0 + cb + CP - 1 < n) {
            // the thread's entries are contiguous: 16-byte loads (k(W-1) + k0 is even for even bw)
#pragma unroll
            for (int i = 0; i < CP / 2; ++i) {
                const double2 q = *reinterpret_cast<const double2*>(col + cb + 2 * i);
                lt[2 * i] = q.x; lt[2 * i + 1] = q.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int c = cb + i;
                lt[i] = (upd && k0 - k + c <= bw && k0 + c < n) ? col[c] : 0.0;
            }
        }
    }
    if (apply_only) {
        if (blockIdx.x == 0) return;
        if (tid < HB2_NB) xb[tid] = (k0 + tid < n) ? x[(size_t)img * n + k0 + tid] : 0.0;
    } else {
        if (tid < HB2_NB) v[tid] = (k0 + tid < n) ? yv[k0 + tid] : 0.0;
        __syncthreads();
        double s0 = 0.0, s1 = 0.0;
        bcr_mv_partial(LinvT + ((size_t)img * npanel + k0 / HB2_NB) * HB2_NB * HB2_NB, HB2_NB, v, mvmode, s0, s1);
        const double val = bcr_mv_reduce(red, HB2_NB, s0, s1);
        if (blockIdx.x == 0) {
            if (tid < HB2_NB && k0 + tid < n) {
                x[(size_t)img * n + k0 + tid] = val;
                if (acc) acc[(size_t)img * n + k0 + tid] += val;
            }
            return;
        }
        if (tid < HB2_NB) xb[tid] = (k0 + tid < n) ? val : 0.0;
    }
    __syncthreads();
    double s = 0.0;
    if (upd) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < CP; i += 2) {
            a0 = __builtin_fma(lt[i], xb[cb + i], a0);
            a1 = __builtin_fma(lt[i + 1], xb[cb + i + 1], a1);
        }
        s = a0 + a1;
    }
    red[part * RW + rr] = s;
    __syncthreads();
    if (tid < RW && upd) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NPART; ++q) t += red[q * RW + tid];
        yv[k] -= t;
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
