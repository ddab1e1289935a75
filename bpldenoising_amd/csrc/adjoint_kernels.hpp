// adjoint_kernels.hpp -- device side of the adjoint-state gradients d(0.5||u(alpha)-ubar||^2)/d alpha.
//
// Replaces gradient / gradient_reg of /root/reference/src/TVLearningFunctionVec.jl:98-161,192-254
// (sparse LU of a 3n^2 saddle system per image).  Same reduced SPD system as oracle/bpltv_oracle.c:
//     (I + S G^T W G S) q = rhs,   W_k = c_k t_k t_k^T + kap_k I   (per pixel 2x2),
// whose matrix has bandwidth M in the column-major pixel order.  Per image, one workgroup:
//   assemble (4 diagonals) -> banded Cholesky with the (M+1)x(M+1) trailing window resident in LDS
//   (133 KB of the 160 KB for M = 128) -> blocked forward/backward substitution -> iterative
//   refinement with matrix-free residuals -> per-pixel gradient contributions -> patch sums.
// Images are independent, so O workgroups run concurrently on O CUs.
#pragma once
#include <hip/hip_runtime.h>
#include "pdhg_kernels.hpp"

namespace bpltv {

struct AdjCoef {  // planar per-pixel arrays, [O][M*N] each
    double* t1;
    double* t2;
    double* c;
    double* kap;
    double* h1;
    double* h2;
    double* s;  // node scaling (reg + patch only), else unused
    double* rhs;
};

constexpr double ADJ_ACT_TOL = 1e-12;  // TVLearningFunctionVec.jl:109,231
constexpr double ADJ_GAMMA = 1e8;      // TVLearningFunctionVec.jl:142,197

// Per-pixel coefficients (xi, Den, prodesc, active/inactive masks of the reference, fused).
__global__ __launch_bounds__(256) void adj_setup_kernel(const double* __restrict__ u,
                                                        const double* __restrict__ ubar,
                                                        const double* __restrict__ alpha, int am, int an,
                                                        int M, int N, int O, int patch, int reg,
                                                        double kappa_act, AdjCoef C) {
    const size_t npx = (size_t)M * N;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= npx * O) return;
    const int k = (int)(q % npx);
    const int i = k % M, j = k / M;
    const double uk = u[q];
    const double g1 = (i < M - 1) ? u[q + 1] - uk : 0.0;
    const double g2 = (j < N - 1) ? u[q + M] - uk : 0.0;
    const double ng = sqrt(g1 * g1 + g2 * g2);
    const double a = alpha_at(alpha, am, an, M, N, i, j);
    double t1 = 0.0, t2 = 0.0, c = 0.0, kap = 0.0, h1 = 0.0, h2 = 0.0, rhs, s = 1.0;
    if (!reg) {
        if (ng < ADJ_ACT_TOL) {
            kap = kappa_act;
        } else {
            t1 = -g2 / ng; t2 = g1 / ng;
            c = a / ng;
            h1 = g1 / ng; h2 = g2 / ng;
        }
        rhs = uk - ubar[q];
    } else {
        if (ng > 1.0 / ADJ_GAMMA) {
            t1 = -g2 / ng; t2 = g1 / ng;
            c = patch ? 1.0 / ng : a / ng;
            h1 = g1 / ng; h2 = g2 / ng;
        } else {
            kap = patch ? ADJ_GAMMA : a * ADJ_GAMMA;
            h1 = ADJ_GAMMA * g1; h2 = ADJ_GAMMA * g2;
        }
        rhs = ubar[q] - uk;
        if (patch) {
            s = sqrt(a);
            rhs = rhs / s;
        }
    }
    C.t1[q] = t1; C.t2[q] = t2; C.c[q] = c; C.kap[q] = kap; C.h1[q] = h1; C.h2[q] = h2;
    C.rhs[q] = rhs; C.s[q] = s;
}

// Element (pixel a) contribution to the node pairs of {a, b=a+1, c=a+M}.
struct Elem { double aa, bb, cc, ab, ac, bc; };
__device__ __forceinline__ Elem adj_elem(const AdjCoef& C, size_t q, int i, int j, int M, int N) {
    const bool hb = (i < M - 1), hc = (j < N - 1);
    const double e1 = hb ? C.t1[q] : 0.0, e2 = hc ? C.t2[q] : 0.0;
    const double sa = C.s[q], sb = hb ? C.s[q + 1] : 1.0, sc = hc ? C.s[q + M] : 1.0;
    const double va = -(e1 + e2) * sa, vb = e1 * sb, vc = e2 * sc;
    const double c = C.c[q], kp = C.kap[q];
    Elem e;
    e.aa = c * va * va + kp * ((hb ? sa * sa : 0.0) + (hc ? sa * sa : 0.0));
    e.bb = c * vb * vb + (hb ? kp * sb * sb : 0.0);
    e.cc = c * vc * vc + (hc ? kp * sc * sc : 0.0);
    e.ab = c * va * vb - (hb ? kp * sa * sb : 0.0);
    e.ac = c * va * vc - (hc ? kp * sa * sc : 0.0);
    e.bc = c * vb * vc;
    return e;
}

// The four non-zero diagonals of the lower band: band4[0]=diag, [1]=offset 1, [2]=offset M-1,
// [3]=offset M; planar [4][O][M*N].  Gather form (no atomics).
__global__ __launch_bounds__(256) void adj_assemble_kernel(AdjCoef C, int M, int N, int O,
                                                           double* __restrict__ band4) {
    const size_t npx = (size_t)M * N, tot = npx * O;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= tot) return;
    const int k = (int)(q % npx);
    const int i = k % M, j = k / M;
    const Elem e0 = adj_elem(C, q, i, j, M, N);
    double diag = 1.0 + e0.aa;
    double o1 = e0.ab, oM = e0.ac, oMm1 = 0.0;
    if (i > 0) {
        const Elem e = adj_elem(C, q - 1, i - 1, j, M, N);
        diag += e.bb;
        oMm1 = e.bc;  // pair (b = q, c = q - 1 + M)
    }
    if (j > 0) {
        const Elem e = adj_elem(C, q - M, i, j - 1, M, N);
        diag += e.cc;
    }
    band4[q] = diag;
    band4[tot + q] = o1;
    band4[2 * tot + q] = oMm1;
    band4[3 * tot + q] = oM;
}

__device__ __forceinline__ double band_init(const double* __restrict__ band4, size_t tot, size_t q,
                                            int d, int M) {
    double v = 0.0;
    if (d == 0) v += band4[q];
    if (d == 1) v += band4[tot + q];
    if (d == M - 1) v += band4[2 * tot + q];
    if (d == M) v += band4[3 * tot + q];
    return v;
}

// Same matrix in the reversed ordering k' = n-1-k (used by the bottom-up half of the twisted
// factorisation): entry (k'+d, k') of the reversed matrix is A[k][k-d], k = n-1-k'.
__device__ __forceinline__ double band_init_rev(const double* __restrict__ band4, size_t tot, size_t ib,
                                                int n, int kp, int d, int M) {
    const int k = n - 1 - kp, j = k - d;
    if (j < 0) return 0.0;
    double v = 0.0;
    if (d == 0) v += band4[ib + k];
    if (d == 1) v += band4[tot + ib + j];
    if (d == M - 1) v += band4[2 * tot + ib + j];
    if (d == M) v += band4[3 * tot + ib + j];
    return v;
}

// Split of the n columns for the twisted (two-sided) factorisation: the top half eliminates
// [0, m) downwards, the bottom half eliminates the last nbot columns upwards (both multiples of the
// panel width), the nm = n - m - nbot >= bw middle columns are factored last as a dense block.
constexpr int ADJ_G = 64;  // granularity of the split = block size of the substitution kernels
struct AdjSplit { int m, nbot, nm; };
__host__ __device__ inline AdjSplit adj_split(int n, int bw, int NB = ADJ_G) {
    AdjSplit s;
    int half = (n - bw) / 2;
    if (half < 0) half = 0;
    s.m = (half / NB) * NB;
    s.nbot = ((n - bw - s.m) / NB) * NB;
    if (s.nbot < 0) s.nbot = 0;
    s.nm = n - s.m - s.nbot;
    return s;
}

// Broadcast of lane `src` (wave-uniform, a constant after unrolling) through SGPRs: v_readlane_b32,
// a few cycles -- not ds_bpermute, whose ~100-cycle latency would sit in the substitution chain.
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// sqrt and 1/sqrt of a positive pivot from v_rsq_f64 + Newton (the adjoint solve is refined
// iteratively and compared with a tolerance, so no correctly-rounded sqrt/divide chain is needed
// in the sequential part of the factorisation).
__device__ __forceinline__ void sqrt_rsqrt(double a, double& s, double& r) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * __builtin_fma(-h, y * y, 1.5);
    y = y * __builtin_fma(-h, y * y, 1.5);
    double g = a * y;
    g = __builtin_fma(0.5 * y, __builtin_fma(-g, g, a), g);
    s = g;
    r = y;
}

// Banded Cholesky A = L L^T, bandwidth bw = M, one workgroup (ADJ_FT threads) per image, blocked by
// panels of NB columns.
//   LDS: ring of RS = bw+NB column slots x W = bw+1 rows (the trailing window; 141 KB for M = 128,
//        NB = 8), the current panel lp[NB][bw+NB] and the NB x NB diagonal factor.
//   per panel: (1) every thread redundantly factors the NB x NB diagonal block (no barrier),
//   (2) one thread per row forward-substitutes its NB panel entries, stores them to LDS and to L,
//   (3) barrier; the bw(bw+1)/2 trailing elements -- a fixed, balanced list of (column, offset)
//   pairs per thread, decoded once -- each subtract an NB-term dot product of panel rows; the NB
//   consumed slots are re-initialised with the next columns of A; barrier.
// L is written to global memory as [O][n][W] (column k: L[k+d][k] at d).
constexpr int ADJ_FT = 512;         // threads of the factorisation workgroup (256 VGPRs each)

template <int NB, int MC>  // MC: compile-time M (0 = run-time)
__global__ __launch_bounds__(ADJ_FT) void adj_factor_kernel(const double* __restrict__ band4, int Mrt, int N,
                                                          int O, double* __restrict__ L0,
                                                          double* __restrict__ L1, int twisted,
                                                          double* __restrict__ dump,
                                                          int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int M = MC ? MC : Mrt;
    const int bw = M, W = M + 1, RS = bw + NB, PR = bw + NB;  // ring slots, panel rows
    double* win = smem;                // [RS][W]
    double* lp = smem + RS * W;        // [NB][PR]
    const int img = blockIdx.x;
    const int n = M * N;
    const size_t tot = (size_t)n * O;
    const size_t ib = (size_t)img * n;
    const int tid = threadIdx.x;
    // twisted: blockIdx.y = 0 eliminates columns [0, m) of A, blockIdx.y = 1 the first nbot columns
    // of the reversed matrix; both leave their trailing window (the middle block) in `dump`.
    const int rev = twisted ? (int)blockIdx.y : 0;
    const AdjSplit sp = adj_split(n, bw);
    const int ncols = twisted ? (rev ? sp.nbot : sp.m) : n;
    double* __restrict__ L = rev ? L1 : L0;
    auto A0 = [&](int c, int d) -> double {  // initial entry (c+d, c) in this side's ordering
        if (c >= n) return d == 0 ? 1.0 : 0.0;
        return rev ? band_init_rev(band4, tot, ib, n, c, d, M) : band_init(band4, tot, ib + c, d, M);
    };

    for (int e = tid; e < RS * W; e += ADJ_FT) {
        const int c = e / W, d = e - c * W;
        win[e] = A0(c, d);
    }
    // trailing elements of a panel step: window column t (0..bw-1, i.e. panel-relative column NB+t)
    // holds bw - t elements (row offsets o = 0 .. bw-t-1); linear index -> (t, o), decoded once.
    // The bw(bw+1)/2 trailing elements as a rectangle: row p pairs window column p (bw-p elements)
    // with column bw-1-p (p+1 elements) -> bw+1 = W entries per row, HR = ceil(bw/2) rows (for odd
    // bw the middle column stands alone).  Linear index e = p*W + q walks it with stride ADJ_FT.
    const int HR = (bw + 1) / 2;
    const int NEr = HR * W;
    const int dp = ADJ_FT / W, dq = ADJ_FT - dp * W;
    const int p0 = tid / W, q0 = tid - p0 * W;
    __syncthreads();

    int slot0 = 0;  // ring slot of column k
    for (int k = 0; k < ncols; k += NB) {
        // ---- (1) diagonal block, redundantly in every thread of the row waves (no barrier needed):
        //          m = chol(A[k..k+NB, k..k+NB]).  The other waves go straight to the barrier.
        double m[NB][NB], dinv[NB];
        const bool rowwave = (tid & ~63) < PR;  // wave holds at least one panel row
        if (rowwave) {
            bool bad = false;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int sb = slot0 + b; if (sb >= RS) sb -= RS;
#pragma unroll
                for (int a = b; a < NB; ++a) m[a][b] = (a - b <= bw) ? win[sb * W + (a - b)] : 0.0;  // bw < NB-1: outside the band
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                double piv = m[c][c];
#pragma unroll
                for (int q = 0; q < c; ++q) piv -= m[c][q] * m[c][q];
                if (!(piv > 0.0)) bad = true;
                double d, di;
                sqrt_rsqrt(piv, d, di);
                m[c][c] = d;
                dinv[c] = di;
#pragma unroll
                for (int a = c + 1; a < NB; ++a) {
                    double v = m[a][c];
#pragma unroll
                    for (int q = 0; q < c; ++q) v -= m[a][q] * m[c][q];
                    m[a][c] = v * dinv[c];
                }
            }
            if (bad && tid == 0) fail[img] = k + 1;  // factor continues with NaNs; host reports the failure
        }
        // ---- (2) panel rows
        if (rowwave && tid < PR) {
            const int rr = tid;
            double l[NB];
            if (rr < NB) {
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int a = 0; a < NB; ++a)
                        if (a == rr && c <= a) v = m[a][c];
                    l[c] = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    int sc = slot0 + c; if (sc >= RS) sc -= RS;
                    double v = (rr - c <= bw) ? win[sc * W + (rr - c)] : 0.0;
#pragma unroll
                    for (int q = 0; q < c; ++q) v -= l[q] * m[c][q];
                    l[c] = v * dinv[c];
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                lp[c * PR + rr] = l[c];
                const int d = rr - c;
                if (d >= 0 && d <= bw && k + c < n) L[(ib + k + c) * W + d] = (k + rr < n) ? l[c] : 0.0;
            }
        }
        __syncthreads();
        // ---- (3a) recycle the NB consumed slots for columns k+RS .. k+RS+NB-1
        for (int e = tid; e < NB * W; e += ADJ_FT) {
            const int q = e / W, d = e - q * W;
            int sq = slot0 + q; if (sq >= RS) sq -= RS;
            win[sq * W + d] = A0(k + RS + q, d);
        }
        // ---- (3b) trailing update: A[j+o][j] -= sum_c lp[c][row] * lp[c][col]
        {
            int p = p0, q = q0;
            for (int e = tid; e < NEr; e += ADJ_FT) {
                const bool first = q < bw - p;
                const int t = first ? p : bw - 1 - p;
                const int o = first ? q : q - (bw - p);
                if (first || 2 * p != bw - 1) {  // odd bw: the middle column is not paired with itself
                    const int jr = NB + t, rrow = jr + o;
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < NB; ++c) acc = __builtin_fma(lp[c * PR + rrow], lp[c * PR + jr], acc);
                    int sj = slot0 + jr; if (sj >= RS) sj -= RS;
                    win[sj * W + o] -= acc;
                }
                p += dp; q += dq;
                if (q >= W) { q -= W; ++p; }
            }
        }
        __syncthreads();
        slot0 += NB; if (slot0 >= RS) slot0 -= RS;
    }
    if (twisted) {  // window columns ncols .. ncols+nm-1, rows inside the middle block
        double* dd = dump + ((size_t)img * 2 + rev) * (size_t)(bw + ADJ_G) * (bw + ADJ_G);
        const int nm = sp.nm;
        for (int e = tid; e < nm * nm; e += ADJ_FT) {
            const int q = e / nm, d = e - q * nm;
            if (q + d < nm) {
                int sq = slot0 + q; if (sq >= RS) sq -= RS;
                dd[q * nm + d] = (d <= bw) ? win[sq * W + d] : 0.0;
            }
        }
    }
}

// Dense Cholesky of the middle block of the twisted factorisation: D = (top window) + (bottom
// window, index-reversed) - A, nm x nm with bw <= nm < bw + NB, one workgroup per image, D in LDS.
// Lm: [O][nm*nm] row-major lower triangle.
__global__ __launch_bounds__(256) void adj_mid_factor_kernel(const double* __restrict__ band4,
                                                             const double* __restrict__ dump, int M, int N,
                                                             int O, double* __restrict__ Lm,
                                                             int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bw = M, n = M * N;
    const AdjSplit sp = adj_split(n, bw);
    const int nm = sp.nm, ld = nm + 1;
    double* D = smem;  // [nm][ld]
    const int img = blockIdx.x, tid = threadIdx.x;
    const size_t tot = (size_t)n * O, ib = (size_t)img * n;
    const double* dt = dump + ((size_t)img * 2) * (size_t)(bw + ADJ_G) * (bw + ADJ_G);
    const double* db = dt + (size_t)(bw + ADJ_G) * (bw + ADJ_G);
    for (int e = tid; e < nm * nm; e += 256) {
        const int r = e / nm, c = e - r * nm;
        double v = 0.0;
        if (r >= c) {
            const int d = r - c;
            v = dt[c * nm + d] + db[(nm - 1 - r) * nm + d] - ((d <= bw) ? band_init(band4, tot, ib + sp.m + c, d, M) : 0.0);
        }
        D[r * ld + c] = v;
    }
    __syncthreads();
    for (int k = 0; k < nm; ++k) {
        const double piv = D[k * ld + k];
        if (!(piv > 0.0) && tid == 0) fail[img] = sp.m + k + 1;
        double d, di;
        sqrt_rsqrt(piv, d, di);
        __syncthreads();
        for (int r = k + tid; r < nm; r += 256) D[r * ld + k] = (r == k) ? d : D[r * ld + k] * di;
        __syncthreads();
        const int rem = nm - 1 - k;
        for (int e = tid; e < rem * rem; e += 256) {
            const int r = k + 1 + e / rem, c = k + 1 + e % rem;
            if (r >= c) D[r * ld + c] -= D[r * ld + k] * D[c * ld + k];
        }
        __syncthreads();
    }
    double* out = Lm + (size_t)img * (bw + ADJ_G) * (bw + ADJ_G);
    for (int e = tid; e < nm * nm; e += 256) {
        const int r = e / nm, c = e - r * nm;
        out[e] = (r >= c) ? D[r * ld + c] : 0.0;
    }
}

// Inverses of the 64x64 diagonal blocks of L, so that the block substitutions of the solve become
// matrix-vector products without a 64-step dependent chain.  One wave per block: lane j builds
// column j of inv(L_bb) by forward substitution (L_bb staged in LDS).  Output, per image and block,
// two layouts of the same lower-triangular matrix X = inv(L_bb):
//   invF[(c*64 + r)] = X[r][c]  (lane r reads row r, coalesced over r)     -> forward  y = X b
//   invB[(r*64 + c)] = X[r][c]  (lane c reads column c, coalesced over c)  -> backward x = X^T z
// Rows/columns beyond n are padded with the identity.
constexpr int SB = 64;
__global__ __launch_bounds__(64) void adj_invdiag_kernel(const double* __restrict__ L, int M, int N, int ncols,
                                                         double* __restrict__ invF, double* __restrict__ invB) {
    __shared__ double tri[SB * (SB + 1)];  // L_bb element (r, c) at r*65 + c
    __shared__ double X[SB * (SB + 1)];    // X element (r, j) at r*65 + j
    const int W = M + 1, bw = M, ntot = M * N;
    const int n = ncols;  // columns of this factor (the whole matrix, or one side of the twisted split)
    const int nblk = (ntot + SB - 1) / SB;
    const int bi = blockIdx.x, img = blockIdx.y;
    const int k0 = bi * SB, lane = threadIdx.x;
    if (k0 >= n) return;
    const double* Li = L + (size_t)img * ntot * W;
    for (int c = 0; c < SB; ++c) {  // column c of the block: rows r = lane, coalesced
        const int r = lane;
        double v = (r == c) ? 1.0 : 0.0;
        if (r >= c && k0 + r < n && k0 + c < n && r - c <= bw) v = Li[(size_t)(k0 + c) * W + (r - c)];
        tri[r * (SB + 1) + c] = v;
    }
    __syncthreads();
    const int j = lane;
    for (int r = 0; r < SB; ++r) {
        double v = 0.0;
        if (r == j) {
            v = 1.0 / tri[r * (SB + 1) + r];
        } else if (r > j) {
            double sacc = 0.0;
            for (int c = j; c < r; ++c) sacc = __builtin_fma(tri[r * (SB + 1) + c], X[c * (SB + 1) + j], sacc);
            v = -sacc / tri[r * (SB + 1) + r];
        }
        X[r * (SB + 1) + j] = v;  // lane j only touches column j: no barrier needed
    }
    __syncthreads();
    const size_t ob = ((size_t)img * nblk + bi) * SB * SB;
    for (int q = 0; q < SB; ++q) {
        invF[ob + (size_t)q * SB + lane] = X[lane * (SB + 1) + q];  // c = q, r = lane
        invB[ob + (size_t)q * SB + lane] = X[q * (SB + 1) + lane];  // r = q, c = lane
    }
}

// Solve L L^T x = b in place (x: [O][n]), blocks of 64 columns, one workgroup of 256 per image.
// The running right-hand side lives in an LDS ring of 256 rows; per block, wave 0 applies the
// inverted diagonal block (64 fmas per lane, operands broadcast by v_readlane), the other waves
// have already prefetched their rows of L and apply the rank-64 update to the following bw rows
// (forward) / the tail dot products with already solved entries (backward).
// If `acc` is non-null the solution is also added to acc (p += dp).
constexpr int RING = 256;
__global__ __launch_bounds__(256) void adj_solve_kernel(const double* __restrict__ L,
                                                        const double* __restrict__ invF,
                                                        const double* __restrict__ invB, int M, int N,
                                                        double* __restrict__ x,
                                                        double* __restrict__ acc) {
    __shared__ double ring[RING];
    __shared__ double xs[SB];
    __shared__ double ps[SB][4];
    const int W = M + 1, bw = M;
    const int n = M * N;
    const size_t ib = (size_t)blockIdx.x * n;
    const double* Li = L + ib * W;
    double* xv = x + ib;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + SB - 1) / SB;
    const double* iF = invF + (size_t)blockIdx.x * nblk * SB * SB;
    const double* iB = invB + (size_t)blockIdx.x * nblk * SB * SB;

    // ---------------- forward: L y = b.  ring[r & 255] holds the current b[r] for r in [k0, k0+256)
    ring[tid] = (tid < n) ? xv[tid] : 0.0;
    __syncthreads();
    for (int bi = 0; bi < nblk; ++bi) {
        const int k0 = bi * SB;
        const int rend = (k0 + SB - 1 + bw < n - 1) ? (k0 + SB - 1 + bw) : (n - 1);
        double pre[SB];
        const int urow = k0 + SB + (tid - 64);  // rows beyond the block, one per thread of waves 1..3
        if (wv == 0) {
            double xr[SB];
            const double* blk = iF + (size_t)bi * SB * SB;
#pragma unroll
            for (int c = 0; c < SB; ++c) xr[c] = blk[c * SB + lane];  // X[lane][c]
            const double bv = ring[(k0 + lane) & (RING - 1)];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int c = 0; c < SB; c += 4) {
                a0 = __builtin_fma(xr[c], readlane_f64(bv, c), a0);
                a1 = __builtin_fma(xr[c + 1], readlane_f64(bv, c + 1), a1);
                a2 = __builtin_fma(xr[c + 2], readlane_f64(bv, c + 2), a2);
                a3 = __builtin_fma(xr[c + 3], readlane_f64(bv, c + 3), a3);
            }
            const double val = (a0 + a1) + (a2 + a3);
            xs[lane] = val;
            if (k0 + lane < n) xv[k0 + lane] = val;
        } else {
#pragma unroll
            for (int c = 0; c < SB; ++c) {
                const int d = urow - (k0 + c);
                pre[c] = (urow <= rend && d <= bw) ? Li[(size_t)(k0 + c) * W + d] : 0.0;
            }
        }
        __syncthreads();
        if (wv == 0) {  // refill the 64 ring slots that just left the window
            const int rnew = k0 + RING + lane;
            ring[(k0 + lane) & (RING - 1)] = (rnew < n) ? xv[rnew] : 0.0;
        } else if (urow <= rend) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int c = 0; c < SB; c += 2) {
                s0 = __builtin_fma(pre[c], xs[c], s0);
                s1 = __builtin_fma(pre[c + 1], xs[c + 1], s1);
            }
            ring[urow & (RING - 1)] -= s0 + s1;
        }
        __syncthreads();
    }
    // ---------------- backward: L^T x = y.  ring[r & 255] holds solved x[r] for r in [k0+64, k0+256)
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int k0 = bi * SB;
        const int nb = (n - k0 < SB) ? (n - k0) : SB;
        double xc[SB];
        if (wv == 0) {
            const double* blk = iB + (size_t)bi * SB * SB;
#pragma unroll
            for (int r = 0; r < SB; ++r) xc[r] = blk[r * SB + lane];  // X[r][lane]
        }
        {   // tails: column c, rows beyond the block (already solved, in the ring)
            const int c = tid >> 2, part = tid & 3;
            double sacc = 0.0;
            if (c < nb) {
                const int kc = k0 + c;
                const int dlo = k0 + nb - kc;
                const int dhi = (n - 1 - kc < bw) ? (n - 1 - kc) : bw;
                double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                for (int d0 = dlo + part; d0 <= dhi; d0 += 32) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int d = d0 + 4 * u;
                        if (d <= dhi) a8[u] = __builtin_fma(Li[(size_t)kc * W + d], ring[(kc + d) & (RING - 1)], a8[u]);
                    }
                }
                sacc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            }
            ps[c][part] = sacc;
        }
        __syncthreads();
        if (wv == 0) {
            double zv = 0.0;
            if (lane < nb) zv = xv[k0 + lane] - (((ps[lane][0] + ps[lane][1]) + ps[lane][2]) + ps[lane][3]);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int r = 0; r < SB; r += 4) {
                a0 = __builtin_fma(xc[r], readlane_f64(zv, r), a0);
                a1 = __builtin_fma(xc[r + 1], readlane_f64(zv, r + 1), a1);
                a2 = __builtin_fma(xc[r + 2], readlane_f64(zv, r + 2), a2);
                a3 = __builtin_fma(xc[r + 3], readlane_f64(zv, r + 3), a3);
            }
            const double val = (a0 + a1) + (a2 + a3);
            ring[(k0 + lane) & (RING - 1)] = val;
            if (lane < nb) {
                if (acc) acc[ib + k0 + lane] += val;
                xv[k0 + lane] = val;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Twisted (two-sided) variant of the solve: two workgroups per image.  Side 0 works on columns
// [0, m) of A in natural order, side 1 on the first nbot columns of the index-reversed matrix; both
// column counts are multiples of 64.  PHASE 0: forward substitution of the side, the updated
// right-hand side of the nm middle rows is left in `spill`.  PHASE 1: backward substitution of the
// side, starting from the already solved middle rows.  adj_mid_solve_kernel sits in between.
// ------------------------------------------------------------------------------------------
template <int PHASE>
__global__ __launch_bounds__(256) void adj_solve_tw_kernel(const double* __restrict__ L0, const double* __restrict__ L1,
                                                           const double* __restrict__ invF, const double* __restrict__ invB,
                                                           int M, int N, double* __restrict__ x,
                                                           double* __restrict__ acc, double* __restrict__ spill) {
    __shared__ double ring[RING];
    __shared__ double xs[SB];
    __shared__ double ps[SB][4];
    const int W = M + 1, bw = M, n = M * N;
    const AdjSplit sp = adj_split(n, bw);
    const int img = blockIdx.x, rev = blockIdx.y;
    const int ncol = rev ? sp.nbot : sp.m, nrow = ncol + sp.nm;
    const size_t ib = (size_t)img * n;
    const double* Li = (rev ? L1 : L0) + ib * W;
    const int nblk_tot = (n + SB - 1) / SB;
    const size_t ioff = ((size_t)rev * gridDim.x + img) * nblk_tot * SB * SB;
    const double* iF = invF + ioff;
    const double* iB = invB + ioff;
    double* xv = x + ib;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = ncol / SB;
    auto XI = [&](int k) -> int { return rev ? n - 1 - k : k; };  // side-local row -> global row
    double* sp_out = spill + ((size_t)img * 2 + rev) * RING;

    if (PHASE == 0) {
        ring[tid] = (tid < nrow) ? xv[XI(tid)] : 0.0;
        __syncthreads();
        for (int bi = 0; bi < nblk; ++bi) {
            const int k0 = bi * SB;
            const int rend = (k0 + SB - 1 + bw < nrow - 1) ? (k0 + SB - 1 + bw) : (nrow - 1);
            double pre[SB];
            const int urow = k0 + SB + (tid - 64);
            if (wv == 0) {
                double xr[SB];
                const double* blk = iF + (size_t)bi * SB * SB;
#pragma unroll
                for (int c = 0; c < SB; ++c) xr[c] = blk[c * SB + lane];
                const double bv = ring[(k0 + lane) & (RING - 1)];
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int c = 0; c < SB; c += 4) {
                    a0 = __builtin_fma(xr[c], readlane_f64(bv, c), a0);
                    a1 = __builtin_fma(xr[c + 1], readlane_f64(bv, c + 1), a1);
                    a2 = __builtin_fma(xr[c + 2], readlane_f64(bv, c + 2), a2);
                    a3 = __builtin_fma(xr[c + 3], readlane_f64(bv, c + 3), a3);
                }
                const double val = (a0 + a1) + (a2 + a3);
                xs[lane] = val;
                xv[XI(k0 + lane)] = val;
            } else {
#pragma unroll
                for (int c = 0; c < SB; ++c) {
                    const int d = urow - (k0 + c);
                    pre[c] = (urow <= rend && d <= bw) ? Li[(size_t)(k0 + c) * W + d] : 0.0;
                }
            }
            __syncthreads();
            if (wv == 0) {
                const int rnew = k0 + RING + lane;
                ring[(k0 + lane) & (RING - 1)] = (rnew < nrow) ? xv[XI(rnew)] : 0.0;
            } else if (urow <= rend) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int c = 0; c < SB; c += 2) {
                    s0 = __builtin_fma(pre[c], xs[c], s0);
                    s1 = __builtin_fma(pre[c + 1], xs[c + 1], s1);
                }
                ring[urow & (RING - 1)] -= s0 + s1;
            }
            __syncthreads();
        }
        if (tid < sp.nm) sp_out[tid] = ring[(ncol + tid) & (RING - 1)];  // b_mid - (this side's updates)
    } else {
        if (tid < sp.nm) ring[(ncol + tid) & (RING - 1)] = xv[XI(ncol + tid)];  // solved middle rows
        __syncthreads();
        for (int bi = nblk - 1; bi >= 0; --bi) {
            const int k0 = bi * SB;
            double xc[SB];
            if (wv == 0) {
                const double* blk = iB + (size_t)bi * SB * SB;
#pragma unroll
                for (int r = 0; r < SB; ++r) xc[r] = blk[r * SB + lane];
            }
            {
                const int c = tid >> 2, part = tid & 3;
                const int kc = k0 + c;
                const int dlo = k0 + SB - kc;
                const int dhi = (nrow - 1 - kc < bw) ? (nrow - 1 - kc) : bw;
                double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                for (int d0 = dlo + part; d0 <= dhi; d0 += 32) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int d = d0 + 4 * u;
                        if (d <= dhi) a8[u] = __builtin_fma(Li[(size_t)kc * W + d], ring[(kc + d) & (RING - 1)], a8[u]);
                    }
                }
                ps[c][part] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            }
            __syncthreads();
            if (wv == 0) {
                const double zv = xv[XI(k0 + lane)] - (((ps[lane][0] + ps[lane][1]) + ps[lane][2]) + ps[lane][3]);
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int r = 0; r < SB; r += 4) {
                    a0 = __builtin_fma(xc[r], readlane_f64(zv, r), a0);
                    a1 = __builtin_fma(xc[r + 1], readlane_f64(zv, r + 1), a1);
                    a2 = __builtin_fma(xc[r + 2], readlane_f64(zv, r + 2), a2);
                    a3 = __builtin_fma(xc[r + 3], readlane_f64(zv, r + 3), a3);
                }
                const double val = (a0 + a1) + (a2 + a3);
                ring[(k0 + lane) & (RING - 1)] = val;
                if (acc) acc[ib + XI(k0 + lane)] += val;
                xv[XI(k0 + lane)] = val;
            }
            __syncthreads();
        }
    }
}

// Middle block of the twisted solve: rhs = (top spill) + (bottom spill, reversed) - b_mid, dense
// forward and backward substitution with Lm (nm x nm, staged in LDS), result to x[m .. m+nm).
__global__ __launch_bounds__(256) void adj_mid_solve_kernel(const double* __restrict__ Lm, int M, int N,
                                                            double* __restrict__ x, double* __restrict__ acc,
                                                            const double* __restrict__ spill) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bw = M, n = M * N;
    const AdjSplit sp = adj_split(n, bw);
    const int nm = sp.nm, ld = nm + 1;
    double* D = smem;            // [nm][ld]
    double* v = smem + nm * ld;  // [nm]
    const int img = blockIdx.x, tid = threadIdx.x;
    const size_t ib = (size_t)img * n;
    const double* Lg = Lm + (size_t)img * (bw + ADJ_G) * (bw + ADJ_G);
    const double* st = spill + ((size_t)img * 2) * RING;
    const double* sb = st + RING;
    for (int e = tid; e < nm * nm; e += 256) D[(e / nm) * ld + (e % nm)] = Lg[e];
    for (int r = tid; r < nm; r += 256) v[r] = st[r] + sb[nm - 1 - r] - x[ib + sp.m + r];
    __syncthreads();
    for (int k = 0; k < nm; ++k) {  // forward, column oriented
        const double xk = v[k] / D[k * ld + k];
        __syncthreads();
        if (tid == 0) v[k] = xk;
        for (int r = k + 1 + tid; r < nm; r += 256) v[r] -= D[r * ld + k] * xk;
        __syncthreads();
    }
    for (int k = nm - 1; k >= 0; --k) {  // backward: L^T, uses row k of L
        const double xk = v[k] / D[k * ld + k];
        __syncthreads();
        if (tid == 0) v[k] = xk;
        for (int c = tid; c < k; c += 256) v[c] -= D[k * ld + c] * xk;
        __syncthreads();
    }
    for (int r = tid; r < nm; r += 256) {
        x[ib + sp.m + r] = v[r];
        if (acc) acc[ib + sp.m + r] += v[r];
    }
}

// out = rhs - (I + S G^T W G S) p   (matrix free; the flux of the three pixels that touch a node
// is recomputed instead of being staged, so one pass and no atomics).
__device__ __forceinline__ void adj_flux(const AdjCoef& C, const double* __restrict__ p, size_t q, int i,
                                         int j, int M, int N, double& w1, double& w2) {
    const double pk = C.s[q] * p[q];
    const double d1 = (i < M - 1) ? C.s[q + 1] * p[q + 1] - pk : 0.0;
    const double d2 = (j < N - 1) ? C.s[q + M] * p[q + M] - pk : 0.0;
    const double t1 = C.t1[q], t2 = C.t2[q];
    const double bp = t1 * d1 + t2 * d2;
    const double cb = C.c[q] * bp, kp = C.kap[q];
    w1 = cb * t1 + kp * d1;
    w2 = cb * t2 + kp * d2;
}

__global__ __launch_bounds__(256) void adj_residual_kernel(AdjCoef C, const double* __restrict__ p, int M,
                                                           int N, int O, double* __restrict__ out) {
    const size_t npx = (size_t)M * N;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= npx * O) return;
    const int k = (int)(q % npx);
    const int i = k % M, j = k / M;
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, c1 = 0.0, c2 = 0.0;
    adj_flux(C, p, q, i, j, M, N, a1, a2);
    if (i > 0) adj_flux(C, p, q - 1, i - 1, j, M, N, b1, b2);
    if (j > 0) adj_flux(C, p, q - M, i, j - 1, M, N, c1, c2);
    const double w1 = (i < M - 1) ? a1 : 0.0, w1m = (i > 0) ? b1 : 0.0;
    const double w2 = (j < N - 1) ? a2 : 0.0, w2m = (j > 0) ? c2 : 0.0;
    const double gt = (w1m - w1) + (w2m - w2);
    out[q] = C.rhs[q] - (p[q] + C.s[q] * gt);
}

// Per-pixel gradient contribution (TVLearningFunctionVec.jl:133-134, :158, :213, :250).
// p holds q of the scaled system; the physical adjoint state is S q.
__global__ __launch_bounds__(256) void adj_gradpix_kernel(AdjCoef C, const double* __restrict__ p, int M,
                                                          int N, int O, int patch, int reg,
                                                          double* __restrict__ gpix) {
    const size_t npx = (size_t)M * N;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= npx * O) return;
    const int k = (int)(q % npx);
    const int i = k % M, j = k / M;
    const double pk = C.s[q] * p[q];
    if (!(reg && patch)) {
        const double d1 = (i < M - 1) ? C.s[q + 1] * p[q + 1] - pk : 0.0;
        const double d2 = (j < N - 1) ? C.s[q + M] * p[q + M] - pk : 0.0;
        const double v = d1 * C.h1[q] + d2 * C.h2[q];
        gpix[q] = reg ? v : -v;
    } else {
        const double a = (i < M - 1) ? C.h1[q] : 0.0, am_ = (i > 0) ? C.h1[q - 1] : 0.0;
        const double b = (j < N - 1) ? C.h2[q] : 0.0, bm = (j > 0) ? C.h2[q - M] : 0.0;
        gpix[q] = pk * ((am_ - a) + (bm - b));
    }
}

// calc_adjoint(PatchOp, .): partial[(pa + am*pb)*O + k] = sum of image k's pixel contributions over
// the patch.  grid (am*an, O); sum_final_kernel then adds the O images of each patch in image order
// (fixed order -> bitwise reproducible, and every image/patch pair gets its own workgroup).
__global__ __launch_bounds__(256) void patch_sum_kernel(const double* __restrict__ gpix, int M, int N, int O,
                                                        int am, int an, double* __restrict__ partial) {
    __shared__ double sh[4];
    const int pa = blockIdx.x % am, pb = blockIdx.x / am, k = blockIdx.y;
    // pixels i with (i*am)/M == pa  <=>  i in [ceil(pa*M/am), ceil((pa+1)*M/am))
    const int i0 = (int)(((long)pa * M + am - 1) / am), i1 = (int)(((long)(pa + 1) * M + am - 1) / am);
    const int j0 = (int)(((long)pb * N + an - 1) / an), j1 = (int)(((long)(pb + 1) * N + an - 1) / an);
    const int wi = i1 - i0, wj = j1 - j0;
    const int cnt = wi * wj;
    const double* g = gpix + (size_t)k * M * N;
    double s = 0.0;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int i = i0 + e % wi, j = j0 + e / wi;
        s += g[i + (size_t)M * j];
    }
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * O + k] = s;
}

// Residual statistics per image: out[k*4 + {0: ||r||^2, 1: ||rhs||^2, 2: r^T D^-1 r, 3: rhs^T D^-1 rhs}],
// D = diag(A) (plane 0 of band4).  The diagonally scaled pair is the quality gate: the raw residual is
// dominated by the rounding of the rows that carry the 1e14 active-set weight (|r_k| ~ eps * 1e14 * |p|),
// which say nothing about the solve; scaled by 1/sqrt(d_k) = 1e-7 they fall to rounding level.
// grid (RESN_BLK, O): partials[(img * RESN_BLK + b) * 4 + {0..3}]; adj_resnorm_final_kernel adds them in block order
// (no atomics: reproducible) into out[img * 4 + {0..3}].
constexpr int RESN_BLK = 64;
__global__ __launch_bounds__(256) void adj_resnorm_kernel(const double* __restrict__ r,
                                                          const double* __restrict__ rhs,
                                                          const double* __restrict__ diag, int npx,
                                                          double* __restrict__ partial) {
    __shared__ double sh[4];
    const size_t base = (size_t)blockIdx.y * npx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < (size_t)npx; q += (size_t)RESN_BLK * 256) {
        const double rv = r[base + q], bv = rhs[base + q], di = 1.0 / diag[base + q];
        s0 += rv * rv;
        s1 += bv * bv;
        s2 += rv * rv * di;
        s3 += bv * bv * di;
    }
    s0 = block_sum<256>(s0, sh);
    s1 = block_sum<256>(s1, sh);
    s2 = block_sum<256>(s2, sh);
    s3 = block_sum<256>(s3, sh);
    if (threadIdx.x == 0) {
        double* o = partial + 4 * ((size_t)blockIdx.y * RESN_BLK + blockIdx.x);
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
}
__global__ void adj_resnorm_final_kernel(const double* __restrict__ partial, int O, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // (img, component)
    if (e >= 4 * O) return;
    const int img = e >> 2, c = e & 3;
    double s = 0.0;
    for (int b = 0; b < RESN_BLK; ++b) s += partial[4 * ((size_t)img * RESN_BLK + b) + c];
    out[e] = s;
}

}  // namespace bpltv
