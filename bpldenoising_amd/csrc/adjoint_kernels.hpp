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

// Banded Cholesky A = L L^T, bandwidth bw = M, one workgroup (1024 threads) per image.
// LDS: ring of W = M+1 columns x W rows (the trailing window) + the current column.
// L is written to global memory as [O][n][W] (column k: L[k+d][k] at d).
__global__ __launch_bounds__(1024) void adj_factor_kernel(const double* __restrict__ band4, int M,
                                                          int N, int O, double* __restrict__ L,
                                                          int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int W = M + 1, bw = M;
    double* win = smem;          // [W][W]
    double* lv = smem + W * W;   // [W]
    const int img = blockIdx.x;
    const int n = M * N;
    const size_t tot = (size_t)n * O;
    const size_t ib = (size_t)img * n;
    const int tid = threadIdx.x;
    for (int e = tid; e < W * W; e += 1024) {
        const int c = e / W, d = e - c * W;
        win[e] = (c < n) ? band_init(band4, tot, ib + c, d, M) : 0.0;
    }
    __syncthreads();
    const int g = tid >> 7, qq = tid & 127;
    for (int k = 0; k < n; ++k) {
        const int s = k % W;
        const int lim = (n - 1 - k < bw) ? (n - 1 - k) : bw;
        const double piv = win[s * W];
        if (!(piv > 0.0)) {
            if (tid == 0) fail[img] = k + 1;
            return;  // uniform: every thread reads the same pivot
        }
        const double d = sqrt(piv);
        const double inv = 1.0 / d;
        for (int r = tid; r <= bw; r += 1024) {
            double v = 0.0;
            if (r == 0) v = d;
            else if (r <= lim) v = win[s * W + r] * inv;
            lv[r] = v;
            L[(ib + k) * W + r] = v;
        }
        __syncthreads();
        // recycle slot s for column k + W
        for (int r = tid; r <= bw; r += 1024)
            win[s * W + r] = (k + W < n) ? band_init(band4, tot, ib + k + W, r, M) : 0.0;
        // trailing update: win[col k+c][r-c] -= l_r * l_c, 1 <= c <= r <= lim
        for (int c = 1 + g; c <= lim; c += 8) {
            const double lc = lv[c];
            double* col = win + ((k + c) % W) * W;
            for (int o = qq; c + o <= lim; o += 128) col[o] -= lv[c + o] * lc;
        }
        __syncthreads();
    }
    if (tid == 0) fail[img] = 0;
}

// Solve L L^T x = b in place (x: [O][n]), blocked by 64 columns; one workgroup of 256 per image.
// If `acc` is non-null the result is added to acc (p += dp) instead of being left in x.
__global__ __launch_bounds__(256) void adj_solve_kernel(const double* __restrict__ L, int M, int N,
                                                        double* __restrict__ x,
                                                        double* __restrict__ acc) {
    constexpr int B = 64;
    __shared__ double xs[B];
    __shared__ double ps[B][4];
    const int W = M + 1, bw = M;
    const int n = M * N;
    const size_t ib = (size_t)blockIdx.x * n;
    const double* Li = L + ib * W;
    double* xv = x + ib;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // ---- forward: L y = b
    for (int k0 = 0; k0 < n; k0 += B) {
        const int nb = (n - k0 < B) ? (n - k0) : B;
        if (wv == 0) {
            double val = (lane < nb) ? xv[k0 + lane] : 0.0;
            for (int c = 0; c < nb; ++c) {
                const double dc = Li[(size_t)(k0 + c) * W];
                const double v = __shfl(val, c, 64) / dc;
                if (lane == c) val = v;
                if (lane > c && lane < nb && lane - c <= bw) val -= Li[(size_t)(k0 + c) * W + (lane - c)] * v;
            }
            if (lane < nb) { xs[lane] = val; xv[k0 + lane] = val; }
        }
        __syncthreads();
        // rows beyond the block: r in [k0+nb, k0+nb-1+bw]
        const int rend = (k0 + nb - 1 + bw < n - 1) ? (k0 + nb - 1 + bw) : (n - 1);
        for (int r = k0 + nb + tid; r <= rend; r += 256) {
            double sacc = 0.0;
            for (int c = 0; c < nb; ++c) {
                const int d = r - (k0 + c);
                if (d <= bw) sacc += Li[(size_t)(k0 + c) * W + d] * xs[c];
            }
            xv[r] -= sacc;
        }
        __syncthreads();
    }
    // ---- backward: L^T x = y
    const int nblk = (n + B - 1) / B;
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int k0 = bi * B;
        const int nb = (n - k0 < B) ? (n - k0) : B;
        // contributions of already-solved entries beyond the block: column c, rows k0+nb .. k0+c+bw
        {
            const int c = tid >> 2, part = tid & 3;  // 64 columns x 4 partial sums
            double sacc = 0.0;
            if (c < nb) {
                const int kc = k0 + c;
                const int dlo = k0 + nb - kc;  // first offset beyond the block (>= 1)
                const int dhi = (n - 1 - kc < bw) ? (n - 1 - kc) : bw;
                for (int d = dlo + part; d <= dhi; d += 4) sacc += Li[(size_t)kc * W + d] * xv[kc + d];
            }
            ps[c][part] = sacc;
        }
        __syncthreads();
        if (wv == 0) {
            double val = 0.0;
            if (lane < nb) val = xv[k0 + lane] - (((ps[lane][0] + ps[lane][1]) + ps[lane][2]) + ps[lane][3]);
            for (int c = nb - 1; c >= 0; --c) {
                const double dc = Li[(size_t)(k0 + c) * W];
                const double v = __shfl(val, c, 64) / dc;
                if (lane == c) val = v;
                // x_lane (lane < c) loses L[k0+c][k0+lane] * x_c = column (k0+lane), offset c-lane
                if (lane < c && c - lane <= bw) val -= Li[(size_t)(k0 + lane) * W + (c - lane)] * v;
            }
            if (lane < nb) {
                if (acc) acc[ib + k0 + lane] += val;
                xv[k0 + lane] = val;
            }
        }
        __syncthreads();
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

// calc_adjoint(PatchOp, .) summed over images: out[pa + am*pb] = sum over the patch's pixels and
// all O images.  One workgroup per patch; fixed order -> reproducible.
__global__ __launch_bounds__(256) void patch_sum_kernel(const double* __restrict__ gpix, int M, int N, int O,
                                                        int am, int an, double* __restrict__ out) {
    __shared__ double sh[4];
    const int pa = blockIdx.x % am, pb = blockIdx.x / am;
    // pixels i with (i*am)/M == pa  <=>  i in [ceil(pa*M/am), ceil((pa+1)*M/am))
    const int i0 = (int)(((long)pa * M + am - 1) / am), i1 = (int)(((long)(pa + 1) * M + am - 1) / am);
    const int j0 = (int)(((long)pb * N + an - 1) / an), j1 = (int)(((long)(pb + 1) * N + an - 1) / an);
    const int wi = i1 - i0, wj = j1 - j0;
    const long cnt = (long)wi * wj * O;
    double s = 0.0;
    for (long e = threadIdx.x; e < cnt; e += 256) {
        const int i = i0 + (int)(e % wi);
        const long r = e / wi;
        const int j = j0 + (int)(r % wj);
        const int k = (int)(r / wj);
        s += gpix[(size_t)k * M * N + i + (size_t)M * j];
    }
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ||r||^2 and ||rhs||^2 per image for the residual statistic: partial[k*2 + {0,1}].
__global__ __launch_bounds__(256) void adj_resnorm_kernel(const double* __restrict__ r,
                                                          const double* __restrict__ rhs, int npx,
                                                          double* __restrict__ out) {
    __shared__ double sh[4];
    const size_t base = (size_t)blockIdx.x * npx;
    double s0 = 0.0, s1 = 0.0;
    for (int q = threadIdx.x; q < npx; q += 256) {
        s0 += r[base + q] * r[base + q];
        s1 += rhs[base + q] * rhs[base + q];
    }
    s0 = block_sum<256>(s0, sh);
    s1 = block_sum<256>(s1, sh);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = s0; out[2 * blockIdx.x + 1] = s1; }
}

}  // namespace bpltv
