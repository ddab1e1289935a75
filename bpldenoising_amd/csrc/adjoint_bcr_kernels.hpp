// adjoint_bcr_kernels.hpp -- adjoint solve by block cyclic reduction (M <= 128).
//
// The reduced SPD system of adjoint_kernels.hpp is block tridiagonal: N diagonal blocks D_j (one image
// column of M pixels each, tridiagonal) coupled by C_j = A[block j+1, block j] (bidiagonal).  A banded
// Cholesky walks its M*N columns one after the other on one or two workgroups per image
// (adj_factor_kernel); here the blocks are eliminated in odd-even (nested dissection) order instead:
// level l (stride s = 2^l) eliminates the blocks j = s + 2s*t, all of them -- and all images -- at
// once, leaving a block tridiagonal system on the blocks 2s*t.  That is a block Cholesky of the
// permuted matrix (backward stable for an SPD matrix), log2(N) levels deep, and every level is
// made of dense MP x MP operations (MP = M rounded up to 16) that fill the chip:
//
//   bcr_potrf_kernel   L_j = chol(D_j) and its inverse, in LDS, 16x16 tiles on the f64 MFMA
//                      (left-looking; the diagonal tile in wave-0 registers with readlane broadcasts)
//   bcr_x_kernel       XA_j = L_j^-1 C_a,  XB_j = L_j^-1 C_j^T            (a = j-s, b = j+s)
//   bcr_upd_kernel     D_a -= XA_j^T XA_j (+ XB_j'^T XB_j' from the other side),  C_a = -XB_j^T XA_j
//
// (64x64 output tiles, v_mfma_f64_16x16x4_f64, operands staged through LDS in 32-deep chunks).
// The substitutions walk the same levels with matrix-vector products:
//   forward  z_j = L_j^-1 r_j ;  r_a -= XA_j^T z_j ;  r_b -= XB_j^T z_j
//   backward p_j = L_j^-T (z_j - XA_j p_a - XB_j p_b)
// Both orientations of L^-1, XA, XB are stored so that every product reads coalesced columns.
// Explicit inverses of the (triangular) diagonal factors are accurate enough here because the
// solve is refined iteratively against the matrix-free operator (round-1 numpy prototype: the
// gradient agrees with the banded solve to 4e-9).
//
// Block storage: column major, leading dimension MP; element (r, c) at r + MP*c; block j of image
// k at ((k*N + j) * MP*MP).  Padding rows/columns (r >= M) carry the identity.
#pragma once
#include <hip/hip_runtime.h>
#include "adjoint_kernels.hpp"

namespace bpltv {

typedef double bcr_d4 __attribute__((ext_vector_type(4)));

#ifdef BCR_PROBE_ON   // tools/potrf_probe.hip only: time stamps (s_memrealtime, 100 MHz) of workgroup 0's phases
__device__ long long bcr_probe_buf[64];
__device__ int bcr_probe_on = 1;   // tools set it per launch range (e.g. only the chain kernel of one mid-run panel)
#define BCR_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && bcr_probe_on) bcr_probe_buf[i] = (long long)wall_clock64(); } while (0)
// span of a launch over all its workgroups: [kind][panel & 63] = {earliest start, latest end}
__device__ unsigned long long bcr_span[4][64][2];
#define BCR_SPAN_BEGIN(kind, panel) do { if (threadIdx.x == 0) atomicMin(&bcr_span[kind][(panel) & 63][0], (unsigned long long)wall_clock64()); } while (0)
#define BCR_SPAN_END(kind, panel) do { if (threadIdx.x == 0) atomicMax(&bcr_span[kind][(panel) & 63][1], (unsigned long long)wall_clock64()); } while (0)
#define BCR_SPAN_RESET(kind, panel) do { if (threadIdx.x == 0 && blockIdx.x == 0) { bcr_span[kind][((panel) + 32) & 63][0] = ~0ull; bcr_span[kind][((panel) + 32) & 63][1] = 0ull; } } while (0)
#else
#define BCR_PROBE(i) do { } while (0)
#define BCR_SPAN_BEGIN(kind, panel) do { } while (0)
#define BCR_SPAN_END(kind, panel) do { } while (0)
#define BCR_SPAN_RESET(kind, panel) do { } while (0)
#endif

#ifdef BCR_DBG   // timing experiments of tools/bcr_unit.hip only (results are wrong when set)
__device__ int bcr_dbg = 0;   // 1 skip operand global loads, 2 skip MFMAs, 4 skip LDS staging, 8 skip the output phase
#define BCR_DBG_ON(bit) (bcr_dbg & (bit))
#else
#define BCR_DBG_ON(bit) false
#endif

__host__ __device__ inline int bcr_levels(int N) {
    int L = 0;
    while ((1 << L) < N) ++L;
    return L;
}
// blocks eliminated at level l: j = s + 2s*t < N
__host__ __device__ inline int bcr_nelim(int N, int l) {
    const int s = 1 << l;
    return N > s ? (N - s + 2 * s - 1) / (2 * s) : 0;
}
// blocks that survive level l: a = 2s*t < N
__host__ __device__ inline int bcr_nsurv(int N, int l) {
    const int s = 1 << l;
    return (N + 2 * s - 1) / (2 * s);
}

__device__ __forceinline__ bcr_d4 bcr_mfma(double a, double b, bcr_d4 c) {
    // v_mfma_f64_16x16x4_f64: lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15];
    // result register g of lane l is C[row = (l>>4) + 4g][col = l&15].
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Dense blocks from the four assembled diagonals.  grid (N, O), block 256.
__global__ __launch_bounds__(256) void bcr_init_kernel(const double* __restrict__ band4, int M, int N, int O,
                                                       int MP, double* __restrict__ D, double* __restrict__ C) {
    const int j = blockIdx.x, img = blockIdx.y;
    const size_t npx = (size_t)M * N, tot = npx * O;
    const size_t q0 = (size_t)img * npx + (size_t)j * M;
    const size_t bo = ((size_t)img * N + j) * MP * MP;
    for (int e = threadIdx.x; e < MP * MP; e += 256) {
        const int r = e % MP, c = e / MP;
        double d = 0.0, cc = 0.0;
        if (r < M && c < M) {
            if (r == c) d = band4[q0 + r];
            else if (r == c + 1) d = band4[tot + q0 + c];
            else if (c == r + 1) d = band4[tot + q0 + r];
            if (j + 1 < N) {
                if (r == c) cc = band4[3 * tot + q0 + c];            // A[q+M, q]
                else if (r == c - 1) cc = band4[2 * tot + q0 + c];   // A[q+M-1, q]
            }
        } else if (r == c) {
            d = 1.0;
        }
        D[bo + e] = d;
        C[bo + e] = cc;
    }
}

// Inverse of the lower triangular 16x16 tile T (LDS, leading dimension ld) in place, by one wave: lane c computes
// column c of W = L^-1 by forward substitution; L(r, k) is read from the tile as an LDS broadcast, 1 / L(r, r) from
// di[r].  On exit the tile holds W (lower triangular, explicit zeros above).
__device__ __forceinline__ void bcr_tile_inverse(double* __restrict__ T, int ld, int lane, const double* __restrict__ di) {
    const int lr = lane & 15;
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double acc = (lr == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) acc = __builtin_fma(-T[r + ld * k], x[k], acc);
        x[r] = acc * di[r];
        asm volatile("" : "+v"(x[r]) : : "memory");  // row by row: keeps the 120 LDS loads from being hoisted into 240 live registers
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r + ld * lane] = x[r];
    }
}

// Tile (p, q), q < p, of W = L^-1:  W_pq = -W_pp sum_{k=q}^{p-1} L_pk W_kq, by one wave on the MFMA.
// W_kq (k >= q) is kept in the upper tile (q, k): element (r', c') at S[(16q+r') + ld(16k+c')].
__device__ __forceinline__ void bcr_winv_tile(double* __restrict__ S, int ld, int p, int q, int lr, int lk) {
    bcr_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k = q; k < p; ++k) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double a = S[(16 * p + lr) + ld * (16 * k + 4 * kk + lk)];      // L_pk(r, k')
            const double b = S[(16 * q + 4 * kk + lk) + ld * (16 * k + lr)];      // W_kq(k', col)
            acc = bcr_mfma(a, b, acc);
        }
    }
    bcr_d4 out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double a = S[(16 * p + lr) + ld * (16 * p + 4 * kk + lk)];          // W_pp(r, k')
        out = bcr_mfma(-a, acc[kk], out);  // the accumulator tile is the B operand of k-step kk
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) S[(16 * q + lk + 4 * g) + ld * (16 * p + lr)] = out[g];
}

// Block column p (16 columns, rows 16p .. MP-1, already updated with the columns to its left) of the matrix in S,
// factored by one wave in registers: lane l holds row 16p + l (and row 16p + 64 + l when TWO: more than 64 rows).
// Writes L back (zeros above the diagonal of the diagonal tile) and 1 / L(r, r) to dinv.
// nact < 16: only the first nact columns of the block column are real, the others are identity padding (the fronts of
// the nested-dissection solver pad their pivot block to 16): their pivot chains are skipped.
template <bool TWO>
__device__ __forceinline__ bool bcr_panel_factor(double* __restrict__ S, int ld, int MP, int p, int lane,
                                                 double* __restrict__ dinv, int nact = 16) {
    const int r0 = 16 * p + lane, r1 = r0 + 64;
    const bool v0 = r0 < MP, v1 = TWO && r1 < MP;
    double m0[16], m1[16], dv[16], dd[16];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        m0[c] = v0 ? S[r0 + ld * (16 * p + c)] : 0.0;
        if (TWO) m1[c] = v1 ? S[r1 + ld * (16 * p + c)] : 0.0;
    }
    // Left-looking inside the block column, written in the order the dependent chain wants: column q receives its
    // last term (from column q-1), its pivot goes into the rsqrt chain, and while that runs column q+1 is brought up
    // to date with the columns 0..q-1.  Every multiplier L(16p + q, c) = readlane(column c, q) is used at once.
    // (Reading the off-chain multipliers back from LDS instead -- each finished column stored at once -- was slower:
    // 2.5 us against 2.0 us per block column; the wait for the store sits on the chain.)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (q >= nact) {   // identity column: pivot 1, nothing to eliminate (wave-uniform)
            dd[q] = 1.0; dv[q] = 1.0;
            continue;
        }
        if (q > 0) {
            const double l = readlane_f64(m0[q - 1], q);
            m0[q] = __builtin_fma(-m0[q - 1], l, m0[q]);
        }
        const double piv = readlane_f64(m0[q], q);
        if (!(piv > 0.0)) bad = true;
        sqrt_rsqrt(piv, dd[q], dv[q]);
        if (q + 1 < 16 && q + 1 < nact) {
#pragma unroll
            for (int c = 0; c < q; ++c) {
                const double l = readlane_f64(m0[c], q + 1);
                m0[q + 1] = __builtin_fma(-m0[c], l, m0[q + 1]);
            }
        }
        m0[q] *= dv[q];   // the scaled column (its diagonal entry, the sqrt, is merged in at the write-back)
    }
    // the diagonal tile and the rows below it (first 64 rows of the block column) back to LDS
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (v0) S[r0 + ld * (16 * p + c)] = (lane < 16 && c >= lane) ? ((c == lane) ? dd[c] : 0.0) : m0[c];
    // The rows 64.. do not feed the pivots: they follow in a pass of their own with the finished tile, its entries
    // L(16p + q, c) read from LDS as broadcast loads (right-looking, 15 - c independent updates per column).
    // Interleaved with the chain above -- or fed by readlane here -- this pass cost as much as the chain itself
    // (an SGPR broadcast costs ~13 cycles per instruction around it): 4.1 -> 3.1 us per block column.
    if (TWO) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double* Lt = S + 16 * p + ld * (16 * p);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            m1[c] *= dv[c];
#pragma unroll
            for (int q = c + 1; q < 16; ++q) m1[q] = __builtin_fma(-m1[c], Lt[q + ld * c], m1[q]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (v1) S[r1 + ld * (16 * p + c)] = m1[c];
    }
    if (lane < 16) {
        double mine = dv[0];
#pragma unroll
        for (int c = 1; c < 16; ++c) mine = (lane == c) ? dv[c] : mine;
        dinv[16 * p + lane] = mine;
    }
    return bad;
}

// Cholesky factor and its inverse of the symmetric MP x MP matrix held in LDS (S[r + (MP+1) c], lower triangle read),
// by the BCR_PT threads of the workgroup; S is followed by MP doubles of scratch (bcr_potrf_lds).  On exit the lower
// tiles of S hold L (strictly below the diagonal tiles) and the diagonal + upper tiles hold W = L^-1: W(r, c), r >= c,
// at S[(16*(c>>4) + (r&15)) + ld*(16*(r>>4) + (c&15))].  Returns true (in wave 0) on a non-positive pivot.
//
// The dependent chain is MP pivots long; everything else is kept off it.  Per block column p (16 columns):
//   (1) all waves: tile (i, p) -= L_i,p-1 L_p,p-1^T on the MFMA -- only the newest block column; the older ones were
//       applied right-looking by the waves 2.. while wave 0 factored (2)
//   (2) wave 0: the whole block column -- up to 128 rows, lane l holds rows 16p + l and 16p + 64 + l -- is factored
//       in registers: per pivot one rsqrt chain, then rank-1 updates with the pivot column's entries broadcast by
//       readlane.  The rows below the diagonal tile come out as L directly (no product with an inverted diagonal tile).
//       Meanwhile wave 1 inverts diagonal tile p-1 and waves 2.. compute row p-2 of W (bcr_winv_tile): the inverse
//       trails the factorisation by two block columns and costs the chain nothing but a short tail.
constexpr int BCR_PT = 512;
__device__ __forceinline__ bool bcr_potrf_lds_body(double* __restrict__ S, int MP) {
    const int ld = MP + 1, P = MP >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    constexpr int NW = BCR_PT / 64;
    double* dinv = S + (size_t)ld * MP;   // 1 / L(r, r)
    bool bad = false;
    for (int p = 0; p < P; ++p) {
        // (1) all waves: block column p receives the contribution of block column p-1 -- the only one still missing
        //     (4 MFMAs per tile; the older ones were applied behind the panel factorisations, see (2))
        if (p > 0) {
            for (int i = p + wave; i < P; i += NW) {
                bcr_d4 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = S[(16 * i + lk + 4 * g) + ld * (16 * p + lr)];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double a = S[(16 * i + lr) + ld * (16 * (p - 1) + 4 * kk + lk)];
                    const double b = S[(16 * p + lr) + ld * (16 * (p - 1) + 4 * kk + lk)];
                    acc = bcr_mfma(-a, b, acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * p + lr)] = acc[g];
            }
            __syncthreads();
        }
        BCR_PROBE(8 + 2 * p);
        if (wave == 0) {
            // (2) block column p in registers
            if (MP - 16 * p > 64) bad |= bcr_panel_factor<true>(S, ld, MP, p, lane, dinv);
            else bad |= bcr_panel_factor<false>(S, ld, MP, p, lane, dinv);
        } else if (wave == 1) {
            if (p >= 1) bcr_tile_inverse(S + 16 * (p - 1) + ld * (16 * (p - 1)), ld, lane, dinv + 16 * (p - 1));
        } else {
            if (p >= 2)
                for (int q = wave - 2; q < p - 2; q += NW - 2) bcr_winv_tile(S, ld, p - 2, q, lr, lk);
            // right-looking, off the chain: block column p-1 applied to the lower tiles (i, j), j > p, of the rest
            if (p >= 1) {
                const int m = P - p - 1;   // tile rows / columns p+1 .. P-1
                for (int t = wave - 2; t < m * (m + 1) / 2; t += NW - 2) {
                    int a = 0, c = t;
                    while (c > a) { c -= a + 1; ++a; }
                    const int i = p + 1 + a, j = p + 1 + c;
                    bcr_d4 acc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = S[(16 * i + lk + 4 * g) + ld * (16 * j + lr)];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double x = S[(16 * i + lr) + ld * (16 * (p - 1) + 4 * kk + lk)];
                        const double y = S[(16 * j + lr) + ld * (16 * (p - 1) + 4 * kk + lk)];
                        acc = bcr_mfma(-x, y, acc);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * j + lr)] = acc[g];
                }
            }
        }
        __syncthreads();
        BCR_PROBE(9 + 2 * p);
    }
    // tail: the last diagonal tile and the last two rows of W
    if (wave == 0) bcr_tile_inverse(S + 16 * (P - 1) + ld * (16 * (P - 1)), ld, lane, dinv + 16 * (P - 1));
    else if (wave >= 2 && P >= 2)
        for (int q = wave - 2; q < P - 2; q += NW - 2) bcr_winv_tile(S, ld, P - 2, q, lr, lk);
    __syncthreads();
    for (int q = wave; q < P - 1; q += NW) bcr_winv_tile(S, ld, P - 1, q, lr, lk);
    __syncthreads();
    return bad;
}

// Cholesky factor of D_j and its inverse, in LDS.  grid (nelim, O) (or (1, O) with s = 0 for the
// last block 0), block BCR_PT; dynamic LDS bcr_potrf_lds(MP).  On exit D_j holds L^-1 (lower
// triangular, zeros above) and DT_j its transpose.
__global__ __launch_bounds__(BCR_PT) void bcr_potrf_kernel(double* __restrict__ D, double* __restrict__ DT, int N,
                                                           int MP, int s, int* __restrict__ fail) {
    extern __shared__ double S[];
    const int ld = MP + 1;
    const int img = blockIdx.y;
    const int j = (s == 0) ? 0 : s + 2 * s * (int)blockIdx.x;
    const size_t bo = ((size_t)img * N + j) * MP * MP;
    double* Dj = D + bo;
    double* DTj = DT + bo;
    const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll 8
    for (int e = tid; e < MP * MP; e += BCR_PT) {
        const int r = e % MP, c = e / MP;
        S[r + ld * c] = Dj[e];
    }
    __syncthreads();
    const bool bad = bcr_potrf_lds_body(S, MP);
    if (bad && lane == 0 && fail[img] == 0) fail[img] = j + 1;
    // entry e = r + MP c of either array is W(hi, lo), hi = max(r, c), or zero: one LDS read per entry, in batches of
    // 8 reads followed by their stores (a rolled loop pays the LDS latency once per entry)
    for (int e0 = tid; e0 < MP * MP; e0 += 8 * BCR_PT) {
        double w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = min(e0 + i * BCR_PT, MP * MP - 1), r = e % MP, c = e / MP;
            const int hi = r > c ? r : c, lo = r > c ? c : r;
            w[i] = S[(16 * (lo >> 4) + (hi & 15)) + ld * (16 * (hi >> 4) + (lo & 15))];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * BCR_PT, r = e % MP, c = e / MP;
            if (e < MP * MP) {
                Dj[e] = (r >= c) ? w[i] : 0.0;    // L^-1 (r, c)
                DTj[e] = (c >= r) ? w[i] : 0.0;   // L^-T (r, c) = L^-1 (c, r)
            }
        }
    }
}

// ---- 64x64 output tile of sum_prod A B on the f64 MFMA ------------------------------------------
constexpr int BG_T = 256;   // 4 waves, each a 32x32 quarter of the tile
constexpr int BG_KC = 32;   // depth of one staged chunk
constexpr int BG_LD = 65;

// 64 x BG_KC chunk of Op(rr, kk), rr in [r0, r0+64), kk in [k0, k0+BG_KC) -> v[8] (zero outside MP).
// RFAST: element (rr, kk) at rr + MP*kk; otherwise at kk + MP*rr.  Staged as Ls[kk][rr] with a row
// stride of BG_LD = 65 doubles: the 16 lanes of an MFMA operand row read 16 consecutive doubles.
// (An interleaved [kk/4][rr][kk%4] layout and whole-block 128x128 workgroups were measured too: no
// gain resp. 1.5x slower; PMC: MFMA pipe 29 % busy, waves 52 % waiting -- the products are bound by
// the latency of the operand panels streaming from HBM, not by LDS or MFMA issue.)
template <bool RFAST>
__device__ __forceinline__ void bg_fetch(const double* __restrict__ base, int MP, int r0, int k0, int tid,
                                         double (&v)[8]) {
    if (RFAST) {
        const int r = r0 + (tid & 63), kq = k0 + (tid >> 6);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kq + 4 * i;
            v[i] = (r < MP && k < MP) ? base[r + (size_t)MP * k] : 0.0;
        }
    } else {
        const int k = k0 + (tid & 31), rq = r0 + (tid >> 5);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = rq + 8 * i;
            v[i] = (r < MP && k < MP) ? base[k + (size_t)MP * r] : 0.0;
        }
    }
}
template <bool RFAST>
__device__ __forceinline__ void bg_stage(double* __restrict__ Ls, int tid, const double (&v)[8]) {
    if (RFAST) {
        const int r = tid & 63, kq = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) Ls[(kq + 4 * i) * BG_LD + r] = v[i];
    } else {
        const int k = tid & 31, rq = tid >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) Ls[k * BG_LD + rq + 8 * i] = v[i];
    }
}

struct BgAcc { bcr_d4 c[2][2]; };

// acc += A0 B0 (+ A1 B1 when nprod = 2) over the depth [0, kend).  ARF: A(r,k) at r + MP k (else k + MP r);
// BCF: B(k,c) at c + MP k (else k + MP c).  lds: 2 * BG_KC * BG_LD doubles.  The global loads of chunk
// i+1 are issued before the MFMAs of chunk i and land in registers while those run.
template <bool ARF, bool BCF>
__device__ __forceinline__ void bg_product(const double* __restrict__ A0, const double* __restrict__ B0,
                                           const double* __restrict__ A1, const double* __restrict__ B1, int nprod,
                                           int MP, int r0, int c0, int kend, double* __restrict__ lds, BgAcc& acc) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    const int nk = (kend + BG_KC - 1) / BG_KC, nchunk = nk * nprod;
    double va[8], vb[8];
    bg_fetch<ARF>(A0, MP, r0, 0, tid, va);
    bg_fetch<BCF>(B0, MP, c0, 0, tid, vb);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();  // the previous chunk has been consumed
        if (!BCR_DBG_ON(4)) {
            bg_stage<ARF>(As, tid, va);
            bg_stage<BCF>(Bs, tid, vb);
        }
        __syncthreads();
        if (ch + 1 < nchunk && !BCR_DBG_ON(1)) {
            const int nx = ch + 1;
            const bool second = nx >= nk;
            const int k0 = (second ? nx - nk : nx) * BG_KC;
            bg_fetch<ARF>(second ? A1 : A0, MP, r0, k0, tid, va);
            bg_fetch<BCF>(second ? B1 : B0, MP, c0, k0, tid, vb);
        }
        if (BCR_DBG_ON(2)) continue;
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
}

// accumulators -> LDS tile T[c*BG_LD + r] (64 x 64), after all waves are done with the staging buffers
__device__ __forceinline__ void bg_to_lds(const BgAcc& acc, double* __restrict__ T) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                T[(wc + 16 * tj + lr) * BG_LD + wr + 16 * ti + lk + 4 * g] = acc.c[ti][tj][g];
    __syncthreads();
}

constexpr int BG_LDS = 64 * BG_LD;  // doubles: staging (2 x 32 x 65) and the output tile (64 x 65) share it

// Workgroup id -> (product, tile).  Consecutive workgroup ids go round-robin over the 8 XCDs, each with
// its own L2; the nt*nt tiles of one product re-read the same operand panels, so they are given ids of
// equal residue mod 8 (same XCD, dispatched together) and the second read of a panel hits that L2
// instead of HBM.  Launch bg_grid(P, nt) workgroups; returns false for the padding ids.
__host__ __device__ inline unsigned bg_grid(unsigned P, unsigned nt) { return ((P + 7) / 8) * 8 * nt * nt; }
__device__ __forceinline__ bool bg_decode(unsigned id, unsigned P, int nt, unsigned& prod, int& r0, int& c0) {
    const unsigned G = (unsigned)(nt * nt), span = 8 * G;
    const unsigned q = id / span, rem = id - q * span;
    const unsigned t = rem >> 3;
    prod = 8 * q + (rem & 7);
    r0 = (int)(t % nt) * 64;
    c0 = (int)(t / nt) * 64;
    return prod < P;
}

// XA_j = Linv_j C_a,  XB_j = Linv_j C_j^T, both orientations stored.
// grid bg_grid(2*nelim*O, nt), nt = ceil(MP/64); block BG_T.
__global__ __launch_bounds__(BG_T) __attribute__((amdgpu_waves_per_eu(4))) void bcr_x_kernel(const double* __restrict__ Linv, const double* __restrict__ C,
                                                     double* __restrict__ XA, double* __restrict__ XAT,
                                                     double* __restrict__ XB, double* __restrict__ XBT, int N,
                                                     int O, int MP, int s) {
    __shared__ __attribute__((aligned(32))) double lds[BG_LDS];
    const int nt = (MP + 63) / 64;
    const unsigned ny = 2 * (unsigned)bcr_nelim(N, 31 - __builtin_clz(s));
    unsigned prod;
    int r0, c0;
    if (!bg_decode(blockIdx.x, ny * (unsigned)O, nt, prod, r0, c0)) return;
    const int yy = (int)(prod % ny), img = (int)(prod / ny);
    const int which = yy & 1, t = yy >> 1;
    const int j = s + 2 * s * t, a = j - s, b = j + s;
    if (which == 1 && b >= N) return;
    const size_t bsz = (size_t)MP * MP, ib = (size_t)img * N;
    const double* Lj = Linv + (ib + j) * bsz;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    int kend = r0 + 64;  // Linv is lower triangular: columns beyond the tile's last row are zero
    if (kend > MP) kend = MP;
    double *X, *XT;
    if (which == 0) {
        bg_product<true, false>(Lj, C + (ib + a) * bsz, nullptr, nullptr, 1, MP, r0, c0, kend, lds, acc);
        X = XA + (ib + j) * bsz; XT = XAT + (ib + j) * bsz;
    } else {
        bg_product<true, true>(Lj, C + (ib + j) * bsz, nullptr, nullptr, 1, MP, r0, c0, kend, lds, acc);
        X = XB + (ib + j) * bsz; XT = XBT + (ib + j) * bsz;
    }
    bg_to_lds(acc, lds);
    const int tid = threadIdx.x;
    const int l = tid & 63, h = tid >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int o = h + 4 * i;
        if (r0 + l < MP && c0 + o < MP) X[(r0 + l) + (size_t)MP * (c0 + o)] = lds[o * BG_LD + l];    // (r=l, c=o)
        if (c0 + l < MP && r0 + o < MP) XT[(c0 + l) + (size_t)MP * (r0 + o)] = lds[l * BG_LD + o];   // (c=l, r=o)
    }
}

// Schur update of the surviving blocks a = 2s*t:
//   which 0:  D_a -= XA_{a+s}^T XA_{a+s} + XB_{a-s}^T XB_{a-s}
//   which 1:  C_a  = -XB_{a+s}^T XA_{a+s}                       (new coupling of a+2s with a)
// grid bg_grid(2*nsurv*O, nt); block BG_T.
__global__ __launch_bounds__(BG_T) __attribute__((amdgpu_waves_per_eu(4))) void bcr_upd_kernel(double* __restrict__ D, double* __restrict__ C,
                                                       const double* __restrict__ XA, const double* __restrict__ XAT,
                                                       const double* __restrict__ XB, const double* __restrict__ XBT,
                                                       int N, int O, int MP, int s) {
    __shared__ __attribute__((aligned(32))) double lds[BG_LDS];
    const int nt = (MP + 63) / 64;
    const unsigned ny = 2 * (unsigned)bcr_nsurv(N, 31 - __builtin_clz(s));
    unsigned prod;
    int r0, c0;
    if (!bg_decode(blockIdx.x, ny * (unsigned)O, nt, prod, r0, c0)) return;
    const int yy = (int)(prod % ny), img = (int)(prod / ny);
    const int which = yy & 1, t = yy >> 1;
    const int a = 2 * s * t;
    const size_t bsz = (size_t)MP * MP, ib = (size_t)img * N;
    if (which == 1 && a + 2 * s >= N) return;
    if (which == 0 && a + s >= N && a < s) return;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    double* Out;
    if (which == 0) {
        const bool up = a + s < N, lo = a >= s;
        const double *P0t = XAT + (ib + a + s) * bsz, *P0 = XA + (ib + a + s) * bsz;
        const double *P1t = XBT + (ib + a - s) * bsz, *P1 = XB + (ib + a - s) * bsz;
        if (up && lo) bg_product<true, false>(P0t, P0, P1t, P1, 2, MP, r0, c0, MP, lds, acc);
        else if (up) bg_product<true, false>(P0t, P0, nullptr, nullptr, 1, MP, r0, c0, MP, lds, acc);
        else bg_product<true, false>(P1t, P1, nullptr, nullptr, 1, MP, r0, c0, MP, lds, acc);
        Out = D + (ib + a) * bsz;
    } else {
        bg_product<true, false>(XBT + (ib + a + s) * bsz, XA + (ib + a + s) * bsz, nullptr, nullptr, 1, MP, r0, c0, MP, lds,
                                acc);
        Out = C + (ib + a) * bsz;
    }
    if (BCR_DBG_ON(8)) {
        if (acc.c[0][0][0] == 12345.678) Out[0] = 0.0;
        return;
    }
    const int tid = threadIdx.x;
    const int l = tid & 63, h = tid >> 6;
    // the 16 old values of the thread's entries are requested together, before the accumulators go
    // through LDS (a read-modify-write loop would pay one memory latency per unrolled batch)
    double old[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = h + 4 * i;
        old[i] = (which == 0 && r0 + l < MP && c0 + o < MP) ? Out[(r0 + l) + (size_t)MP * (c0 + o)] : 0.0;
    }
    bg_to_lds(acc, lds);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = h + 4 * i;
        if (r0 + l < MP && c0 + o < MP) Out[(r0 + l) + (size_t)MP * (c0 + o)] = old[i] - lds[o * BG_LD + l];
    }
}

// ---- substitutions -------------------------------------------------------------------------------
// Matrix-vector products with MP x MP blocks are pure memory latency: a block is 128 KB and a
// workgroup that loops over it with a few loads in flight sees ~10 GB/s.  Here the whole block is
// requested at once: 16 waves, lane l owns the rows (2l, 2l+1) (one 16-byte load per column), wave w
// owns the MP/16 columns [w*MP/16, (w+1)*MP/16); the 16 partial sums per row meet in LDS.
constexpr int BS_T = 1024;
constexpr int BS_MP = 128;  // largest MP
constexpr int BS_W = BS_T / 64;

// A thread's slice of a block: rows (2l, 2l+1) x the wave's MP/16 <= 8 columns.  Loading and applying
// are separate so that every block a kernel needs is requested up front -- also the blocks whose
// right-hand side is only known after an earlier product (their latency is then hidden behind it).
struct BcrSlice { double2 m[8]; };

// mode 0 full, 1 lower triangular (Mx(r,k) = 0 for k > r), 2 upper triangular (Mx(r,k) = 0 for k < r)
__device__ __forceinline__ void bcr_mv_load(const double* __restrict__ Mx, int MP, int mode, BcrSlice& t) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = 2 * lane, cw = MP >> 4, k0 = cw * w;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = k0 + i;
        const bool on = (r < MP) && (i < cw) && !(mode == 1 && k > r + 1) && !(mode == 2 && k < r);
        t.m[i] = on ? *reinterpret_cast<const double2*>(Mx + r + (size_t)MP * k) : double2{0.0, 0.0};
    }
}
__device__ __forceinline__ void bcr_mv_apply(const BcrSlice& t, int MP, const double* __restrict__ v, double& s0,
                                             double& s1) {
    const int w = threadIdx.x >> 6, cw = MP >> 4, k0 = cw * w;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < cw) {
            const double vk = v[k0 + i];
            s0 = __builtin_fma(t.m[i].x, vk, s0);
            s1 = __builtin_fma(t.m[i].y, vk, s1);
        }
}
__device__ __forceinline__ void bcr_mv_partial(const double* __restrict__ Mx, int MP, const double* __restrict__ v,
                                               int mode, double& s0, double& s1) {
    BcrSlice t;
    bcr_mv_load(Mx, MP, mode, t);
    bcr_mv_apply(t, MP, v, s0, s1);
}

// partial sums -> red[w][row]; returns the row sum for tid < MP (all threads must call)
__device__ __forceinline__ double bcr_mv_reduce(double* __restrict__ red, int MP, double s0, double s1) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (2 * lane < MP) *reinterpret_cast<double2*>(red + w * BS_MP + 2 * lane) = double2{s0, s1};
    __syncthreads();
    double sum = 0.0;
    if (tid < MP) {
#pragma unroll
        for (int q = 0; q < BS_W; ++q) sum += red[q * BS_MP + tid];
    }
    __syncthreads();
    return sum;
}

// z_j = Linv_j r_j for the blocks eliminated at level 0 (j odd).  grid (nelim(0), O), block BS_T.
__global__ __launch_bounds__(BS_T) void bcr_fz_kernel(const double* __restrict__ Linv, int M, int N, int MP,
                                                      double* __restrict__ vec) {
    __shared__ double v[BS_MP];
    __shared__ __attribute__((aligned(16))) double red[BS_W * BS_MP];
    const int j = 1 + 2 * (int)blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    double* vj = vec + ((size_t)img * N + j) * M;
    if (tid < MP) v[tid] = (tid < M) ? vj[tid] : 0.0;
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    bcr_mv_partial(Linv + ((size_t)img * N + j) * MP * MP, MP, v, 1, s0, s1);
    const double z = bcr_mv_reduce(red, MP, s0, s1);
    if (tid < M) vj[tid] = z;
}

// Forward step of level l for the surviving blocks a = 2s*t:
//   r_a -= XA_{a+s}^T z_{a+s} + XB_{a-s}^T z_{a-s}       (level 0, band4 != null: operator form, see below);
//   if a is eliminated at the next level (or is the last block): z_a = Linv_a r_a,
//   and for the last block also p_0 = Linv_0^T z_0 (+= into accv).
// grid (nsurv(l), O), block BS_T.
template <bool PRE>  // PRE: request every block up front (fewer round trips, 2.5x the registers: small grids)
__global__ __launch_bounds__(BS_T) void bcr_fwd_kernel(const double* __restrict__ Linv, const double* __restrict__ LinvT,
                                                       const double* __restrict__ XAT, const double* __restrict__ XBT,
                                                       int M, int N, int MP, int s, int last,
                                                       double* __restrict__ vec, double* __restrict__ accv,
                                                       const double* __restrict__ band4, int O) {
    __shared__ double za[BS_MP], zb[BS_MP], rn[BS_MP];
    __shared__ __attribute__((aligned(16))) double red[BS_W * BS_MP];
    const int t = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int a = 2 * s * t;
    const size_t bsz = (size_t)MP * MP, ib = (size_t)img * N;
    double* va = vec + (ib + a) * M;
    const bool hasA = (a + s < N), hasB = (a >= s);
    if (tid < MP) {
        za[tid] = (hasA && tid < M) ? vec[(ib + a + s) * M + tid] : 0.0;
        zb[tid] = (hasB && tid < M) ? vec[(ib + a - s) * M + tid] : 0.0;
    }
    const double rold = (tid < M) ? va[tid] : 0.0;
    const bool top = last != 0;                        // only a = 0 survives the last level
    const bool next = top || ((t & 1) != 0);           // eliminated at level l+1
    BcrSlice tA, tB, tL;
    const bool dense = (band4 == nullptr);
    if (PRE) {
        if (dense && hasA) bcr_mv_load(XAT + (ib + a + s) * bsz, MP, 0, tA);
        if (dense && hasB) bcr_mv_load(XBT + (ib + a - s) * bsz, MP, 0, tB);
        if (next) bcr_mv_load(Linv + (ib + a) * bsz, MP, 1, tL);   // needed after the first product: in flight meanwhile
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    double rnew;
    if (band4) {
        // level 0 in operator form (za = y_{a+1}, zb = y_{a-1}):  r_a -= C_a^T y_{a+1} + C_{a-1} y_{a-1}
        double t = 0.0;
        if (tid < M) {
            const size_t tot = (size_t)M * N * O, qa = (ib + a) * M;
            if (hasA) t += band4[3 * tot + qa + tid] * za[tid] + ((tid >= 1) ? band4[2 * tot + qa + tid] * za[tid - 1] : 0.0);
            if (hasB)
                t += band4[3 * tot + qa - M + tid] * zb[tid] +
                     ((tid + 1 < M) ? band4[2 * tot + qa - M + tid + 1] * zb[tid + 1] : 0.0);
        }
        rnew = rold - t;
    } else {
        if (PRE) {
            if (hasA) bcr_mv_apply(tA, MP, za, s0, s1);
            if (hasB) bcr_mv_apply(tB, MP, zb, s0, s1);
        } else {
            if (hasA) bcr_mv_partial(XAT + (ib + a + s) * bsz, MP, za, 0, s0, s1);
            if (hasB) bcr_mv_partial(XBT + (ib + a - s) * bsz, MP, zb, 0, s0, s1);
        }
        rnew = rold - bcr_mv_reduce(red, MP, s0, s1);
    }
    if (!next) {
        if (tid < M) va[tid] = rnew;
        return;
    }
    if (tid < MP) rn[tid] = (tid < M) ? rnew : 0.0;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (PRE && top) bcr_mv_load(LinvT + (ib + a) * bsz, MP, 2, tA);   // last block: its transpose for the way back
    if (!PRE) bcr_mv_load(Linv + (ib + a) * bsz, MP, 1, tL);
    bcr_mv_apply(tL, MP, rn, s0, s1);
    if (!PRE && top) bcr_mv_load(LinvT + (ib + a) * bsz, MP, 2, tA);
    const double z = bcr_mv_reduce(red, MP, s0, s1);
    if (!top) {
        if (tid < M) va[tid] = z;
        return;
    }
    if (tid < MP) za[tid] = (tid < M) ? z : 0.0;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    bcr_mv_apply(tA, MP, za, s0, s1);
    const double p = bcr_mv_reduce(red, MP, s0, s1);
    if (tid < M) {
        va[tid] = p;
        if (accv) accv[(ib + a) * M + tid] += p;
    }
}

// Backward step of level l for the eliminated blocks j = s + 2s*t:
//   p_j = Linv_j^T (z_j - XA_j p_a - XB_j p_b)   (+= into accv).   grid (nelim(l), O), block BS_T.
template <bool PRE>
__global__ __launch_bounds__(BS_T) void bcr_bwd_kernel(const double* __restrict__ LinvT, const double* __restrict__ XA,
                                                       const double* __restrict__ XB, int M, int N, int MP, int s,
                                                       double* __restrict__ vec, double* __restrict__ accv) {
    __shared__ double pa[BS_MP], pb[BS_MP], w[BS_MP];
    __shared__ __attribute__((aligned(16))) double red[BS_W * BS_MP];
    const int t = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int j = s + 2 * s * t, a = j - s, b = j + s;
    const size_t bsz = (size_t)MP * MP, ib = (size_t)img * N;
    const bool hasB = b < N;
    double* vj = vec + (ib + j) * M;
    if (tid < MP) {
        pa[tid] = (tid < M) ? vec[(ib + a) * M + tid] : 0.0;
        pb[tid] = (hasB && tid < M) ? vec[(ib + b) * M + tid] : 0.0;
    }
    const double zj = (tid < M) ? vj[tid] : 0.0;
    BcrSlice tA, tB, tL;
    if (PRE) {
        bcr_mv_load(XA + (ib + j) * bsz, MP, 0, tA);
        if (hasB) bcr_mv_load(XB + (ib + j) * bsz, MP, 0, tB);
        bcr_mv_load(LinvT + (ib + j) * bsz, MP, 2, tL);   // needed after the first product: in flight meanwhile
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    if (PRE) {
        bcr_mv_apply(tA, MP, pa, s0, s1);
        if (hasB) bcr_mv_apply(tB, MP, pb, s0, s1);
    } else {
        bcr_mv_partial(XA + (ib + j) * bsz, MP, pa, 0, s0, s1);
        if (hasB) bcr_mv_partial(XB + (ib + j) * bsz, MP, pb, 0, s0, s1);
    }
    const double acc = bcr_mv_reduce(red, MP, s0, s1);
    if (tid < MP) w[tid] = (tid < M) ? zj - acc : 0.0;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (!PRE) bcr_mv_load(LinvT + (ib + j) * bsz, MP, 2, tL);
    bcr_mv_apply(tL, MP, w, s0, s1);
    const double p = bcr_mv_reduce(red, MP, s0, s1);
    if (tid < M) {
        vj[tid] = p;
        if (accv) accv[(ib + j) * M + tid] += p;
    }
}

// ---- level 0 in operator form -----------------------------------------------------------------------
// The blocks eliminated first (j odd) are still sparse: D_j = T is tridiagonal and the couplings
// C_a (a = j-1) and C_j are bidiagonal.  Treating them as dense blocks makes level 0 half of all the
// work and memory traffic, so level 0 keeps them as what they are:
//   * T = L D L^T (pivots ell_i, 1/delta_i per odd block, [O][N][M]) and tridiagonal solves (Thomas);
//   * (C_a p)(r)   = g_a[r] p(r) + h_a[r+1] p(r+1)      (block a -> block a+1)
//     (C_a^T y)(c) = g_a[c] y(c) + h_a[c] y(c-1)        (block a+1 -> block a)
//     with g_a = the assembled diagonal at offset M, h_a = the one at offset M-1 (band4 planes 3, 2);
//   * the Schur complements that level 1 starts from,
//       D_a  <- T_a - XA^T XA - XB'^T XB',   C'_a = -XB^T XA,     X = D^-1/2 L^-1 C  (as in the dense levels),
//     in closed form.  A column of L^-1 C is one "head" entry followed by a geometric tail:
//       w = (0 .. 0, eta at s-1, omega at s, omega*v(s,s+1), omega*v(s,s+2), ...),  v(s,i) = prod_{m=s}^{i-1} (-ell_m)
//     (XA column c: s = c, eta = h_a[c], omega = g_a[c] - ell_{c-1} eta;  XB column r: s = r+1, eta = g_j[r],
//     omega = h_j[r+1] - ell_r eta), so for s <= s'
//       <w, w'> = [s = s'] eta eta'/delta_{s-1} + [s < s'] eta' omega v(s,s'-1)/delta_{s'-1} + omega omega' v(s,s') Q_{s'}
//     with Q_s = sum_{i>=s} v(s,i)^2/delta_i = 1/delta_s + ell_s^2 Q_{s+1}: products and sums of positive
//     terms only, O(1) per entry with a running product.  (Forming T^-1 by Thomas solves and applying the
//     4-term stencil of C^T T^-1 C instead loses the positive definiteness at kappa = 1e14: tried.)
// No dense block is read or written for an odd block, and no product runs on them.

// Thomas solve T x = v with the pivots (ell, invd) by ONE lane; all arrays in LDS, x may alias v.
__device__ __forceinline__ void bcr0_thomas(const double* __restrict__ ell, const double* __restrict__ invd,
                                            const double* v, double* x, int M) {
    double w = v[0];
    x[0] = w;
    for (int i = 1; i < M; ++i) {
        w = __builtin_fma(-ell[i - 1], w, v[i]);
        x[i] = w;
    }
    double xn = x[M - 1] * invd[M - 1];
    x[M - 1] = xn;
    for (int i = M - 2; i >= 0; --i) {
        xn = __builtin_fma(-ell[i], xn, x[i] * invd[i]);
        x[i] = xn;
    }
}

// 1/x from v_rcp_f64 + two Newton steps (the pivots need not be correctly rounded; an IEEE divide
// would put ~40 dependent instructions into every step of the sequential pivot recurrence)
__device__ __forceinline__ double bcr0_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// pivots of the tridiagonal (d, e) by ONE lane (LDS arrays): delta_0 = d_0, ell_i = e_i/delta_i,
// delta_{i+1} = d_{i+1} - ell_i e_i; then Q (M entries, Q_M = 0 implied).  Returns false on a
// non-positive pivot.
__device__ __forceinline__ bool bcr0_pivots(const double* __restrict__ d, const double* __restrict__ e,
                                            double* __restrict__ ell, double* __restrict__ invd,
                                            double* __restrict__ Q, int M) {
    double del = d[0];
    bool ok = del > 0.0;
    for (int i = 0; i < M; ++i) {
        const double inv = bcr0_rcp(del);
        invd[i] = inv;
        if (i + 1 < M) {
            const double l = e[i] * inv;
            ell[i] = l;
            del = __builtin_fma(-l, e[i], d[i + 1]);
            ok = ok && (del > 0.0);
        } else {
            ell[i] = 0.0;
        }
    }
    double q = 0.0;
    for (int i = M - 1; i >= 0; --i) {
        q = __builtin_fma(ell[i] * ell[i], q, invd[i]);
        Q[i] = q;
    }
    return ok;
}

// Pivots (ell, 1/delta) and Q of every odd block, once per factorisation.  grid (nelim(0), O), block 64.
// Arrays [O][N][M], indexed by block.
__global__ __launch_bounds__(64) void bcr0_pivot_kernel(const double* __restrict__ band4, int M, int N, int O,
                                                        double* __restrict__ ell, double* __restrict__ invd,
                                                        double* __restrict__ Qg, int* __restrict__ fail) {
    __shared__ double d[BS_MP], e[BS_MP], l[BS_MP], iv[BS_MP], q[BS_MP];
    const int j = 1 + 2 * (int)blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const size_t tot = (size_t)M * N * O;
    const size_t o = ((size_t)img * N + j) * M;
    for (int i = tid; i < M; i += 64) {
        d[i] = band4[o + i];
        e[i] = (i + 1 < M) ? band4[tot + o + i] : 0.0;
    }
    __syncthreads();
    if (tid == 0 && !bcr0_pivots(d, e, l, iv, q, M) && fail[img] == 0) fail[img] = j + 1;
    __syncthreads();
    for (int i = tid; i < M; i += 64) { ell[o + i] = l[i]; invd[o + i] = iv[i]; Qg[o + i] = q[i]; }
}

// v(s0, s1) = prod_{m=s0}^{s1-1} (-ell_m), four independent partial products
__device__ __forceinline__ double bcr0_vprod(const double* __restrict__ ell, int s0, int s1) {
    double p0 = 1.0, p1 = 1.0, p2 = 1.0, p3 = 1.0;
    int m = s0;
    for (; m + 4 <= s1; m += 4) { p0 *= -ell[m]; p1 *= -ell[m + 1]; p2 *= -ell[m + 2]; p3 *= -ell[m + 3]; }
    for (; m < s1; ++m) p0 *= -ell[m];
    return (p0 * p1) * (p2 * p3);
}

constexpr int B0_T = 1024;
constexpr int B0_SEG = 16;  // columns of one row handled by one thread

// Gram matrix of one family of columns (tail start s = x + off) into S (both triangles).  Thread
// (x = tid % 128, g = tid / 128) fills the entries (x, y), y in [16g, 16g+16), y >= x, and their mirrors.
__device__ __forceinline__ void bcr0_gram(double* __restrict__ S, int ld, int M, int off, const double* __restrict__ eta,
                                          const double* __restrict__ om, const double* __restrict__ ell,
                                          const double* __restrict__ invd, const double* __restrict__ Q) {
    const int x = threadIdx.x & 127, g = threadIdx.x >> 7;
    if (x >= M) return;
    const int s = x + off;
    const double ex = eta[x], ox = om[x];
    int y0 = B0_SEG * g, y1 = y0 + B0_SEG;
    if (y1 > M) y1 = M;
    if (y0 <= x && x < y1) {
        S[x + ld * x] = ((s >= 1) ? invd[s - 1] * ex * ex : 0.0) + ((s < M) ? ox * ox * Q[s] : 0.0);
        y0 = x + 1;
    }
    if (y0 <= x || y0 >= y1) return;
    double v = bcr0_vprod(ell, s, y0 + off - 1);  // v(s, s'-1) for s' = y0 + off
    for (int y = y0; y < y1; ++y) {
        const int sp = y + off;          // s' - 1 >= s, s' - 1 <= M - 1
        const double head = invd[sp - 1] * eta[y] * (ox * v);
        v *= -ell[sp - 1];               // v(s, s')
        const double tail = (sp < M) ? (ox * om[y]) * v * Q[sp] : 0.0;
        const double r = head + tail;
        S[x + ld * y] = r;
        S[y + ld * x] = r;
    }
}

// Cross Gram matrix S(r, c) = <XB column r, XA column c> (XB: s = r+1; XA: s = c).  Thread
// (r = tid % 128, g = tid / 128) fills the columns c in [16g, 16g+16).
__device__ __forceinline__ void bcr0_cross(double* __restrict__ S, int ld, int M, const double* __restrict__ etaB,
                                           const double* __restrict__ omB, const double* __restrict__ etaA,
                                           const double* __restrict__ omA, const double* __restrict__ ell,
                                           const double* __restrict__ invd, const double* __restrict__ Q) {
    const int r = threadIdx.x & 127, g = threadIdx.x >> 7;
    if (r >= M) return;
    const int sb = r + 1;
    const double eb = etaB[r], ob = omB[r];
    int c0 = B0_SEG * g, c1 = c0 + B0_SEG;
    if (c1 > M) c1 = M;
    if (c0 >= c1) return;
    // columns c >= r+1: the XB column starts first (s = sb <= s' = c)
    {
        int c = (c0 > sb) ? c0 : sb;
        if (c < c1) {
            double v = 1.0;  // v(sb, c-1)
            if (c == sb) {
                S[r + ld * sb] = invd[sb - 1] * eb * etaA[sb] + ob * omA[sb] * Q[sb];
                ++c;
            } else {
                v = bcr0_vprod(ell, sb, c - 1);
            }
            for (; c < c1; ++c) {
                const double head = invd[c - 1] * etaA[c] * (ob * v);
                v *= -ell[c - 1];
                S[r + ld * c] = head + (ob * omA[c]) * v * Q[c];
            }
        }
    }
    // columns c <= r: the XA column starts first (s = c < s' = sb); head of the XB column at index r
    {
        int c = (c1 - 1 < r) ? c1 - 1 : r;
        if (c >= c0) {
            double v = bcr0_vprod(ell, c, r);      // v(c, r)
            const double qt = (sb < M) ? -ell[r] * Q[sb] : 0.0;
            for (; c >= c0; --c) {
                S[r + ld * c] = invd[r] * eb * (omA[c] * v) + (omA[c] * ob) * (v * qt);
                if (c >= 1) v *= -ell[c - 1];
            }
        }
    }
}

// Level-0 Schur complements.  grid (nsurv(0), O), block B0_T, dynamic LDS bcr0_schur_lds(MP).
// Writes the dense D_a (Li slot) and C'_a (C slot) of every even block a.
__global__ __launch_bounds__(B0_T) void bcr0_schur_kernel(const double* __restrict__ band4, int M, int N, int O, int MP,
                                                          double* __restrict__ D, double* __restrict__ C,
                                                          const double* __restrict__ ell, const double* __restrict__ invd,
                                                          const double* __restrict__ Qg) {
    extern __shared__ double S[];
    const int ld = MP + 1;
    double* co = S + (size_t)ld * MP;
    const int MQ = MP + 1;
    // per side (0 = block a-1, 1 = block a+1): ell, invd, Q
    double *ll[2] = {co, co + MQ}, *iv[2] = {co + 2 * MQ, co + 3 * MQ}, *QQ[2] = {co + 4 * MQ, co + 5 * MQ};
    // couplings: g/h of C_{a-1}, C_a, C_{a+1};  omegas of the three column families
    double *gl = co + 6 * MQ, *hl = co + 7 * MQ, *gu = co + 8 * MQ, *hu = co + 9 * MQ, *g2 = co + 10 * MQ,
           *h2 = co + 11 * MQ, *oBl = co + 12 * MQ, *oA = co + 13 * MQ, *oB2 = co + 14 * MQ;
    const int a = 2 * (int)blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const size_t npx = (size_t)M * N, tot = npx * O;
    const size_t qa = (size_t)img * npx + (size_t)a * M;   // first pixel of block a (== its offset in the pivot arrays)
    const bool has[2] = {a >= 1, a + 1 < N};
    const bool hasC2 = a + 2 < N;
    if (tid <= MP) {
        const int i = tid;
        const bool in = i < M;
        ll[0][i] = (has[0] && in) ? ell[qa - M + i] : 0.0;
        iv[0][i] = (has[0] && in) ? invd[qa - M + i] : 1.0;
        QQ[0][i] = (has[0] && in) ? Qg[qa - M + i] : 0.0;
        ll[1][i] = (has[1] && in) ? ell[qa + M + i] : 0.0;
        iv[1][i] = (has[1] && in) ? invd[qa + M + i] : 1.0;
        QQ[1][i] = (has[1] && in) ? Qg[qa + M + i] : 0.0;
        gl[i] = (has[0] && in) ? band4[3 * tot + qa - M + i] : 0.0;             // C_{a-1}(r, r)
        hl[i] = (has[0] && in && i >= 1) ? band4[2 * tot + qa - M + i] : 0.0;   // C_{a-1}(c-1, c)
        gu[i] = (has[1] && in) ? band4[3 * tot + qa + i] : 0.0;                 // C_a
        hu[i] = (has[1] && in && i >= 1) ? band4[2 * tot + qa + i] : 0.0;
        g2[i] = (hasC2 && in) ? band4[3 * tot + qa + M + i] : 0.0;              // C_{a+1}
        h2[i] = (hasC2 && in && i >= 1) ? band4[2 * tot + qa + M + i] : 0.0;
    }
    __syncthreads();
    if (tid < M) {
        const int i = tid;
        oBl[i] = (i + 1 < M) ? hl[i + 1] - ll[0][i] * gl[i] : 0.0;
        oA[i] = gu[i] - ((i >= 1) ? ll[1][i - 1] * hu[i] : 0.0);
        oB2[i] = (i + 1 < M) ? h2[i + 1] - ll[1][i] * g2[i] : 0.0;
    }
    __syncthreads();
    const size_t bo = ((size_t)img * N + a) * MP * MP;
    auto tridiag = [&](int r, int c) -> double {
        if (r < M && c < M) {
            if (r == c) return band4[qa + r];
            if (r == c + 1) return band4[tot + qa + c];
            if (c == r + 1) return band4[tot + qa + r];
            return 0.0;
        }
        return (r == c) ? 1.0 : 0.0;
    };
    // the thread's entries e = tid + B0_T*i of D_a: tridiagonal part minus the two Gram matrices
    constexpr int NE = BS_MP * BS_MP / B0_T;
    double acc[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + B0_T * i;
        acc[i] = (e < MP * MP) ? tridiag(e % MP, e / MP) : 0.0;
    }
    if (has[0]) {   // XB'^T XB' of block a-1
        bcr0_gram(S, ld, M, 1, gl, oBl, ll[0], iv[0], QQ[0]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + B0_T * i;
            const int r = e % MP, c = e / MP;
            if (e < MP * MP && r < M && c < M) acc[i] -= S[r + ld * c];
        }
        __syncthreads();
    }
    if (has[1]) {   // XA^T XA of block a+1
        bcr0_gram(S, ld, M, 0, hu, oA, ll[1], iv[1], QQ[1]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + B0_T * i;
            const int r = e % MP, c = e / MP;
            if (e < MP * MP && r < M && c < M) acc[i] -= S[r + ld * c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + B0_T * i;
        if (e < MP * MP) D[bo + e] = acc[i];
    }
    if (hasC2) {    // the new coupling -XB^T XA
        bcr0_cross(S, ld, M, g2, oB2, hu, oA, ll[1], iv[1], QQ[1]);
        __syncthreads();
        for (int e = tid; e < MP * MP; e += B0_T) {
            const int r = e % MP, c = e / MP;
            C[bo + e] = (r < M && c < M) ? -S[r + ld * c] : 0.0;
        }
    }
}

// y_j = T_j^-1 r_j for the odd blocks.  grid (nelim(0), O), block 64.
__global__ __launch_bounds__(64) void bcr0_y_kernel(const double* __restrict__ ell, const double* __restrict__ invd,
                                                    int M, int N, double* __restrict__ vec) {
    __shared__ double l[BS_MP], iv[BS_MP], v[BS_MP];
    const int j = 1 + 2 * (int)blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const size_t o = ((size_t)img * N + j) * M;
    for (int i = tid; i < M; i += 64) { l[i] = ell[o + i]; iv[i] = invd[o + i]; v[i] = vec[o + i]; }
    __syncthreads();
    if (tid == 0) bcr0_thomas(l, iv, v, v, M);
    __syncthreads();
    for (int i = tid; i < M; i += 64) vec[o + i] = v[i];
}

// p_j = y_j - T_j^-1 (C_a p_a + C_j^T p_b) for the odd blocks (+= into accv).  grid (nelim(0), O), block 64.
__global__ __launch_bounds__(64) void bcr0_bwd_kernel(const double* __restrict__ band4,
                                                      const double* __restrict__ ell, const double* __restrict__ invd,
                                                      int M, int N, int O, double* __restrict__ vec,
                                                      double* __restrict__ accv) {
    __shared__ double l[BS_MP], iv[BS_MP], v[BS_MP], pa[BS_MP + 1], pb[BS_MP + 1];
    const int j = 1 + 2 * (int)blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const size_t npx = (size_t)M * N, tot = npx * O;
    const size_t o = ((size_t)img * N + j) * M;           // == first pixel of block j
    const bool hasB = j + 1 < N;
    for (int i = tid; i < M; i += 64) {
        l[i] = ell[o + i]; iv[i] = invd[o + i];
        pa[i] = vec[o - M + i];
        pb[i + 1] = hasB ? vec[o + M + i] : 0.0;           // pb[c] = p_b(c-1)
    }
    if (tid == 0) { pa[M] = 0.0; pb[0] = 0.0; }
    __syncthreads();
    for (int i = tid; i < M; i += 64) {
        // (C_a p_a)(i) = g_a[i] p_a(i) + h_a[i+1] p_a(i+1);  (C_j^T p_b)(i) = g_j[i] p_b(i) + h_j[i] p_b(i-1)
        const double ga = band4[3 * tot + o - M + i], ha = (i + 1 < M) ? band4[2 * tot + o - M + i + 1] : 0.0;
        double t = ga * pa[i] + ha * pa[i + 1];
        if (hasB) {
            const double gj = band4[3 * tot + o + i], hj = (i >= 1) ? band4[2 * tot + o + i] : 0.0;
            t += gj * pb[i + 1] + hj * pb[i];
        }
        v[i] = t;
    }
    __syncthreads();
    if (tid == 0) bcr0_thomas(l, iv, v, v, M);
    __syncthreads();
    for (int i = tid; i < M; i += 64) {
        const double p = vec[o + i] - v[i];
        vec[o + i] = p;
        if (accv) accv[o + i] += p;
    }
}

// ---- host-side launch sequences -------------------------------------------------------------------
struct BcrArrays {
    double *Li, *LiT, *C, *XA, *XAT, *XB, *XBT;  // [O][N][MP*MP] each
    double *ell, *invd, *Q;                        // pivots of the odd blocks, [O][N][M] each
    static size_t doubles(int M, int N, int O, int MP) { return 7 * (size_t)O * N * MP * MP + 3 * (size_t)O * N * M; }
    static BcrArrays carve(double* base, int M, int N, int O, int MP) {
        const size_t a = (size_t)O * N * MP * MP;
        return BcrArrays{base, base + a, base + 2 * a, base + 3 * a, base + 4 * a, base + 5 * a, base + 6 * a,
                         base + 7 * a, base + 7 * a + (size_t)O * N * M, base + 7 * a + 2 * (size_t)O * N * M};
    }
};
inline size_t bcr_potrf_lds(int MP) { return ((size_t)(MP + 1) * MP + MP) * sizeof(double); }   // matrix + 1/diag(L)
inline size_t bcr0_schur_lds(int MP) { return ((size_t)(MP + 1) * MP + 15 * (size_t)(MP + 1)) * sizeof(double); }

// Dense levels l >= l0 of the factorisation of the block tridiagonal matrices held in
// (B.Li = diagonal blocks, B.C = couplings at stride 2^l0), then the last block.
inline void bcr_factor_launch(hipStream_t st, const BcrArrays& B, int N, int O, int MP, int* d_fail, int l0 = 0) {
    const int BL = bcr_levels(N);
    const unsigned nt = (unsigned)((MP + 63) / 64);
    for (int l = l0; l < BL; ++l) {
        const int s = 1 << l, ne = bcr_nelim(N, l), ns = bcr_nsurv(N, l);
        hipLaunchKernelGGL(bcr_potrf_kernel, dim3(ne, O), dim3(BCR_PT), bcr_potrf_lds(MP), st, B.Li, B.LiT, N, MP, s, d_fail);
        hipLaunchKernelGGL(bcr_x_kernel, dim3(bg_grid(2u * ne * O, nt)), dim3(BG_T), 0, st, B.Li, B.C, B.XA, B.XAT, B.XB, B.XBT,
                           N, O, MP, s);
        hipLaunchKernelGGL(bcr_upd_kernel, dim3(bg_grid(2u * ns * O, nt)), dim3(BG_T), 0, st, B.Li, B.C, B.XA, B.XAT, B.XB,
                           B.XBT, N, O, MP, s);
    }
    hipLaunchKernelGGL(bcr_potrf_kernel, dim3(1, O), dim3(BCR_PT), bcr_potrf_lds(MP), st, B.Li, B.LiT, N, MP, 0, d_fail);
}

// Assembled diagonals -> factorisation: level 0 in operator form, the rest dense.
inline void bcr_factor_band4_launch(hipStream_t st, const BcrArrays& B, const double* band4, int M, int N, int O, int MP,
                                    int* d_fail) {
    hipLaunchKernelGGL(bcr0_pivot_kernel, dim3(bcr_nelim(N, 0), O), dim3(64), 0, st, band4, M, N, O, B.ell, B.invd, B.Q, d_fail);
    hipLaunchKernelGGL(bcr0_schur_kernel, dim3(bcr_nsurv(N, 0), O), dim3(B0_T), bcr0_schur_lds(MP), st, band4, M, N, O, MP,
                       B.Li, B.C, B.ell, B.invd, B.Q);
    bcr_factor_launch(st, B, N, O, MP, d_fail, 1);
}

// vec <- A^-1 vec (in place, [O][N*M]); accv += the solution when non-null.  band4 != null: the
// factorisation came from bcr_factor_band4_launch (level 0 in operator form).
inline void bcr_solve_launch(hipStream_t st, const BcrArrays& B, int M, int N, int O, int MP, double* vec, double* accv,
                             const double* band4 = nullptr) {
    const int BL = bcr_levels(N);
    if (band4)
        hipLaunchKernelGGL(bcr0_y_kernel, dim3(bcr_nelim(N, 0), O), dim3(64), 0, st, B.ell, B.invd, M, N, vec);
    else
        hipLaunchKernelGGL(bcr_fz_kernel, dim3(bcr_nelim(N, 0), O), dim3(BS_T), 0, st, B.Li, M, N, MP, vec);
    // grids of at most one workgroup per CU pair: latency bound -> prefetching instance
    auto small = [&](int blocks) { return (size_t)blocks * O <= 192; };
    for (int l = 0; l < BL; ++l) {
        const int ns = bcr_nsurv(N, l);
        if (small(ns))
            hipLaunchKernelGGL(bcr_fwd_kernel<true>, dim3(ns, O), dim3(BS_T), 0, st, B.Li, B.LiT, B.XAT, B.XBT, M, N, MP, 1 << l,
                               (l == BL - 1) ? 1 : 0, vec, accv, (l == 0) ? band4 : (const double*)nullptr, O);
        else
            hipLaunchKernelGGL(bcr_fwd_kernel<false>, dim3(ns, O), dim3(BS_T), 0, st, B.Li, B.LiT, B.XAT, B.XBT, M, N, MP, 1 << l,
                               (l == BL - 1) ? 1 : 0, vec, accv, (l == 0) ? band4 : (const double*)nullptr, O);
    }
    for (int l = BL - 1; l >= 1; --l) {
        const int ne = bcr_nelim(N, l);
        if (small(ne))
            hipLaunchKernelGGL(bcr_bwd_kernel<true>, dim3(ne, O), dim3(BS_T), 0, st, B.LiT, B.XA, B.XB, M, N, MP, 1 << l, vec, accv);
        else
            hipLaunchKernelGGL(bcr_bwd_kernel<false>, dim3(ne, O), dim3(BS_T), 0, st, B.LiT, B.XA, B.XB, M, N, MP, 1 << l, vec, accv);
    }
    if (band4)
        hipLaunchKernelGGL(bcr0_bwd_kernel, dim3(bcr_nelim(N, 0), O), dim3(64), 0, st, band4, B.ell, B.invd, M, N, O, vec,
                           accv);
    else
        hipLaunchKernelGGL(bcr_bwd_kernel<false>, dim3(bcr_nelim(N, 0), O), dim3(BS_T), 0, st, B.LiT, B.XA, B.XB, M, N, MP, 1,
                           vec, accv);
}

}  // namespace bpltv
