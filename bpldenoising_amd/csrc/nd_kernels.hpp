// nd_kernels.hpp -- numeric kernels of the nested-dissection (multifrontal) Cholesky, gfx950.
//
// Replaces the sparse direct solve behind Julia's `\` at /root/reference/src/TVLearningFunctionVec.jl:131,248 (and
// /root/reference/src/SumRegsLearningFunction.jl:324,394) for images too wide for block cyclic reduction: the tree
// of csrc/nd_symbolic.hpp turns the stencil matrix into dense fronts, and every level of the tree is eliminated by a
// handful of launches over (front, image).  Front F of a node with p pivots and b boundary pixels (f = p + b):
//     F = [ F11 F21^T ]      F11 = L11 L11^T,  L21 = F21 L11^-T,  U = F22 - L21 L21^T  (update matrix, to the parent)
//         [ F21 F22   ]
// Storage (per image): the factor columns of the node, f x p with leading dimension f -- rows [0, p) end up holding
// W = L11^-1 (lower triangular; the substitutions only ever apply L11^-1, never L11), rows [p, f) hold L21 -- and
// the update matrix U, b x b, in the workspace of the node's level parity (the parent reads it one level later).
// Only lower triangles are assembled and read.
//
// Two regimes, chosen per level:
//   small  (front padded to 16 fits 128 x 128): one workgroup per (front, image) assembles the front in LDS (matrix
//          entries + the children's update matrices through cmap), factors its pivot block columns (wave 0 in
//          registers, bcr_panel_factor), applies them to the rest on the f64 MFMA and writes W, L21, U.
//   large  the front stays in HBM/L2: a gather kernel assembles it (each entry written once), then per 128-column pivot panel
//          nd_potrf_kernel (Cholesky + inverse of the diagonal block in LDS, bcr_potrf_lds_body) ->
//          nd_trsm_kernel (rows below: x L11^-T, MFMA tiles, in place) -> nd_syrk_kernel (remaining pivot columns),
//          and one nd_schur_kernel for U with the whole depth p (64 x 64 MFMA tiles).
// Substitutions in gather form (no atomics, fixed summation order, bitwise reproducible): forward, level by level
// from the leaves, a front subtracts its children's update vectors, applies W and hands L21 y to its parent;
// backward from the root, x_p = W^T (y_p - L21^T x_b).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "adjoint_hbm_kernels.hpp"

namespace bpltv {

struct NdNodeDev {
    int p, b;
    int piv_off, cmap_off;
    int orig_off, orig_cnt;
    int child0, child1;
    long long fac_off, u_off, uv_off;
    int inv_off, pad_;         // large-regime fronts: inv[inv_off + ci * f + l] = index of front entry l in child ci's boundary, or -1
};

struct NdArgs {
    const NdNodeDev* nodes;
    const int* pix;
    const int* cmap;
    const int4* orig;          // (r, c, plane, pixel)
    const int* inv;            // parent -> child maps of the large-regime fronts (NdNodeDev::inv_off)
    const double* planes;      // assembled diagonals: planes[plane * tot + img * n + pixel]
    const double* planes2;     // the diagonals of the other triangle (LU variant; = planes for a symmetric matrix)
    size_t tot;
    int n;                     // pixels per image
    double* fac;               // [nimg][fac_stride]
    double* fac2;              // LU variant: the factor columns of the other triangle (second operand of the rank updates;
                               // = fac for Cholesky)
    long long fac_stride;
    double* ws_mine;           // update matrices of this level   [nimg][ws_mine_stride]
    const double* ws_child;    // update matrices of the children [nimg][ws_child_stride]
    double* ws_mine2;          // LU variant, small fronts: the transposed upper triangles (nd_front_small_lu_kernel)
    const double* ws_child2;
    long long ws_mine_stride, ws_child_stride;
    int node0;                 // first node of the batch: blockIdx.y = node - node0
    int* fail;                 // [nimg]: node + 1 of the first non-positive pivot
};

__host__ __device__ inline int nd_up16(int x) { return (x + 15) & ~15; }

#ifdef ND_PROBE_ON   // tools/nd_unit.hip built with -DND_PROBE_ON only: time stamps (100 MHz) of one mid-grid workgroup's phases
__device__ long long nd_probe_buf[16];
__device__ int nd_probe_node0 = -1;   // stamp the small-front launch whose first node is this one
#define ND_PROBE(i) do { if (threadIdx.x == 0 && A.node0 == nd_probe_node0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y - 1) nd_probe_buf[i] = (long long)wall_clock64(); } while (0)
#else
#define ND_PROBE(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------------
// small regime
// ------------------------------------------------------------------------------------------------------------------
// Partial Cholesky in LDS: the first Pp block columns (16 columns each) of the MP x MP matrix S (leading dimension
// ld = MP + 1, lower triangle) are factored, the inverse of the Pp x Pp pivot tile block is formed, and the
// remaining tiles receive -L21 L21^T.  Layout on exit as bcr_potrf_lds_body's: strictly lower tiles hold L, the
// pivot block's diagonal and upper tiles hold W = L11^-1 (tile (p, q), q <= p, at tile position (q, p)).
// dinv: MP doubles of scratch.  Returns true (wave 0) on a non-positive pivot.  NT threads.
// BIG = false: MP <= 64 only (one row per lane in the panel factor: half the registers, twice the workgroups per CU).
// pact: real pivots (<= 16 Pp); the rest of the last block column is identity padding and is skipped by the panel factor.
template <int NT, bool BIG>
__device__ __forceinline__ bool nd_partial_potrf(double* __restrict__ S, int MP, int Pp, double* __restrict__ dinv, int pact) {
    static_assert(NT >= 128, "a wave for the tile inverses beside the trailing update");
    const int ld = MP + 1, P = MP >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    constexpr int NW = NT / 64;
    bool bad = false;
    for (int pc = 0; pc < Pp; ++pc) {
        if (pc > 0) {   // left-looking: block column pc receives the block columns before it
            for (int i = pc + wave; i < P; i += NW) {
                bcr_d4 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = S[(16 * i + lk + 4 * g) + ld * (16 * pc + lr)];
                for (int q = 0; q < pc; ++q) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double a = S[(16 * i + lr) + ld * (16 * q + 4 * kk + lk)];
                        const double b = S[(16 * pc + lr) + ld * (16 * q + 4 * kk + lk)];
                        acc = bcr_mfma(-a, b, acc);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * pc + lr)] = acc[g];
            }
            __syncthreads();
        }
        if (wave == 0) {
            const int nact = min(16, pact - 16 * pc);
            if (BIG && MP - 16 * pc > 64) bad |= bcr_panel_factor<true>(S, ld, MP, pc, lane, dinv, nact);
            else bad |= bcr_panel_factor<false>(S, ld, MP, pc, lane, dinv, nact);
        }
        __syncthreads();
    }
    // trailing tiles (i, j), Pp <= j <= i: -= sum_q L(i, q) L(j, q)^T (they read strictly-lower tiles below the pivot
    // block only), on the waves 1..; meanwhile wave 0 inverts the diagonal tiles of the pivot block (disjoint tiles)
    if (wave == 0) {
        for (int t = 0; t < Pp; ++t) bcr_tile_inverse(S + 16 * t + ld * (16 * t), ld, lane, dinv + 16 * t);
    } else {
        const int m = P - Pp;
        for (int t = wave - 1; t < m * (m + 1) / 2; t += NW - 1) {
            int a = 0, c = t;
            while (c > a) { c -= a + 1; ++a; }
            const int i = Pp + a, j = Pp + c;
            bcr_d4 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = S[(16 * i + lk + 4 * g) + ld * (16 * j + lr)];
            for (int q = 0; q < Pp; ++q) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double x = S[(16 * i + lr) + ld * (16 * q + 4 * kk + lk)];
                    const double y = S[(16 * j + lr) + ld * (16 * q + 4 * kk + lk)];
                    acc = bcr_mfma(-x, y, acc);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * j + lr)] = acc[g];
        }
    }
    // W = L11^-1 below the diagonal tiles, tile row by tile row (row p needs the rows above it)
    __syncthreads();
    for (int p = 1; p < Pp; ++p) {
        for (int q = wave; q < p; q += NW) bcr_winv_tile(S, ld, p, q, lr, lk);
        __syncthreads();
    }
    return bad;
}

constexpr int NDS_T = 256;
constexpr int NDS_CU = 4;     // children's update-matrix columns in flight per wave (nd_front_small_kernel; 8 measured: no gain)
inline size_t nd_small_lds(int MP) { return sizeof(double) * ((size_t)(MP + 1) * MP + MP); }

// One workgroup per (front, image).  grid (nodes of the batch, nimg), block NDS_T, dynamic LDS nd_small_lds(MPmax).
// Loops over matrix entries run column by column -- a wave per column, lanes along the rows -- so that global
// accesses are unit-stride and no index needs a division; the children's maps are staged in LDS once.
// NT = 128 for the smallest fronts (MP <= 48): two waves per front, twice the fronts per CU.
template <bool BIG, int NT>
__global__ __launch_bounds__(NT, BIG ? 2 : 4 * NT / 256) void nd_front_small_kernel(NdArgs A) {
    extern __shared__ double S[];
    __shared__ int cmL[2][128];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = NT / 64;
    ND_PROBE(0);
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b, p16 = nd_up16(p), sh = p16 - p;
    const int MP = nd_up16(p16 + b), ld = MP + 1, Pp = p16 >> 4;
    double* dinv = S + (size_t)ld * MP;
    // children: descriptors and maps are requested now, they land while the front is cleared
    NdNodeDev ch[2];
    int bc[2] = {0, 0};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (cn >= 0) { ch[ci] = A.nodes[cn]; bc[ci] = ch[ci].b; }
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
        for (int k = tid; k < bc[ci]; k += NT) {
            const int R = A.cmap[ch[ci].cmap_off + k];
            cmL[ci][k] = R < p ? R : R + sh;
        }
    for (int e = tid; e < ld * MP + MP; e += NT) S[e] = 0.0;
    __syncthreads();
    ND_PROBE(1);
    for (int k = p + tid; k < p16; k += NT) S[k + ld * k] = 1.0;   // identity padding of the pivot block
    // matrix entries of this front (unique targets)
    {
        const double* pl = A.planes + (size_t)img * A.n;
        for (int e = tid; e < v.orig_cnt; e += NT) {
            const int4 o = A.orig[v.orig_off + e];
            const int r = o.x < p ? o.x : o.x + sh, c = o.y;   // c is a pivot
            S[r + ld * c] += pl[(size_t)(o.z & 15) * A.tot + o.w];
        }
    }
    __syncthreads();
    ND_PROBE(2);
    // the children's update matrices, one child after the other (a child's entries hit distinct targets)
    for (int ci = 0; ci < 2; ++ci) {
        if (bc[ci] == 0) continue;
        const double* Uc = A.ws_child + (size_t)img * A.ws_child_stride + ch[ci].u_off;
        const int n = bc[ci];
        const int* cm = cmL[ci];
        for (int j0 = wave; j0 < n; j0 += NDS_CU * NW) {   // NDS_CU columns per wave and pass: their loads are in flight together
            double x[NDS_CU][2];
#pragma unroll
            for (int u = 0; u < NDS_CU; ++u) {
                const int j = j0 + u * NW;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = j + lane + 64 * h;
                    x[u][h] = (j < n && i < n) ? Uc[i + (size_t)n * j] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < NDS_CU; ++u) {
                const int j = j0 + u * NW;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = j + lane + 64 * h;
                    if (j < n && i < n) S[cm[i] + ld * cm[j]] += x[u][h];
                }
            }
        }
        __syncthreads();
    }
    ND_PROBE(3);
    const bool bad = nd_partial_potrf<NT, BIG>(S, MP, Pp, dinv, p);
    if (bad && lane == 0 && A.fail[img] == 0) A.fail[img] = node + 1;
    __syncthreads();
    ND_PROBE(4);
    // factor columns: rows [0, p) = W (zeros above the diagonal), rows [p, f) = L21
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    for (int c = wave; c < p; c += NW)
        for (int r = lane; r < f; r += 64) {
            double x;
            if (r < p) x = (r >= c) ? S[(16 * (c >> 4) + (r & 15)) + ld * (16 * (r >> 4) + (c & 15))] : 0.0;
            else x = S[(r + sh) + ld * c];
            fc[r + (size_t)f * c] = x;
        }
    double* U = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
    for (int j = wave; j < b; j += NW)
        for (int i = j + lane; i < b; i += 64) U[i + (size_t)b * j] = S[(p16 + i) + ld * (p16 + j)];
    ND_PROBE(5);
}

// ------------------------------------------------------------------------------------------------------------------
// small regime, SKINNY fronts: p <= 16 pivots, up to 128 rows -- the front is never assembled
// ------------------------------------------------------------------------------------------------------------------
// The two levels below the large regime (1024^2: 49 152 fronts of 7 .. 16 pivots and 45 .. 109 rows per eight images).
// nd_front_small_kernel assembles such a front as a 112 x 113 image in LDS -- 101 KB: ONE workgroup per CU -- although almost
// all of it is the update matrix, which only passes through: U = (children's entries that land in F22) - L21 L21^T.  Here
// only the pivot block column lives in LDS (MP x 16, 16.5 KB: eight workgroups per CU): matrix entries and the children's
// entries in gather form through the parent -> child maps (`inv`, as nd_gather_kernel; the host builds them for these
// levels too), factored by wave 0 (bcr_panel_factor, the pivot chain of nd_partial_potrf) while the other waves already
// request the children's entries of their first update-matrix tiles; then every 16 x 16 tile of U is produced once:
// the gathered sum (child 0 + child 1) as the MFMA accumulator, minus L21_i L21_j^T (one block column: four
// v_mfma_f64_16x16x4), straight to HBM.  Same operations in the same order as nd_front_small_kernel: the same bits.
// grid (fronts of the level, nimg), block 256; dynamic LDS nd_skinny_lds(MPmax).
inline size_t nd_skinny_lds(int MP) { return sizeof(double) * ((size_t)(MP + 1) * 16 + 16) + sizeof(int) * 2 * 128; }

__global__ __launch_bounds__(256, 3) void nd_front_skinny_kernel(NdArgs A) {
    extern __shared__ double S[];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b, sh = 16 - p;              // boundary row i of the front sits at panel row 16 + i
    const int MP = 16 + nd_up16(b), ld = MP + 1, P = MP >> 4;
    double* dinv = S + (size_t)ld * 16;
    int* iv = reinterpret_cast<int*>(dinv + 16);                     // iv[ci * 128 + front row]: index in child ci's boundary, or -1
    const bool has = v.inv_off >= 0;
    const double* Uc[2] = {A.ws_child, A.ws_child};
    int bc[2] = {0, 0};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (has && cn >= 0) {
            const NdNodeDev ch = A.nodes[cn];
            bc[ci] = ch.b;
            Uc[ci] = A.ws_child + (size_t)img * A.ws_child_stride + ch.u_off;
        }
    }
    for (int e = tid; e < 2 * 128; e += 256) {
        const int ci = e >> 7, r = e & 127;
        iv[e] = (has && bc[ci] > 0 && r < f) ? A.inv[v.inv_off + ci * f + r] : -1;
    }
    for (int e = tid; e < ld * 16 + 16; e += 256) S[e] = 0.0;
    __syncthreads();
    for (int k = p + tid; k < 16; k += 256) S[k + ld * k] = 1.0;    // identity padding of the pivot block
    {   // matrix entries of this front (unique targets, all in pivot columns)
        const double* pl = A.planes + (size_t)img * A.n;
        for (int e = tid; e < v.orig_cnt; e += 256) {
            const int4 o = A.orig[v.orig_off + e];
            const int r = o.x < p ? o.x : o.x + sh;
            S[r + ld * o.y] += pl[(size_t)(o.z & 15) * A.tot + o.w];
        }
    }
    __syncthreads();
    // the children's entries of the pivot columns: child 0, then child 1 (panel entry (R, c), R >= c, c < p)
    for (int e0 = 0; e0 < f * p; e0 += 256 * 4) {
        double g[4][2];
        int rr[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + 256 * u + tid, f * p - 1);
            cc[u] = e / f; rr[u] = e - cc[u] * f;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int a = iv[ci * 128 + rr[u]], c = iv[ci * 128 + cc[u]];
                const bool ok = a >= 0 && c >= 0 && rr[u] >= cc[u];
                g[u][ci] = Uc[ci][ok ? a + bc[ci] * c : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e0 + 256 * u + tid >= f * p || rr[u] < cc[u]) continue;
            double* t = S + (rr[u] < p ? rr[u] : rr[u] + sh) + ld * cc[u];
            double x = *t;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const bool ok = iv[ci * 128 + rr[u]] >= 0 && iv[ci * 128 + cc[u]] >= 0;
                if (ok) x += g[u][ci];
            }
            *t = x;
        }
    }
    __syncthreads();
    // update-matrix tiles (i, j), 1 <= j <= i < P, dealt to the waves; wave 0 factors the panel first
    const int m = P - 1, ntile = m * (m + 1) / 2;
    auto tile_of = [&](int t, int& i, int& j) {
        int a = 0, c = t;
        while (c > a) { c -= a + 1; ++a; }
        i = 1 + a; j = 1 + c;
    };
    auto request = [&](int t, double (&g)[4][2]) {
        int i, j;
        tile_of(min(t, ntile - 1), i, j);
        const int C = 16 * j + lr - sh;                              // front row / column indices
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = 16 * i + lk + 4 * q - sh;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int a = R < f ? iv[ci * 128 + R] : -1, c = C < f ? iv[ci * 128 + C] : -1;
                g[q][ci] = Uc[ci][(a >= 0 && c >= 0 && R >= C) ? a + bc[ci] * c : 0];
            }
        }
    };
    double gq[4][2];
    const int t0 = wave == 0 ? 3 : wave - 1;                         // wave 0 joins after the factorisation
    if (t0 < ntile) request(t0, gq);
    bool bad = false;
    if (wave == 0) {
        if (MP > 64) bad = bcr_panel_factor<true>(S, ld, MP, 0, lane, dinv, p);
        else bad = bcr_panel_factor<false>(S, ld, MP, 0, lane, dinv, p);
        if (bad && lane == 0 && A.fail[img] == 0) A.fail[img] = node + 1;
    }
    __syncthreads();
    if (wave == 0) bcr_tile_inverse(S, ld, lane, dinv);
    double* U = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
    for (int t = t0; t < ntile; t += 4) {
        int i, j;
        tile_of(t, i, j);
        const int C = 16 * j + lr - sh;
        bcr_d4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = 16 * i + lk + 4 * q - sh;
            double x = 0.0;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int a = R < f ? iv[ci * 128 + R] : -1, c = C < f ? iv[ci * 128 + C] : -1;
                if (a >= 0 && c >= 0 && R >= C) x += gq[q][ci];
            }
            acc[q] = x;
        }
        if (t + 4 < ntile) request(t + 4, gq);                       // the next tile's entries fly during this tile's products
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double x = S[(16 * i + lr) + ld * (4 * kk + lk)];
            const double y = S[(16 * j + lr) + ld * (4 * kk + lk)];
            acc = bcr_mfma(-x, y, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = 16 * i + lk + 4 * q - sh;
            if (R < f && C < f && R >= C) U[(R - p) + (size_t)b * (C - p)] = acc[q];
        }
    }
    __syncthreads();                                                // wave 0's inverse before the factor columns go out
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    for (int c = wave; c < p; c += 4)
        for (int r = lane; r < f; r += 64) {
            double x;
            if (r < p) x = (r >= c) ? S[r + ld * c] : 0.0;
            else x = S[(r + sh) + ld * c];
            fc[r + (size_t)f * c] = x;
        }
}

// The same for fronts of 17 .. 32 pivots and up to 256 rows (1024^2: the two lowest levels of the large regime, 12 288 fronts per
// eight images, which cost five launches per level -- gather, matrix entries, potrf, trsm, Schur -- and three passes over
// their update matrices): TWO pivot block columns in LDS (MP x 32, <= 58 KB).  Block column 0 is factored by wave 0 for its
// first 64 rows (bcr_panel_factor) and by one thread per row below them (the row-wise pass bcr_panel_factor itself uses for
// rows 64 ..), block column 1 receives -L(:, 0) L(1, 0)^T on the MFMA and is factored the same way; W = L11^-1 (two tile
// inverses and one off-diagonal tile) is formed by wave 0 while the other waves already produce update-matrix tiles:
// gathered children's entries as the accumulator, minus two block columns of products.
// grid (fronts of the level, nimg), block 256; dynamic LDS nd_skinny2_lds(MPmax).
inline size_t nd_skinny2_lds(int MP) { return sizeof(double) * ((size_t)(MP + 1) * 32 + 32) + sizeof(int) * 2 * 256; }

// rows 16 pc + 64 .. MP - 1 of block column pc against its finished diagonal tile (strictly lower entries L(q, c) in the tile,
// 1 / l_cc in dinv): right-looking over the 16 columns, one thread per row -- the arithmetic of bcr_panel_factor<true>'s second half
__device__ __forceinline__ void nd_panel_rows_below(double* __restrict__ S, int ld, int MP, int pc, const double* __restrict__ dinv, int tid, int nthr) {
    const double* Lt = S + 16 * pc + ld * (16 * pc);
    for (int r = 16 * pc + 64 + tid; r < MP; r += nthr) {
        double m[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) m[c] = S[r + ld * (16 * pc + c)];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            m[c] *= dinv[16 * pc + c];
#pragma unroll
            for (int q = c + 1; q < 16; ++q) m[q] = __builtin_fma(-m[c], Lt[q + ld * c], m[q]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) S[r + ld * (16 * pc + c)] = m[c];
    }
}

__global__ __launch_bounds__(256, 2) void nd_front_skinny2_kernel(NdArgs A) {
    extern __shared__ double S[];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b, sh = 32 - p;              // boundary row i of the front sits at panel row 32 + i
    const int MP = 32 + nd_up16(b), ld = MP + 1, P = MP >> 4;
    double* dinv = S + (size_t)ld * 32;
    int* iv = reinterpret_cast<int*>(dinv + 32);                     // iv[ci * 256 + front row]: index in child ci's boundary, or -1
    const bool has = v.inv_off >= 0;
    const double* Uc[2] = {A.ws_child, A.ws_child};
    int bc[2] = {0, 0};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (has && cn >= 0) {
            const NdNodeDev ch = A.nodes[cn];
            bc[ci] = ch.b;
            Uc[ci] = A.ws_child + (size_t)img * A.ws_child_stride + ch.u_off;
        }
    }
    for (int e = tid; e < 2 * 256; e += 256) {
        const int ci = e >> 8, r = e & 255;
        iv[e] = (has && bc[ci] > 0 && r < f) ? A.inv[v.inv_off + ci * f + r] : -1;
    }
    for (int e = tid; e < ld * 32 + 32; e += 256) S[e] = 0.0;
    __syncthreads();
    for (int k = p + tid; k < 32; k += 256) S[k + ld * k] = 1.0;    // identity padding of the pivot block
    {   // matrix entries of this front (unique targets, all in pivot columns)
        const double* pl = A.planes + (size_t)img * A.n;
        for (int e = tid; e < v.orig_cnt; e += 256) {
            const int4 o = A.orig[v.orig_off + e];
            const int r = o.x < p ? o.x : o.x + sh;
            S[r + ld * o.y] += pl[(size_t)(o.z & 15) * A.tot + o.w];
        }
    }
    __syncthreads();
    // the children's entries of the pivot columns: child 0, then child 1 (panel entry (R, c), R >= c, c < p)
    for (int e0 = 0; e0 < f * p; e0 += 256 * 4) {
        double g[4][2];
        int rr[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + 256 * u + tid, f * p - 1);
            cc[u] = e / f; rr[u] = e - cc[u] * f;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int a = iv[ci * 256 + rr[u]], c = iv[ci * 256 + cc[u]];
                const bool ok = a >= 0 && c >= 0 && rr[u] >= cc[u];
                g[u][ci] = Uc[ci][ok ? a + bc[ci] * c : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e0 + 256 * u + tid >= f * p || rr[u] < cc[u]) continue;
            double* t = S + (rr[u] < p ? rr[u] : rr[u] + sh) + ld * cc[u];
            double x = *t;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const bool ok = iv[ci * 256 + rr[u]] >= 0 && iv[ci * 256 + cc[u]] >= 0;
                if (ok) x += g[u][ci];
            }
            *t = x;
        }
    }
    __syncthreads();
    // update-matrix tiles (i, j), 2 <= j <= i < P
    const int m = P - 2, ntile = m * (m + 1) / 2;
    auto tile_of = [&](int t, int& i, int& j) {
        int a = 0, c = t;
        while (c > a) { c -= a + 1; ++a; }
        i = 2 + a; j = 2 + c;
    };
    auto request = [&](int t, double (&g)[4][2]) {
        int i, j;
        tile_of(min(t, ntile - 1), i, j);
        const int C = 16 * j + lr - sh;                              // front row / column indices
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = 16 * i + lk + 4 * q - sh;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int a = R < f ? iv[ci * 256 + R] : -1, c = C < f ? iv[ci * 256 + C] : -1;
                g[q][ci] = Uc[ci][(a >= 0 && c >= 0 && R >= C) ? a + bc[ci] * c : 0];
            }
        }
    };
    double gq[4][2];
    const int t0 = wave == 0 ? 3 : wave - 1;                         // wave 0 joins after the inverse of the pivot block
    if (ntile > 0 && t0 < ntile) request(t0, gq);
    bool bad = false;
    // block column 0: first 64 rows by wave 0, the rows below them one thread each
    if (wave == 0) bad |= bcr_panel_factor<false>(S, ld, min(MP, 64), 0, lane, dinv, min(p, 16));
    __syncthreads();
    nd_panel_rows_below(S, ld, MP, 0, dinv, tid, 256);
    __syncthreads();
    // block column 1 receives block column 0 (left-looking), then the same
    for (int i = 1 + wave; i < P; i += 4) {
        bcr_d4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = S[(16 * i + lk + 4 * q) + ld * (16 + lr)];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double x = S[(16 * i + lr) + ld * (4 * kk + lk)];
            const double y = S[(16 + lr) + ld * (4 * kk + lk)];
            acc = bcr_mfma(-x, y, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) S[(16 * i + lk + 4 * q) + ld * (16 + lr)] = acc[q];
    }
    __syncthreads();
    if (wave == 0) {
        bad |= bcr_panel_factor<false>(S, ld, min(MP, 80), 1, lane, dinv, p - 16);
        if (bad && lane == 0 && A.fail[img] == 0) A.fail[img] = node + 1;
    }
    __syncthreads();
    nd_panel_rows_below(S, ld, MP, 1, dinv, tid, 256);
    __syncthreads();
    if (wave == 0) {   // W = L11^-1: the two diagonal tiles, then tile (1, 0) into the upper tile (0, 1)
        bcr_tile_inverse(S, ld, lane, dinv);
        bcr_tile_inverse(S + 16 + ld * 16, ld, lane, dinv + 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bcr_winv_tile(S, ld, 1, 0, lr, lk);
    }
    double* U = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
    for (int t = t0; t < ntile; t += 4) {
        int i, j;
        tile_of(t, i, j);
        const int C = 16 * j + lr - sh;
        bcr_d4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = 16 * i + lk + 4 * q - sh;
            double x = 0.0;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int a = R < f ? iv[ci * 256 + R] : -1, c = C < f ? iv[ci * 256 + C] : -1;
                if (a >= 0 && c >= 0 && R >= C) x += gq[q][ci];
            }
            acc[q] = x;
        }
        if (t + 4 < ntile) request(t + 4, gq);                       // the next tile's entries fly during this tile's products
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double x = S[(16 * i + lr) + ld * (16 * qq + 4 * kk + lk)];
                const double y = S[(16 * j + lr) + ld * (16 * qq + 4 * kk + lk)];
                acc = bcr_mfma(-x, y, acc);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int R = 16 * i + lk + 4 * q - sh;
            if (R < f && C < f && R >= C) U[(R - p) + (size_t)b * (C - p)] = acc[q];
        }
    }
    __syncthreads();                                                // wave 0's inverse before the factor columns go out
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    for (int c = wave; c < p; c += 4)
        for (int r = lane; r < f; r += 64) {
            double x;
            if (r < p) x = (r >= c) ? S[(16 * (c >> 4) + (r & 15)) + ld * (16 * (r >> 4) + (c & 15))] : 0.0;
            else x = S[(r + sh) + ld * c];
            fc[r + (size_t)f * c] = x;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// small regime, one WAVE per front: f <= F <= 64 rows (one per lane), p <= P <= 32 pivots
// ------------------------------------------------------------------------------------------------------------------
// The bottom levels of the tree are hundreds of thousands of fronts of a few dozen rows; nd_front_small_kernel spends
// its time between workgroup barriers with one wave of two or four working.  Here a front belongs to ONE wave and, once
// assembled, lives in registers: lane r holds row r, register j column j (lower triangle, a[j] of lane r = F(r, j)).
// Right-looking elimination, step k: the scaled column l = F(:, k) / sqrt(d) is replicated into every row of 16 lanes
// (ds_bpermute), and column j receives a[j] -= l * l_j with l_j taken by the DPP row broadcast of v_fmac_f64 -- one
// VALU instruction per entry, no SGPR round trip (tools/dpp_probe.hip: 1.17 x a plain v_fmac_f64).  W = L11^-1 rides
// along: L = Lt D (Lt unit lower, D = diag l_kk), X = Lt^-1 is built by the same row operations on the identity, held
// TRANSPOSED (lane c, register r = X(r, c)) so that its multipliers lt_r = l_r / l_kk are the same kind of broadcast
// and the pivot row X(k, :) is the lane's own register; W(k, :) = X(k, :) / l_kk is final after step k - 1.
// Register indices are static: column k and row k are picked by a switch over k (P cases), whole blocks of eight
// columns left of k or right of f are skipped by wave-uniform branches.  W and L21 are staged in LDS in the layout
// of the factor (f x p, leading dimension f) and leave as one linear copy; U goes out of the registers column by column.
// LDS: the packed lower triangle of the front while it is assembled (targets are data dependent), then the staging area.
__device__ __forceinline__ int nd_tri(int F, int r, int c) { return r + ((c * (2 * F - 1 - c)) >> 1); }   // r >= c

template <int N>
__device__ __forceinline__ void nd_fmac_bc(double& acc, double src, double mul) {   // acc += src[lane 16 * (lane / 16) + N] * mul
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}
template <int J0, int NA, int NR, int... I>
__device__ __forceinline__ void nd_fmac_block(double (&a)[NA], const double (&rep)[NR], double mul, std::integer_sequence<int, I...>) {
    (nd_fmac_bc<(J0 + I) & 15>(a[J0 + I], rep[(J0 + I) >> 4], mul), ...);
}
// blocks lo <= jb < hi of eight registers each (lo, hi wave-uniform)
template <int NA, int NR, int... JB>
__device__ __forceinline__ void nd_fmac_blocks(double (&a)[NA], const double (&rep)[NR], double mul, int lo, int hi, std::integer_sequence<int, JB...>) {
    ((JB >= lo && JB < hi ? nd_fmac_block<8 * JB>(a, rep, mul, std::make_integer_sequence<int, 8>()) : (void)0), ...);
}
__device__ __forceinline__ double nd_bperm_f64(int addr, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int P, int NA, int NX>
__device__ __forceinline__ void nd_pick(const double (&a)[NA], const double (&xt)[NX], int k, double& ak, double& xk) {
    // a branch per case (the empty asm keeps the compiler from turning the cases into 2 P selects)
    switch (k) {
        default: ak = a[0]; xk = xt[0]; asm volatile("" : "+v"(ak), "+v"(xk)); break;
#define ND_PICK_CASE(c) case c: if (c < P) { ak = a[c < NA ? c : 0]; xk = xt[c < NX ? c : 0]; asm volatile("" : "+v"(ak), "+v"(xk)); } break;
        ND_PICK_CASE(1) ND_PICK_CASE(2) ND_PICK_CASE(3) ND_PICK_CASE(4) ND_PICK_CASE(5) ND_PICK_CASE(6) ND_PICK_CASE(7)
        ND_PICK_CASE(8) ND_PICK_CASE(9) ND_PICK_CASE(10) ND_PICK_CASE(11) ND_PICK_CASE(12) ND_PICK_CASE(13) ND_PICK_CASE(14) ND_PICK_CASE(15)
        ND_PICK_CASE(16) ND_PICK_CASE(17) ND_PICK_CASE(18) ND_PICK_CASE(19) ND_PICK_CASE(20) ND_PICK_CASE(21) ND_PICK_CASE(22) ND_PICK_CASE(23)
        ND_PICK_CASE(24) ND_PICK_CASE(25) ND_PICK_CASE(26) ND_PICK_CASE(27) ND_PICK_CASE(28) ND_PICK_CASE(29) ND_PICK_CASE(30) ND_PICK_CASE(31)
#undef ND_PICK_CASE
    }
}

template <int F, int P>
constexpr int nd_wave_lds_doubles() { return (F * (F + 1) / 2 > F * P ? F * (F + 1) / 2 : F * P) + 2; }

// grid (fronts of the level, nimg), block 64.  Fronts with f <= F and p <= P only (the host picks the instance per level).
// Every global load of the assembly is requested before the first one is consumed (matrix-entry list, both children's
// maps and the first NDW_U x 64 entries of each update matrix); the pivot of step k + 1 is brought up to date and its
// rsqrt chain started while the replicated column of step k is still on its way through the LDS crossbar.
constexpr int NDW_U = 16;   // update-matrix entries in flight per lane and child
template <int F, int P>
__global__ __launch_bounds__(64) void nd_front_wave_kernel(NdArgs A) {
    static_assert(F % 8 == 0 && F <= 64 && P % 8 == 0 && P <= 32 && P <= F, "one row per lane, pivots picked by a 32-way switch");
    constexpr int NQ = (F + 15) / 16, NQX = (P + 15) / 16, TRI = F * (F + 1) / 2;
    __shared__ __attribute__((aligned(16))) double S[nd_wave_lds_doubles<F, P>()];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
    ND_PROBE(0);
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    // requests: matrix-entry list (first four per lane), children's descriptors, maps and update matrices
    int4 o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        o[u] = e < v.orig_cnt ? A.orig[v.orig_off + e] : make_int4(0, 0, 0, 0);
    }
    NdNodeDev ch[2];
    int bc[2] = {0, 0}, cmv[2] = {0, 0};
    const double* Uc[2] = {nullptr, nullptr};
    double x[2][NDW_U];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (cn >= 0) { ch[ci] = A.nodes[cn]; bc[ci] = ch[ci].b; }
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int n = bc[ci], nn = n * n;
        if (n > 0) {
            Uc[ci] = A.ws_child + (size_t)img * A.ws_child_stride + ch[ci].u_off;
            cmv[ci] = lane < n ? A.cmap[ch[ci].cmap_off + lane] : 0;
        }
#pragma unroll
        for (int u = 0; u < NDW_U; ++u) {
            const int e = 64 * u + lane;
            x[ci][u] = e < nn ? Uc[ci][e] : 0.0;
        }
    }
    const double* pl = A.planes + (size_t)img * A.n;
    double ov[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ov[u] = lane + 64 * u < v.orig_cnt ? pl[(size_t)(o[u].z & 15) * A.tot + o[u].w] : 0.0;
    for (int e = 2 * lane; e < TRI; e += 128) *reinterpret_cast<double2*>(S + e) = make_double2(0.0, 0.0);
    ND_PROBE(1);
    // matrix entries of this front (unique targets, lower triangle: the front is ordered by elimination index)
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < v.orig_cnt) S[nd_tri(F, o[u].x, o[u].y)] += ov[u];
    for (int e = lane + 256; e < v.orig_cnt; e += 64) {
        const int4 oo = A.orig[v.orig_off + e];
        S[nd_tri(F, oo.x, oo.y)] += pl[(size_t)(oo.z & 15) * A.tot + oo.w];
    }
    ND_PROBE(2);
    // the children's update matrices, read linearly (b x b, lower triangle valid), scattered through cmap: child 0, then
    // child 1, each entry of a child to its own target -- the order of the additions is that of nd_front_small_kernel
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int n = bc[ci], nn = n * n;
        if (n == 0) continue;
        const unsigned magic = 0xFFFFFFFFu / (unsigned)n + 1u;     // e / n = umulhi(e, magic) for e < 2^16, n <= 64
        for (int e0 = 0; e0 < nn; e0 += 64 * NDW_U) {
            if (e0 > 0) {
#pragma unroll
                for (int u = 0; u < NDW_U; ++u) {
                    const int e = e0 + 64 * u + lane;
                    x[ci][u] = e < nn ? Uc[ci][e] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < NDW_U; ++u) {
                if (e0 + 64 * u >= nn) break;
                const int e = min(e0 + 64 * u + lane, nn - 1);
                const int j = (int)__umulhi((unsigned)e, magic), i = e - j * n;
                const int row = __builtin_amdgcn_ds_bpermute(i << 2, cmv[ci]), col = __builtin_amdgcn_ds_bpermute(j << 2, cmv[ci]);
                if (e0 + 64 * u + lane < nn && i >= j) S[nd_tri(F, row, col)] += x[ci][u];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a[F], xt[P];
#pragma unroll
    for (int j = 0; j < F; ++j) a[j] = (lane >= j && lane < F) ? S[lane + ((j * (2 * F - 1 - j)) >> 1)] : 0.0;
#pragma unroll
    for (int r = 0; r < P; ++r) xt[r] = lane == r ? 1.0 : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                 // the triangle is in registers: S becomes the staging area
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    ND_PROBE(3);
    const int bpa = (lane & 15) << 2;
    const int jhi = (f + 7) >> 3, rhi = (p + 7) >> 3;
    bool bad = false;
    double ak = a[0], xk = xt[0], y = 1.0, sq;
    if (p > 0) {
        const double d = readlane_f64(ak, 0);
        if (!(d > 0.0)) bad = true;
        sqrt_rsqrt(d, sq, y);
    }
    for (int k = 0; k < p; ++k) {
        const double l = ak * y;
        const double lz = lane > k ? l : 0.0;
        if (lane >= p && lane < f) S[lane + f * k] = l;          // L21(:, k)
        if (lane < p) S[k + f * lane] = xk * y;                  // W(k, :), zero right of the diagonal
        double rep[NQ], rept[NQX];
#pragma unroll
        for (int q = 0; q < NQ; ++q) rep[q] = nd_bperm_f64(bpa + (q << 6), lz);
        double nl = -lz, nxk = -xk;
        // step k + 1's pivot column and pivot row of X, brought up to date here (the same fused operations the blocks
        // below apply to a[k + 1] and xt[k + 1]) so that its rsqrt chain runs beside the broadcasts and the block updates
        double ak1 = 0.0, xk1 = 0.0, y1 = 1.0;
        if (k + 1 < p) {
            nd_pick<P>(a, xt, k + 1, ak1, xk1);
            const double l1 = readlane_f64(lz, k + 1);
            ak1 = __builtin_fma(l1, nl, ak1);
            xk1 = __builtin_fma(l1 * y, nxk, xk1);
            const double d1 = readlane_f64(ak1, k + 1);
            if (!(d1 > 0.0)) bad = true;
            sqrt_rsqrt(d1, sq, y1);
        }
#pragma unroll
        for (int q = 0; q < NQX; ++q) rept[q] = rep[q] * y;
        // a VALU result read through DPP needs two wait states; the asm below is opaque to the hazard recogniser
        if constexpr (NQX == 1) asm volatile("s_nop 1" : "+v"(rept[0]), "+v"(nl), "+v"(nxk));
        else asm volatile("s_nop 1" : "+v"(rept[0]), "+v"(rept[1]), "+v"(nl), "+v"(nxk));
        const int lo = (k + 1) >> 3;
        nd_fmac_blocks(a, rep, nl, lo, jhi, std::make_integer_sequence<int, F / 8>());
        nd_fmac_blocks(xt, rept, nxk, lo, rhi, std::make_integer_sequence<int, P / 8>());
        ak = ak1; xk = xk1; y = y1;
    }
    ND_PROBE(4);
    if (bad && lane == 0 && A.fail[img] == 0) A.fail[img] = node + 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    for (int e = lane; e < f * p; e += 64) fc[e] = S[e];
    double* U = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
#pragma unroll
    for (int j = 0; j < F; ++j)
        if (j >= p && j < f && lane >= j && lane < f) U[(lane - p) + b * (j - p)] = a[j];
    ND_PROBE(5);
}

// ------------------------------------------------------------------------------------------------------------------
// large regime: the front lives in HBM -- factor columns (f x p) and the update-matrix slot (b x b)
// ------------------------------------------------------------------------------------------------------------------
// Assembly of a large front in GATHER form: every entry (R, C), R >= C, of the front -- factor columns and update-matrix
// slot -- is written exactly once as the sum of the children's update-matrix entries that map to it (child 0, then child
// 1; zero where neither reaches), through the parent -> child maps `inv`.  No zero pass, no read-modify-write, no
// atomics; nd_orig_kernel adds the few matrix entries afterwards.  A wave per front column, lanes down the rows: the
// maps are monotone, so a wave reads runs of consecutive child entries.  grid (blocks, nodes, nimg), block 256.
// Round 4: a wave takes NDG_C consecutive columns at a time; the row maps of a chunk of rows are loaded once for all of
// them, and every load is unconditional (clamped index, select afterwards) -- with a load under an `if` the compiler parks
// each one behind its own exec mask and waits for it there, so the "chunks in flight" of the earlier form were one.
constexpr int NDG_U = 4, NDG_C = 4;
__global__ __launch_bounds__(256) void nd_gather_kernel(NdArgs A) {
    const int node = A.node0 + blockIdx.y, img = blockIdx.z;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    double* U = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
    const bool has = v.inv_off >= 0;
    const int* inv0 = A.inv + (has ? v.inv_off : 0);
    const int* inv1 = inv0 + (has ? f : 0);
    const double *U0 = A.ws_child, *U1 = A.ws_child;     // a valid address also without the child: nothing read from it is used
    int b0 = 0, b1 = 0;
    bool h0 = false, h1 = false;
    if (has && v.child0 >= 0) { const NdNodeDev ch = A.nodes[v.child0]; U0 = A.ws_child + (size_t)img * A.ws_child_stride + ch.u_off; b0 = ch.b; h0 = true; }
    if (has && v.child1 >= 0) { const NdNodeDev ch = A.nodes[v.child1]; U1 = A.ws_child + (size_t)img * A.ws_child_stride + ch.u_off; b1 = ch.b; h1 = true; }
    const int lane = threadIdx.x & 63;
    for (int Cg = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NDG_C; Cg < f; Cg += gridDim.x * 4 * NDG_C) {
        int c0[NDG_C], c1[NDG_C];
#pragma unroll
        for (int q = 0; q < NDG_C; ++q) {
            const int C = min(Cg + q, f - 1);
            c0[q] = h0 ? inv0[C] : -1;                  // wave-uniform
            c1[q] = h1 ? inv1[C] : -1;
        }
        for (int R0 = Cg; R0 < f; R0 += 64 * NDG_U) {
            int a0[NDG_U], a1[NDG_U];
#pragma unroll
            for (int u = 0; u < NDG_U; ++u) {
                const int R = min(R0 + lane + 64 * u, f - 1);
                a0[u] = inv0[has ? R : 0];
                a1[u] = inv1[has ? R : 0];
            }
            double x0[NDG_C][NDG_U], x1[NDG_C][NDG_U];
#pragma unroll
            for (int q = 0; q < NDG_C; ++q)
#pragma unroll
                for (int u = 0; u < NDG_U; ++u) {
                    const bool k0 = h0 && c0[q] >= 0 && a0[u] >= 0, k1 = h1 && c1[q] >= 0 && a1[u] >= 0;
                    x0[q][u] = U0[k0 ? (size_t)a0[u] + (size_t)b0 * c0[q] : 0];
                    x1[q][u] = U1[k1 ? (size_t)a1[u] + (size_t)b1 * c1[q] : 0];
                }
#pragma unroll
            for (int q = 0; q < NDG_C; ++q) {
                const int C = Cg + q;
                double* dst = (C < p) ? fc + (size_t)f * C : U + (size_t)b * (C - p) - p;   // column C of the front, indexed by front row
#pragma unroll
                for (int u = 0; u < NDG_U; ++u) {
                    const int R = R0 + lane + 64 * u;
                    const bool k0 = h0 && c0[q] >= 0 && a0[u] >= 0, k1 = h1 && c1[q] >= 0 && a1[u] >= 0;
                    const double acc = (k0 ? x0[q][u] : 0.0) + (k1 ? x1[q][u] : 0.0);
                    if (C < f && R < f && R >= C) dst[R] = acc;
                }
            }
        }
    }
}

// matrix entries: every target lies in the factor columns (the column index is a pivot).  grid (nodes, nimg).
__global__ __launch_bounds__(256) void nd_orig_kernel(NdArgs A) {
    const int node = A.node0 + blockIdx.x, img = blockIdx.y;
    const NdNodeDev v = A.nodes[node];
    const int f = v.p + v.b;
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    const double* pl = A.planes + (size_t)img * A.n;
    const double* pl2 = A.planes2 + (size_t)img * A.n;   // entries whose row is the smaller pixel (ND_ORIG_UPPER)
    for (int e = threadIdx.x; e < v.orig_cnt; e += 256) {
        const int4 o = A.orig[v.orig_off + e];
        fc[o.x + (size_t)f * o.y] += ((o.z & 16) ? pl2 : pl)[(size_t)(o.z & 15) * A.tot + o.w];
    }
}

// Cholesky + inverse of the diagonal block of pivot panel k (columns [128 k, 128 k + nb)) in LDS; the block is
// replaced by W = L^-1 (zeros above the diagonal).  grid (nodes, nimg), block BCR_PT, LDS bcr_potrf_lds(128).
__global__ __launch_bounds__(BCR_PT) void nd_potrf_kernel(NdArgs A, int k) {
    extern __shared__ double S[];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, f = p + v.b, c0 = HB2_NB * k;
    if (c0 >= p) return;
    const int nb = min(HB2_NB, p - c0), MP = nd_up16(nb), ld = MP + 1;
    double* blk = A.fac + (size_t)img * A.fac_stride + v.fac_off + c0 + (size_t)f * c0;
    for (int e = tid; e < MP * MP; e += BCR_PT) {
        const int r = e % MP, c = e / MP;
        double x = (r == c) ? 1.0 : 0.0;
        if (r < nb && c < nb) x = (r >= c) ? blk[r + (size_t)f * c] : 0.0;
        S[r + ld * c] = x;
    }
    __syncthreads();
    const bool bad = bcr_potrf_lds_body(S, MP);
    if (bad && (tid & 63) == 0 && A.fail[img] == 0) A.fail[img] = node + 1;
    for (int e = tid; e < nb * nb; e += BCR_PT) {
        const int r = e % nb, c = e / nb;
        blk[r + (size_t)f * c] = (r >= c) ? S[(16 * (c >> 4) + (r & 15)) + ld * (16 * (r >> 4) + (c & 15))] : 0.0;
    }
}

// Rows below the diagonal block of panel k: L(r, c) = sum_kk A(r, kk) W(c, kk), in place.  One workgroup owns 64 rows
// and computes both 64-column halves before it writes (its rows are read by nobody else).
// grid (row tiles, nodes, nimg), block BG_T.
// dense != 0 (LU variant): W is the full transposed inverse of the diagonal block, not a triangle.
__global__ __launch_bounds__(BG_T) void nd_trsm_kernel(NdArgs A, int k, int dense) {
    __shared__ double lds[BG_LDS];
    const int node = A.node0 + blockIdx.y, img = blockIdx.z, tid = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, f = p + v.b, c0 = HB2_NB * k;
    if (c0 >= p) return;
    const int nb = min(HB2_NB, p - c0);
    const int R0 = c0 + nb + 64 * (int)blockIdx.x;
    if (R0 >= f) return;
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    const double* Ar = fc + R0 + (size_t)f * c0;           // A(R0 + r, c0 + kk) at Ar[r + f kk]
    const double* Wb = fc + c0 + (size_t)f * c0;           // W(c, kk) at Wb[c + f kk]
    const int rmax = f - R0;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    BgAcc acc[2];
    const int nhalf = nb > 64 ? 2 : 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // unrolled: the accumulators stay in registers
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[h].c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
        if (h >= nhalf) continue;
        const int cc0 = 64 * h;
        const int kend = dense ? nb : min(nb, cc0 + 64);    // Cholesky: W(c, kk) = 0 for kk > c
        const int nchunk = (kend + BG_KC - 1) / BG_KC;
        double va[8], vb[8];
        hb2_fetch_dense(Ar, f, rmax, kend, 0, 0, tid, va);
        hb2_fetch_dense(Wb, f, nb, kend, cc0, 0, tid, vb);
        for (int ch = 0; ch < nchunk; ++ch) {
            __syncthreads();
            bg_stage<true>(As, tid, va);
            bg_stage<true>(Bs, tid, vb);
            __syncthreads();
            if (ch + 1 < nchunk) {
                hb2_fetch_dense(Ar, f, rmax, kend, 0, (ch + 1) * BG_KC, tid, va);
                hb2_fetch_dense(Wb, f, nb, kend, cc0, (ch + 1) * BG_KC, tid, vb);
            }
            hb2_mma_chunk(As, Bs, acc[h]);
        }
    }
    const int l = tid & 63, hq = tid >> 6;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h >= nhalf) continue;
        bg_to_lds(acc[h], lds);      // barriers inside: every read of A by this workgroup has completed before the first write
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int o = hq + 4 * i, c = 64 * h + o;
            if (l < rmax && c < nb) fc[(R0 + l) + (size_t)f * (c0 + c)] = lds[o * BG_LD + l];
        }
    }
}

// A(R, C) -= sum_kk X(R, kk) X(C, kk) on 64 x 64 tiles of a lower-triangular region, MFMA.  Shared by the pivot-column
// update (nd_syrk_kernel) and the update matrix (nd_schur_kernel).
//   X: rows relative to Xb (leading dimension ldx), nrow rows, depth kdepth
//   Out(R, C) at Ob[R + ldo C], R in [0, nrow), C in [0, ncol), R + roff >= C is written (roff: row offset of the
//   output region against its column numbering, 0 for square regions)
// Yb: the second operand, A(R, C) -= sum_kk X(R, kk) Y(C, kk) (LU variant: the other triangle's factor columns); = Xb for
// Cholesky.
__device__ __forceinline__ void nd_tile_rank_update(const double* __restrict__ Xb, const double* __restrict__ Yb, int ldx, int nrow, int kdepth,
                                                    int ta, int tb, double* __restrict__ Ob, size_t ldo, int ncol, double* __restrict__ lds) {
    const int tid = threadIdx.x;
    double* As = lds;
    double* Bs = lds + BG_KC * BG_LD;
    BgAcc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.c[i >> 1][i & 1] = bcr_d4{0.0, 0.0, 0.0, 0.0};
    const int nchunk = (kdepth + BG_KC - 1) / BG_KC;
    double va[8], vb[8];
    hb2_fetch_dense(Xb, ldx, nrow, kdepth, 64 * ta, 0, tid, va);
    hb2_fetch_dense(Yb, ldx, nrow, kdepth, 64 * tb, 0, tid, vb);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        bg_stage<true>(As, tid, va);
        bg_stage<true>(Bs, tid, vb);
        __syncthreads();
        if (ch + 1 < nchunk) {
            hb2_fetch_dense(Xb, ldx, nrow, kdepth, 64 * ta, (ch + 1) * BG_KC, tid, va);
            hb2_fetch_dense(Yb, ldx, nrow, kdepth, 64 * tb, (ch + 1) * BG_KC, tid, vb);
        }
        hb2_mma_chunk(As, Bs, acc);
    }
    const int l = tid & 63, hq = tid >> 6;
    const int R = 64 * ta + l;
    double old[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int C = 64 * tb + hq + 4 * i;
        old[i] = (R < nrow && C < ncol && R >= C) ? Ob[R + ldo * C] : 0.0;
    }
    bg_to_lds(acc, lds);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = hq + 4 * i, C = 64 * tb + o;
        if (R < nrow && C < ncol && R >= C) Ob[R + ldo * C] = old[i] - lds[o * BG_LD + l];
    }
}
// lower tile index t -> (ta, tb), ta >= tb
__device__ __forceinline__ void nd_tri_decode(int t, int& ta, int& tb) {
    int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    ta = a; tb = t - a * (a + 1) / 2;
}

// remaining pivot columns behind panel k: rows and columns counted from s0 = 128 (k + 1); columns < p.
// grid (lower tiles of ceil((f - s0) / 64), nodes, nimg).
__global__ __launch_bounds__(BG_T) void nd_syrk_kernel(NdArgs A, int k) {
    __shared__ double lds[BG_LDS];
    const int node = A.node0 + blockIdx.y, img = blockIdx.z;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, f = p + v.b, c0 = HB2_NB * k, s0 = c0 + HB2_NB;
    if (s0 >= p) return;
    int ta, tb;
    nd_tri_decode((int)blockIdx.x, ta, tb);
    const int nrow = f - s0, ncol = p - s0;
    if (64 * ta >= nrow || 64 * tb >= ncol) return;
    double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    const double* f2 = A.fac2 + (size_t)img * A.fac_stride + v.fac_off;
    nd_tile_rank_update(fc + s0 + (size_t)f * c0, f2 + s0 + (size_t)f * c0, f, nrow, HB2_NB, ta, tb, fc + s0 + (size_t)f * s0, (size_t)f, ncol, lds);
}

// U -= L21 L21^T (depth p).  grid (lower tiles of ceil(b / 64), nodes, nimg).
__global__ __launch_bounds__(BG_T) void nd_schur_kernel(NdArgs A) {
    __shared__ double lds[BG_LDS];
    const int node = A.node0 + blockIdx.y, img = blockIdx.z;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    int ta, tb;
    nd_tri_decode((int)blockIdx.x, ta, tb);
    if (64 * ta >= b) return;
    const double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    const double* f2 = A.fac2 + (size_t)img * A.fac_stride + v.fac_off;
    double* U = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
    nd_tile_rank_update(fc + p, f2 + p, f, b, p, ta, tb, U, (size_t)b, b, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// LU variant (structurally symmetric, numerically non-symmetric matrices: the row-scaled system of
// sumregs_gradient_reg with a patch parameter, /root/reference/src/SumRegsLearningFunction.jl:250).  Same tree, same
// maps.  A front keeps its two triangles apart -- FL: lower triangle and diagonal, FU: the strict upper triangle
// TRANSPOSED (FU(R, C) = F(C, R), R > C) -- each in the layout of the Cholesky front (factor columns f x p, update matrix
// b x b), so the gather, matrix-entry, trsm and rank-update kernels above run on either with the pointers exchanged.
// Block LU without pivoting with INVERTED diagonal blocks, per 128-column pivot panel k (D = F_kk):
//     L'(R, k) = F(R, k) D^-1   (rows below, nd_trsm_kernel with the dense W = D^-T),   U(k, C) = F(k, C) stays,
//     F(R, C) -= L'(R, k) U(k, C):  FL(R, C) -= L'(R, :) FU(C, :)^T  (R >= C),   FU(C, R) -= FU(C, :) L'(R, :)^T  (C > R).
// Substitutions: forward y_k = w_k, rows below lose L' y_k; backward x_k = D^-1 (y_k - U(k, after) x_after); D^-T is
// kept in the diagonal slot of both triangles.
// ------------------------------------------------------------------------------------------------------------------
constexpr int NDG_T = 1024;
inline size_t nd_getri_lds() { return 0; }   // static LDS only (the block is held in registers)
// A pivot that is zero, not finite, or has lost every digit against the diagonal entry it started from means the
// elimination without pivoting has broken down (hb_getri_kernel's criterion).
__device__ __forceinline__ bool nd_lu_pivot_bad(double piv, double d0) { return !(fabs(piv) > 2.220446049250313e-16 * d0 && fabs(piv) < 1.7e308); }

// Inverse of the diagonal block of pivot panel k by Gauss-Jordan elimination without pivoting; its transpose replaces the
// block in both triangles.  A.fac = FL, A.fac2 = FU.  The 128 x 128 block lives in REGISTERS: thread (column j = tid & 127,
// row group g = tid >> 7) owns rows g + 8 m of column j.  A step needs the pivot row and the pivot column only: their
// owners publish them in two LDS vectors (double-buffered: one barrier per step), everybody else reads one pivot-row
// entry and -- as a wave-wide broadcast -- the sixteen pivot-column entries of its rows.  (The first version kept the
// matrix in LDS and moved 3 x 128 KB per step: 196 us per block against ~45 us.)  grid (nodes, nimg), block NDG_T.
__global__ __launch_bounds__(NDG_T) void nd_getri_kernel(NdArgs A, int k) {
    __shared__ double rowb[2][HB2_NB], colb[2][HB2_NB], d0[HB2_NB];
    __shared__ int badk_s;
    constexpr int MP = HB2_NB;
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, f = p + v.b, c0 = HB2_NB * k;
    if (c0 >= p) return;
    const int nb = min(HB2_NB, p - c0);
    double* bL = A.fac + (size_t)img * A.fac_stride + v.fac_off + c0 + (size_t)f * c0;
    double* bU = A.fac2 + (size_t)img * A.fac_stride + v.fac_off + c0 + (size_t)f * c0;
    const int j = tid & (MP - 1);
    const int g = __builtin_amdgcn_readfirstlane(tid >> 7);   // row group: uniform over a wave
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = g + 8 * m;
        double x = (i == j) ? 1.0 : 0.0;   // identity padding
        if (i < nb && j < nb) x = (i >= j) ? bL[i + (size_t)f * j] : bU[j + (size_t)f * i];
        a[m] = x;
        if (i == j) d0[i] = fabs(x);
        if (i == 0) rowb[0][j] = x;
        if (j == 0) { colb[0][i] = x; a[m] = 0.0; }   // a pivot column lives on in its published copy: the in-place
                                                     // inverse starts that column from zero
    }
    if (tid == 0) badk_s = -1;
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        const double piv = rowb[cur][kk];
        // a pivot that is zero, not finite or has lost every digit against its original diagonal entry: broken down
        if (tid == 0 && badk_s < 0 && nd_lu_pivot_bad(piv, d0[kk])) badk_s = kk;
        double pinv = __builtin_amdgcn_rcp(piv);              // 1 / piv by two Newton steps on the hardware estimate
        pinv = __builtin_fma(__builtin_fma(-piv, pinv, 1.0), pinv, pinv);
        pinv = __builtin_fma(__builtin_fma(-piv, pinv, 1.0), pinv, pinv);
        const double rk = (j == kk) ? pinv : rowb[cur][j] * pinv;
#pragma unroll
        for (int m = 0; m < 16; ++m) a[m] = __builtin_fma(-colb[cur][g + 8 * m], rk, a[m]);   // column reads: wave-wide broadcasts
        if (g == (kk & 7)) {          // the waves that hold the pivot row: it becomes rk
#pragma unroll
            for (int m = 0; m < 16; ++m)
                if (m == (kk >> 3)) a[m] = rk;
        }
        if (g == ((kk + 1) & 7)) {    // ... and those that hold the next pivot row publish it
#pragma unroll
            for (int m = 0; m < 16; ++m)
                if (m == ((kk + 1) >> 3)) rowb[nxt][j] = a[m];
        }
        if (j == kk + 1) {            // the next pivot column: publish, then restart it from zero
#pragma unroll
            for (int m = 0; m < 16; ++m) { colb[nxt][g + 8 * m] = a[m]; a[m] = 0.0; }
        }
        __syncthreads();
    }
    if (tid == 0 && badk_s >= 0 && A.fail[img] == 0) A.fail[img] = node + 1;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = g + 8 * m;
        if (i < nb && j < nb) {   // slot(r, c) = D^-1(c, r): this thread's D^-1(i, j) goes to slot (j, i)
            bL[j + (size_t)f * i] = a[m];
            bU[j + (size_t)f * i] = a[m];
        }
    }
}

// Small fronts of the LU variant: the whole front (both triangles) in LDS, one workgroup per (front, image).
// A.planes / planes2 = lower / upper diagonals, A.fac / fac2 = FL / FU, A.ws_mine / ws_mine2 and ws_child / ws_child2 =
// the update matrices' lower (with diagonal) / transposed strict upper triangles.  Gauss-Jordan inverse of the pivot
// block (p <= 48), L' = F21 D^-1 and F22 -= L' F12 on the f64 MFMA.  grid (nodes, nimg), block NDS_T, LDS nd_small_lds(MPmax).
__global__ __launch_bounds__(NDS_T) void nd_front_small_lu_kernel(NdArgs A) {
    extern __shared__ double S[];
    __shared__ int cmL[2][128];
    constexpr int NT = NDS_T, NW = NT / 64;
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b, p16 = nd_up16(p), sh = p16 - p;
    const int MP = nd_up16(p16 + b), ld = MP + 1, Pp = p16 >> 4, P = MP >> 4;
    double* d0 = S + (size_t)ld * MP;   // [MP]
    NdNodeDev ch[2];
    int bc[2] = {0, 0};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (cn >= 0) { ch[ci] = A.nodes[cn]; bc[ci] = ch[ci].b; }
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
        for (int kk = tid; kk < bc[ci]; kk += NT) {
            const int R = A.cmap[ch[ci].cmap_off + kk];
            cmL[ci][kk] = R < p ? R : R + sh;
        }
    for (int e = tid; e < ld * MP + MP; e += NT) S[e] = 0.0;
    __syncthreads();
    for (int kk = p + tid; kk < p16; kk += NT) S[kk + ld * kk] = 1.0;   // identity padding of the pivot block
    {
        const double* plL = A.planes + (size_t)img * A.n;
        const double* plU = A.planes2 + (size_t)img * A.n;
        for (int e = tid; e < v.orig_cnt; e += NT) {
            const int4 o = A.orig[v.orig_off + e];
            const int r = o.x < p ? o.x : o.x + sh, c = o.y;   // c is a pivot
            const size_t at = (size_t)(o.z & 15) * A.tot + o.w;
            const bool up = (o.z & 16) != 0;                   // F(r, c) is an entry of A's upper triangle
            S[r + ld * c] += (up ? plU : plL)[at];
            if (r != c) S[c + ld * r] += (up ? plL : plU)[at];
        }
    }
    __syncthreads();
    for (int ci = 0; ci < 2; ++ci) {
        if (bc[ci] == 0) continue;
        const double* UL = A.ws_child + (size_t)img * A.ws_child_stride + ch[ci].u_off;
        const double* UU = A.ws_child2 + (size_t)img * A.ws_child_stride + ch[ci].u_off;
        const int n = bc[ci];
        const int* cm = cmL[ci];
        for (int j = wave; j < n; j += NW)
            for (int i = j + lane; i < n; i += 64) {
                const double xl = UL[i + (size_t)n * j];
                const double xu = (i > j) ? UU[i + (size_t)n * j] : 0.0;
                S[cm[i] + ld * cm[j]] += xl;
                if (i > j) S[cm[j] + ld * cm[i]] += xu;
            }
        __syncthreads();
    }
    // ---- D^-1 in place (first p steps; the padding is an identity block)
    for (int kk = tid; kk < p16; kk += NT) d0[kk] = fabs(S[kk + ld * kk]);
    __syncthreads();
    {
        const int ne = p16 * p16;
        int badk = -1;
        for (int kk = 0; kk < p; ++kk) {
            const double piv = S[kk + ld * kk];
            if (badk < 0 && nd_lu_pivot_bad(piv, d0[kk])) badk = kk;
            const double pinv = 1.0 / piv;
            double nv[9];   // 48 * 48 / 256
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                const int e = tid + NT * m;
                if (e < ne) {
                    const int i = e % p16, j = e / p16;
                    const double rk = (j == kk) ? pinv : S[kk + ld * j] * pinv;
                    const double fi = S[i + ld * kk];
                    const double old = (j == kk) ? 0.0 : S[i + ld * j];
                    nv[m] = (i == kk) ? rk : old - fi * rk;
                }
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                const int e = tid + NT * m;
                if (e < ne) S[(e % p16) + ld * (e / p16)] = nv[m];
            }
            __syncthreads();
        }
        if (badk >= 0 && tid == 0 && A.fail[img] == 0) A.fail[img] = node + 1;
    }
    // ---- L' = F21 D^-1: a wave per boundary tile row, all Pp (<= 3) column tiles before it writes
    for (int i = Pp + wave; i < P; i += NW) {
        bcr_d4 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = bcr_d4{0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < Pp; ++q) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double a = S[(16 * i + lr) + ld * (16 * q + 4 * kk + lk)];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < Pp) acc[c] = bcr_mfma(a, S[(16 * q + 4 * kk + lk) + ld * (16 * c + lr)], acc[c]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < Pp) {
#pragma unroll
                for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * c + lr)] = acc[c][g];
            }
    }
    __syncthreads();
    // ---- F22 -= L' F12, every tile of the boundary square
    {
        const int m = P - Pp;
        for (int t = wave; t < m * m; t += NW) {
            const int i = Pp + t % m, j = Pp + t / m;
            bcr_d4 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = S[(16 * i + lk + 4 * g) + ld * (16 * j + lr)];
            for (int q = 0; q < Pp; ++q) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double a = S[(16 * i + lr) + ld * (16 * q + 4 * kk + lk)];
                    const double bb = S[(16 * q + 4 * kk + lk) + ld * (16 * j + lr)];
                    acc = bcr_mfma(-a, bb, acc);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) S[(16 * i + lk + 4 * g) + ld * (16 * j + lr)] = acc[g];
        }
    }
    __syncthreads();
    // ---- results: FL rows [p, f) = L'; FU rows [0, p) = D^-T, rows [p, f) = F12^T; update matrix in two triangles
    double* fL = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    double* fU = A.fac2 + (size_t)img * A.fac_stride + v.fac_off;
    for (int c = wave; c < p; c += NW)
        for (int r = lane; r < f; r += 64) {
            if (r < p) {
                const double x = S[c + ld * r];
                fL[r + (size_t)f * c] = x;
                fU[r + (size_t)f * c] = x;
            } else {
                fL[r + (size_t)f * c] = S[(r + sh) + ld * c];
                fU[r + (size_t)f * c] = S[c + ld * (r + sh)];
            }
        }
    double* UL = A.ws_mine + (size_t)img * A.ws_mine_stride + v.u_off;
    double* UU = A.ws_mine2 + (size_t)img * A.ws_mine_stride + v.u_off;
    for (int j = wave; j < b; j += NW)
        for (int i = j + lane; i < b; i += 64) {
            UL[i + (size_t)b * j] = S[(p16 + i) + ld * (p16 + j)];
            UU[i + (size_t)b * j] = (i > j) ? S[(p16 + j) + ld * (p16 + i)] : 0.0;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// substitutions
// ------------------------------------------------------------------------------------------------------------------
struct NdSolveArgs {
    const NdNodeDev* nodes;
    const int* pix;
    const int* cmap;
    const double* fac;
    const double* fac2;     // LU variant: the factor columns of the upper triangle (backward substitution)
    int lu;                 // 0: Cholesky.  1: block LU with inverted diagonal blocks -- forward: y_k = w_k (no triangular
                            // factor to apply), rows below lose L' y_k; backward: x_k = D_k^-1 (y_k - U_k,after x_after),
                            // D_k^-1 stored transposed in the diagonal slot of fac2
    long long fac_stride;
    double* vec;            // [nimg][n]: right-hand side in, solution out
    double* y;              // [nimg][n]: forward result
    double* uv;             // [nimg][uv_stride]: update vectors
    long long uv_stride;
    double* acc;            // += solution when not null
    int n, node0;
};

// acc += sgn * sum_{c < n} col[ld * c] * v[c], added in column order with eight loads in flight: the loads do not depend
// on the running sum, and a plain loop left one global round trip per column on the critical path of the wave that owns a
// front (the substitutions wait 83-95 % of their cycles for memory).  Same summation order as the plain loop: same bits.
__device__ __forceinline__ double nd_dot_cols(const double* __restrict__ col, size_t ld, const double* __restrict__ v, int n, double acc, double sgn) {
    int c = 0;
    for (; c + 8 <= n; c += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = col[ld * (size_t)(c + u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_fma(sgn * a[u], v[c + u], acc);
    }
    for (; c < n; ++c) acc = __builtin_fma(sgn * col[ld * (size_t)c], v[c], acc);
    return acc;
}

// Small fronts, one wave per (front, image): forward.  grid (nodes, nimg), block 64.  f <= 128.
__global__ __launch_bounds__(64) void nd_fwd_small_kernel(NdSolveArgs A) {
    __shared__ double w[128], yv[128];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    const int* px = A.pix + v.piv_off;
    double* uvi = A.uv + (size_t)img * A.uv_stride;
    for (int e = lane; e < f; e += 64) w[e] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (cn < 0) continue;
        const NdNodeDev ch = A.nodes[cn];
        const int* cm = A.cmap + ch.cmap_off;
        const double* uc = uvi + ch.uv_off;
        for (int e = lane; e < ch.b; e += 64) w[cm[e]] += uc[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const double* rhs = A.vec + (size_t)img * A.n;
    for (int e = lane; e < p; e += 64) w[e] = rhs[px[e]] - w[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    double* yo = A.y + (size_t)img * A.n;
    for (int r = lane; r < p; r += 64) {          // y = W w_p (W lower triangular, zeros stored above); LU: y = w_p
        const double acc = A.lu ? w[r] : nd_dot_cols(fc + r, (size_t)f, w, r + 1, 0.0, 1.0);
        yv[r] = acc;
        yo[px[r]] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < b; i += 64)            // update vector: children's sums + L21 y
        uvi[v.uv_off + i] = nd_dot_cols(fc + (p + i), (size_t)f, yv, p, w[p + i], 1.0);
}

// backward: x_p = W^T (y_p - L21^T x_b).  grid (nodes, nimg), block 64.
__global__ __launch_bounds__(64) void nd_bwd_small_kernel(NdSolveArgs A) {
    __shared__ double xb[128], z[128];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    const int* px = A.pix + v.piv_off;
    double* x = A.vec + (size_t)img * A.n;
    const double* yo = A.y + (size_t)img * A.n;
    for (int i = lane; i < b; i += 64) xb[i] = x[px[p + i]];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* fc = (A.lu ? A.fac2 : A.fac) + (size_t)img * A.fac_stride + v.fac_off;
    for (int c = lane; c < p; c += 64)            // unit stride down column c: eight loads in flight
        z[c] = nd_dot_cols(fc + p + (size_t)f * c, 1, xb, b, yo[px[c]], -1.0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double* ac = A.acc ? A.acc + (size_t)img * A.n : nullptr;
    for (int c = lane; c < p; c += 64) {
        const int r0 = A.lu ? 0 : c;
        const double acc = nd_dot_cols(fc + (size_t)f * c + r0, 1, z + r0, p - r0, 0.0, 1.0);
        x[px[c]] = acc;
        if (ac) ac[px[c]] += acc;
    }
}

// The same two substitutions with the front's factor block (f x p doubles, contiguous) STAGED in LDS: the block is fetched
// by full-width consecutive loads, sixteen per lane in flight and none of them behind a use -- the column loops above touch
// a front through 2 p partial-line loads of <= f lanes (forward) or walk p columns with one lane each (backward), eight in
// flight, which is why they move the factor at 2.3 TB/s -- and every dot product then reads LDS.  Same operations in the
// same order as nd_fwd_small_kernel / nd_bwd_small_kernel: the same bits.  For levels whose fronts have f p <= NDS_STAGE
// entries (the bottom five levels of a 1024^2 image); dynamic LDS: 8 * (largest f p of the level) bytes.
constexpr int NDS_STAGE = 2048;
constexpr int NDS_SU = 16;
// first the requests of the block's first 64 * NDS_SU entries (issued by the caller ahead of its dependent loads), then
// their commit to LDS and the rest of the block
__device__ __forceinline__ void nd_stage_request(const double* __restrict__ src, int n, int lane, double (&t)[NDS_SU]) {
#pragma unroll
    for (int u = 0; u < NDS_SU; ++u) t[u] = src[min(64 * u + lane, n - 1)];
}
__device__ __forceinline__ void nd_stage_commit(const double* __restrict__ src, double* __restrict__ S, int n, int lane, double (&t)[NDS_SU]) {
    for (int e0 = 0; e0 < n; e0 += 64 * NDS_SU) {
        if (e0 > 0) {
#pragma unroll
            for (int u = 0; u < NDS_SU; ++u) t[u] = src[min(e0 + 64 * u + lane, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < NDS_SU; ++u) {
            const int e = e0 + 64 * u + lane;
            if (e < n) S[e] = t[u];
        }
    }
}

__global__ __launch_bounds__(64) void nd_fwd_staged_kernel(NdSolveArgs A) {
    extern __shared__ double S[];
    __shared__ double w[128], yv[128];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    const int* px = A.pix + v.piv_off;
    double* uvi = A.uv + (size_t)img * A.uv_stride;
    const double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    const double* rhs = A.vec + (size_t)img * A.n;
    // requests in the order of their dependences: pivot pixel index, the block (needs the descriptor only), then what needs
    // the index and the children's descriptors -- right-hand side, update vectors and maps
    const int pxv = lane < p ? px[lane] : 0;
    double t[NDS_SU];
    const bool need = !A.lu || b > 0;
    if (need) nd_stage_request(fc, f * p, lane, t);
    __builtin_amdgcn_sched_barrier(0);
    double r0 = 0.0;
    if (lane < p) r0 = rhs[pxv];
    int cme[2][2] = {{0, 0}, {0, 0}};
    double uce[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    int cb[2] = {0, 0};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (cn < 0) continue;
        const NdNodeDev ch = A.nodes[cn];
        cb[ci] = ch.b;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = lane + 64 * h;
            if (e < ch.b) { cme[ci][h] = A.cmap[ch.cmap_off + e]; uce[ci][h] = uvi[ch.uv_off + e]; }
        }
    }
    if (need) nd_stage_commit(fc, S, f * p, lane, t);
    for (int e = lane; e < f; e += 64) w[e] = 0.0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {                // child 0, then child 1 (a child's entries hit distinct rows)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (lane + 64 * h < cb[ci]) w[cme[ci][h]] += uce[ci][h];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane < p) w[lane] = r0 - w[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double* yo = A.y + (size_t)img * A.n;
    if (lane < p) {                                 // y = W w_p (W lower triangular); LU: y = w_p
        double acc = 0.0;
        if (A.lu) acc = w[lane];
        else
            for (int c = 0; c <= lane; ++c) acc = __builtin_fma(S[lane + f * c], w[c], acc);
        yv[lane] = acc;
        yo[pxv] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < b; i += 64) {            // update vector: children's sums + L21 y
        double acc = w[p + i];
        for (int c = 0; c < p; ++c) acc = __builtin_fma(S[(p + i) + f * c], yv[c], acc);
        uvi[v.uv_off + i] = acc;
    }
}

__global__ __launch_bounds__(64) void nd_bwd_staged_kernel(NdSolveArgs A) {
    extern __shared__ double S[];
    __shared__ double xb[128], z[128];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    const int* px = A.pix + v.piv_off;
    double* x = A.vec + (size_t)img * A.n;
    const double* yo = A.y + (size_t)img * A.n;
    const double* fc = (A.lu ? A.fac2 : A.fac) + (size_t)img * A.fac_stride + v.fac_off;
    int pxb[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < b) pxb[h] = px[p + lane + 64 * h];
    const int pxv = lane < p ? px[lane] : 0;
    double t[NDS_SU];
    nd_stage_request(fc, f * p, lane, t);          // the block needs the descriptor only: ahead of the loads that need the indices
    __builtin_amdgcn_sched_barrier(0);
    double xr[2] = {0.0, 0.0}, y0 = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < b) xr[h] = x[pxb[h]];
    if (lane < p) y0 = yo[pxv];
    nd_stage_commit(fc, S, f * p, lane, t);
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < b) xb[lane + 64 * h] = xr[h];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < p) {                                 // z = y_p - L21^T x_b, down column `lane`
        double acc = y0;
        const double* col = S + p + f * lane;
        for (int i = 0; i < b; ++i) acc = __builtin_fma(-col[i], xb[i], acc);
        z[lane] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < p) {                                 // x_p = W^T z (LU: the whole column of D^-T)
        const int r0 = A.lu ? 0 : lane;
        double acc = 0.0;
        const double* col = S + f * lane;
        for (int r = r0; r < p; ++r) acc = __builtin_fma(col[r], z[r], acc);
        x[pxv] = acc;
        if (A.acc) A.acc[(size_t)img * A.n + pxv] += acc;
    }
}

// Large fronts, one workgroup per (front, image).  Dynamic LDS: nd_large_lds(largest front of the batch).
constexpr int NDL_T = 1024;
constexpr int NDL_RB = 4;     // row blocks (forward) / columns per wave (backward) whose loads are in flight together
inline size_t nd_large_lds(int fmax) { return sizeof(double) * ((size_t)fmax + (1 + 8 * NDL_RB) * HB2_NB + 64); }

// forward: wf = [rhs_p - s_p ; -s_b]; per pivot panel: y_k = W_kk wf_k, then wf[r] -= L(r, panel k) y_k for every row
// below; update vector = -wf_b.  grid (nodes, nimg), block NDL_T.
// split != 0 (fronts with many boundary rows): only the pivot rows are updated here and the update vector leaves as
// the children's sums; nd_fwd_rows_kernel adds L21 y afterwards, 128 boundary rows per workgroup -- the b x p block is
// most of a large front, and one workgroup per front streams it at one CU's bandwidth.
__global__ __launch_bounds__(NDL_T) void nd_fwd_large_kernel(NdSolveArgs A, int split) {
    extern __shared__ double sm[];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    double* wf = sm;                // [f]
    double* part = sm + f;          // [8][128] partial sums
    const int* px = A.pix + v.piv_off;
    double* uvi = A.uv + (size_t)img * A.uv_stride;
    for (int e = tid; e < f; e += NDL_T) wf[e] = 0.0;
    __syncthreads();
    for (int ci = 0; ci < 2; ++ci) {
        const int cn = ci ? v.child1 : v.child0;
        if (cn < 0) continue;
        const NdNodeDev ch = A.nodes[cn];
        const int* cm = A.cmap + ch.cmap_off;
        const double* uc = uvi + ch.uv_off;
        for (int e = tid; e < ch.b; e += NDL_T) wf[cm[e]] -= uc[e];
        __syncthreads();
    }
    const double* rhs = A.vec + (size_t)img * A.n;
    for (int e = tid; e < p; e += NDL_T) wf[e] += rhs[px[e]];
    __syncthreads();
    const double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    double* yo = A.y + (size_t)img * A.n;
    const int r128 = tid & 127, g8 = tid >> 7;
    for (int c0 = 0; c0 < p; c0 += HB2_NB) {
        const int nb = min(HB2_NB, p - c0);
        // y_k = W_kk wf_k: thread (row r128, column group g8)
        if (A.lu) {   // block LU: y_k = wf_k
            if (tid < nb) yo[px[c0 + tid]] = wf[c0 + tid];
        } else {
            double acc = 0.0;
            if (r128 < nb) {
                const double* Wr = fc + (c0 + r128) + (size_t)f * c0;
                for (int c = g8; c <= r128; c += 8) acc = __builtin_fma(Wr[(size_t)f * c], wf[c0 + c], acc);
            }
            part[g8 * HB2_NB + r128] = acc;
            __syncthreads();
            if (tid < nb) {
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += part[g * HB2_NB + tid];
                wf[c0 + tid] = s;
                yo[px[c0 + tid]] = s;
            }
            __syncthreads();
        }
        // rows below the diagonal block: wf[r] -= sum_c L(r, c0 + c) y[c]; 128 rows x 8 column groups per row block, NDL_RB
        // row blocks per pass: their loads (16 per thread and block) are in flight together -- one workgroup streams a
        // front of up to 12 MB, and what it moves per second is what it has in flight.  Same sums in the same order.
        const int rend = split ? p : f;
        for (int R0 = c0 + nb; R0 < rend; R0 += NDL_RB * HB2_NB) {
            double acc[NDL_RB];
#pragma unroll
            for (int q = 0; q < NDL_RB; ++q) acc[q] = 0.0;
            for (int c = g8; c < nb; c += 8) {
                const double yc = wf[c0 + c];
#pragma unroll
                for (int q = 0; q < NDL_RB; ++q) {
                    const int r = R0 + q * HB2_NB + r128;
                    const double l = fc[min(r, rend - 1) + (size_t)f * (c0 + c)];
                    acc[q] = __builtin_fma(r < rend ? l : 0.0, yc, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < NDL_RB; ++q) part[(q * 8 + g8) * HB2_NB + r128] = acc[q];
            __syncthreads();
            if (tid < NDL_RB * HB2_NB && R0 + tid < rend) {
                const int q = tid >> 7, t = tid & 127;
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += part[(q * 8 + g) * HB2_NB + t];
                wf[R0 + tid] -= s;
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < b; i += NDL_T) uvi[v.uv_off + i] = -wf[p + i];
}

// update vector += L21 y for 128 boundary rows of a front.  grid (ceil(bmax / 128), nodes, nimg), block NDL_T, dynamic
// LDS (pmax + 8 * 128) doubles.
__global__ __launch_bounds__(NDL_T) void nd_fwd_rows_kernel(NdSolveArgs A) {
    extern __shared__ double sm[];
    const int node = A.node0 + blockIdx.y, img = blockIdx.z, tid = threadIdx.x;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b, i0 = HB2_NB * (int)blockIdx.x;
    if (i0 >= b) return;
    double* yp = sm;            // [p]
    double* part = sm + p;      // [8][128]
    const int* px = A.pix + v.piv_off;
    const double* yo = A.y + (size_t)img * A.n;
    for (int c = tid; c < p; c += NDL_T) yp[c] = yo[px[c]];
    __syncthreads();
    const double* fc = A.fac + (size_t)img * A.fac_stride + v.fac_off;
    const int r128 = tid & 127, g8 = tid >> 7, i = i0 + r128;
    double acc = 0.0;
    if (i < b) {
        const double* Lr = fc + (p + i);
        for (int c = g8; c < p; c += 8) acc = __builtin_fma(Lr[(size_t)f * c], yp[c], acc);
    }
    part[g8 * HB2_NB + r128] = acc;
    __syncthreads();
    if (tid < HB2_NB && i0 + tid < b) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += part[g * HB2_NB + tid];
        double* uvi = A.uv + (size_t)img * A.uv_stride + v.uv_off;
        uvi[i0 + tid] += s;
    }
}

// y_c -= sum_i L21(i, c) x_b(i) for the 128 columns of one pivot panel (a wave per column, lanes along the rows): the
// boundary part of the backward step, ahead of nd_bwd_large_kernel(split).  grid (ceil(pmax / 128), nodes, nimg),
// block NDL_T, dynamic LDS bmax doubles.
__global__ __launch_bounds__(NDL_T) void nd_bwd_cols_kernel(NdSolveArgs A) {
    extern __shared__ double sm[];
    const int node = A.node0 + blockIdx.y, img = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b, c0 = HB2_NB * (int)blockIdx.x;
    if (c0 >= p) return;
    const int* px = A.pix + v.piv_off;
    const double* x = A.vec + (size_t)img * A.n;
    double* yo = A.y + (size_t)img * A.n;
    for (int i = tid; i < b; i += NDL_T) sm[i] = x[px[p + i]];
    __syncthreads();
    const double* fc = (A.lu ? A.fac2 : A.fac) + (size_t)img * A.fac_stride + v.fac_off;
    const int nb = min(HB2_NB, p - c0);
    for (int c = wave; c < nb; c += NDL_T / 64) {
        const double* col = fc + p + (size_t)f * (c0 + c);
        double acc = 0.0;
        for (int i = lane; i < b; i += 64) acc = __builtin_fma(col[i], sm[i], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) yo[px[c0 + c]] -= acc;
    }
}

// backward: vf = [x_p ; x_b]; panels from the last to the first: t_c = y_c - sum_{r below the block} L(r, c) vf[r]
// (a wave per column, lanes along the rows), x_k = W_kk^T t.  grid (nodes, nimg), block NDL_T.
// split != 0: the boundary rows have been applied by nd_bwd_cols_kernel; only the pivot rows are swept here.
__global__ __launch_bounds__(NDL_T) void nd_bwd_large_kernel(NdSolveArgs A, int split) {
    extern __shared__ double sm[];
    const int node = A.node0 + blockIdx.x, img = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const NdNodeDev v = A.nodes[node];
    const int p = v.p, b = v.b, f = p + b;
    double* vf = sm;                // [f]
    double* tt = sm + f;            // [128]
    double* part = sm + f + HB2_NB; // [8][128]
    const int* px = A.pix + v.piv_off;
    double* x = A.vec + (size_t)img * A.n;
    const double* yo = A.y + (size_t)img * A.n;
    if (!split)
        for (int i = tid; i < b; i += NDL_T) vf[p + i] = x[px[p + i]];
    __syncthreads();
    const double* fc = (A.lu ? A.fac2 : A.fac) + (size_t)img * A.fac_stride + v.fac_off;
    double* ac = A.acc ? A.acc + (size_t)img * A.n : nullptr;
    const int npan = (p + HB2_NB - 1) / HB2_NB;
    const int rend = split ? p : f;
    for (int k = npan - 1; k >= 0; --k) {
        const int c0 = HB2_NB * k, nb = min(HB2_NB, p - c0), rlo = c0 + nb;
        for (int cb = wave * NDL_RB; cb < nb; cb += (NDL_T / 64) * NDL_RB) {   // NDL_RB columns per wave at a time: their loads in flight together
            double acc[NDL_RB];
#pragma unroll
            for (int q = 0; q < NDL_RB; ++q) acc[q] = 0.0;
            for (int r = rlo + lane; r < rend; r += 64) {
                const double vr = vf[r];
#pragma unroll
                for (int q = 0; q < NDL_RB; ++q) {
                    const double l = fc[r + (size_t)f * (c0 + min(cb + q, nb - 1))];
                    acc[q] = __builtin_fma(l, vr, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < NDL_RB; ++q) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off, 64);
                if (lane == 0 && cb + q < nb) tt[cb + q] = yo[px[c0 + cb + q]] - acc[q];
            }
        }
        __syncthreads();
        // x_k = W_kk^T t: x[c] = sum_{r >= c} W(r, c) t[r]; thread (column c = tid & 127, row group g8)
        {
            const int c = tid & 127, g8 = tid >> 7;
            double acc = 0.0;
            if (c < nb) {
                const double* Wc = fc + c0 + (size_t)f * (c0 + c);
                for (int r = (A.lu ? 0 : c) + g8; r < nb; r += 8) acc = __builtin_fma(Wc[r], tt[r], acc);
            }
            part[g8 * HB2_NB + c] = acc;
        }
        __syncthreads();
        if (tid < nb) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += part[g * HB2_NB + tid];
            vf[c0 + tid] = s;
            x[px[c0 + tid]] = s;
            if (ac) ac[px[c0 + tid]] += s;
        }
        __syncthreads();
    }
}

}  // namespace bpltv
