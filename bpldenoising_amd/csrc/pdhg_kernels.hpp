// pdhg_kernels.hpp -- CDNA4 (gfx950) kernels of the accelerated Chambolle-Pock / PDHG iteration
// for ROF denoising: forward-difference gradient, backward-difference divergence, dual l2-ball
// projection and primal data-fidelity prox, fused, with temporal blocking.
//
// Replaces the external `op_denoise_pdps` loop called at
// /root/reference/src/TVLearningFunctionVec.jl:52,67 (constants :33-43) and
// /root/reference/src/BPLDenoising.jl:56,79.  Arithmetic = "spec v2" of oracle/bpltv_oracle.c,
// reproduced bit for bit (explicit fma only; build with -ffp-contract=off).
//
// Design (MI355X): one workgroup owns one tile of one image.  It loads the tile plus a halo of
// `halo` pixels (x, y1, y2, f: registers), runs `nit <= halo` full PDHG iterations on chip --
// neighbour values travel through three LDS planes (y1, y2, xbar), two barriers per iteration --
// and writes back only the core of the tile.  One launch therefore advances every image by `nit`
// iterations while HBM/L2 sees the state once: algorithmic traffic 56 B/px/iter (64 with an alpha
// map), actual traffic ~ (4 reads * (1+halo overhead) + 3 writes) * 8 B / nit.  State is
// ping-ponged between two buffer sets because neighbouring tiles read each other's halos.
// Pixel -> thread map: interleaved along i (li = ti + TI*pi) so that a wave's LDS and global accesses
// are unit-stride (conflict-free ds_read_b64, coalesced 512 B global rows); consecutive along j (lj = PJ*tj + pj), so
// that a wave owns a block of whole, adjacent rows -- the rows of the halo can then stop early (below).
#pragma once
#include <hip/hip_runtime.h>

#include "tiling.hpp"   // tile_span, tile_count (host + device; also compiled by g++ under the sanitizers)

namespace bpltv {

// LDS bytes of pdhg_tile_kernel for a region RI x RJ: y1 plane with a guard column, y2 plane with a
// guard row, xbar plane with RI+1 pad cells.
constexpr size_t pdhg_lds_bytes(int RI, int RJ, size_t word = sizeof(double)) {
    return word * ((size_t)RJ * (RI + 1) + (size_t)(RJ + 1) * RI + (size_t)RJ * RI + RI + 1);
}

constexpr int TAB_STRIDE = 8;  // doubles per iteration row: tau, sigma, omega, 1/(1+tau), 1+omega, pad

struct PdhgArgs {
    const double* xin;
    const double* y1in;
    const double* y2in;
    double* xout;
    double* y1out;
    double* y2out;
    const double* f;
    const double* alpha;  // device, am*an doubles (column major)
    const double* tab;    // device, [maxiter][TAB_STRIDE]
    double rho;
    int am, an;
    int it0, nit;
    int M, N, O;
    int nTi, nTj, halo;
    int first;  // 1: start from x = f, y = 0 (inputs xin/y1in/y2in ignored)
    int img0;   // first image handled by this launch (grid = tiles per image * images of the chain)
    int Odata;  // images in the dataset; image `img` of the solve uses f[img % Odata] and the
                // parameter block alpha + (img / Odata) * astride (parameter sweeps: K*Odata problems)
    int astride;
    int ntiles; // tiles of this launch (pdhg_wave_kernel: several tiles per workgroup, the last one may be short)
    int grid3d; // 1: the grid is (nTi, nTj, images): tile and image come from blockIdx.x / .y / .z and the kernel's prologue
                // needs no integer division (4 of them, ~100 scalar instructions per wave, with the 1-D grid)
    int seg;    // pdhg_stream_kernel: rows of a segment's region (core + the lead-in rows at its artificial ends)
    int xcd;    // 1 (1-D grid only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2; remap the
                // linear workgroup index so that every XCD works on a contiguous run of tiles (neighbouring tiles re-read
                // each other's halos: from the same L2 instead of from the memory side)
    unsigned* phase = nullptr;   // chain 0 of a two-chain solve: a word the first workgroup of every launch rewrites (first iteration of
                // the launch + 1, PDHG_PHASE_STAMP) -- pdhg_phase_gate_kernel in the other chain waits for it to change
#ifdef BPLTV_EXPERIMENTS
    int dbg;    // timing experiments of tools/ builds only (results are wrong): 1 skip state loads, 2 skip
                // stores, 4 no iterations, 16 nt stores, 32 plain stores, 64 nt loads, 128 no barriers (rows kernel).  The product
                // library is compiled without this field and without the branches it feeds.
#endif
};

#ifdef BPLTV_EXPERIMENTS
#define BPLTV_DBG(A) ((A).dbg)
// dbg & 1024 (tools/chain_phase.py): the first workgroup of every launch stamps the 100 MHz clock at its start into
// pdhg_tlog[chain][launch] -- how the launches of the two chains of a solve lie relative to each other
__device__ long long pdhg_tlog[2][4096];
#define PDHG_TLOG(A)                                                                                                   \
    if (((A).dbg & 1024) && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)                  \
        pdhg_tlog[(A).img0 != 0][min(4095, ((A).it0 + (A).halo - 1) / max((A).halo, 1))] = (long long)wall_clock64();
#else
#define BPLTV_DBG(A) 0
#define PDHG_TLOG(A)
#endif

// Tile (ta, tb) and image of the launch (imgl) of this workgroup; dataset image and parameter block of a solve image
// (parameter sweeps run K * Odata problems: image `img` uses f[img % Odata] and parameter block img / Odata).
// One plain store by one thread.  Anything here that waits for a value costs the whole launch: a launch lasts as long as its
// slowest workgroup, and an atomicAdd (or a clock read plus a load of the previous stamp) in the first workgroup's prologue
// made EVERY launch 25-30 % longer (5.22 -> 6.7 ms per 5000 iterations).
#define PDHG_PHASE_STAMP(A)                                                                              \
    if ((A).phase != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) \
        *(A).phase = (unsigned)(A).it0 + 1u;
#define PDHG_DECODE_BLOCK(A, imgl, ta, tb)                                   \
    PDHG_PHASE_STAMP(A)                                                      \
    PDHG_TLOG(A)                                                             \
    int imgl, ta, tb;                                                        \
    if ((A).grid3d) {                                                        \
        ta = (int)blockIdx.x; tb = (int)blockIdx.y; imgl = (int)blockIdx.z;  \
    } else {                                                                 \
        const int tilesPerImg_ = (A).nTi * (A).nTj;                          \
        int lin_ = (int)blockIdx.x;                                          \
        if ((A).xcd) {                                                       \
            const int q_ = (int)gridDim.x >> 3, r_ = (int)gridDim.x & 7;     \
            const int x_ = lin_ & 7, i_ = lin_ >> 3;                         \
            lin_ = x_ * q_ + (x_ < r_ ? x_ : r_) + i_;                       \
        }                                                                    \
        imgl = lin_ / tilesPerImg_;                                          \
        const int t_ = lin_ - imgl * tilesPerImg_;                           \
        ta = t_ % (A).nTi; tb = t_ / (A).nTi;                                \
    }
__device__ __forceinline__ void pdhg_data_image(int img, int O, int Odata, int& fimg, int& apar) {
    if (O == Odata) { fimg = img; apar = 0; }        // no sweep: the solve images are the dataset's
    else { fimg = img % Odata; apar = img / Odata; }
}

__device__ __forceinline__ double alpha_at(const double* __restrict__ alpha, int am, int an, int M,
                                           int N, int i, int j) {
    if (am == 1 && an == 1) return alpha[0];
    if (am == M && an == N) return alpha[i + (size_t)M * j];
    return alpha[(int)(((long)i * am) / M) + (size_t)am * (int)(((long)j * an) / N)];
}

// 1/sqrt(n2) by four Newton steps from an integer seed -- "spec v2", identical operation sequence
// to rsqrt_nr() in oracle/bpltv_oracle.c (bit-for-bit reproducible; 17 f64 ops instead of the ~31
// of an IEEE sqrt followed by an IEEE divide).
__device__ __forceinline__ double rsqrt_nr(double n2) {
    const unsigned long long u = 0x5FE6EB50C7B537A9ull - ((unsigned long long)__double_as_longlong(n2) >> 1);
    double r = __longlong_as_double((long long)u);
    const double h = 0.5 * n2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t = r * r;
        const double w = __builtin_fma(-h, t, 1.5);
        r = r * w;
    }
    return r;
}

// the same in single precision ("spec v2f"): three Newton steps from the classic 32-bit seed
__device__ __forceinline__ float rsqrt_nr(float n2) {
    const unsigned u = 0x5F375A86u - (__float_as_uint(n2) >> 1);
    float r = __uint_as_float(u);
    const float h = 0.5f * n2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t = r * r;
        const float w = __builtin_fmaf(-h, t, 1.5f);
        r = r * w;
    }
    return r;
}
// rsqrt_nr split into its seed and its Newton steps (pdhg_wave_kernel interleaves the steps of several pixels)
__device__ __forceinline__ void rsqrt_nr_seed(double n2, double& r, double& h) {
    const unsigned long long u = 0x5FE6EB50C7B537A9ull - ((unsigned long long)__double_as_longlong(n2) >> 1);
    r = __longlong_as_double((long long)u);
    h = 0.5 * n2;
}
__device__ __forceinline__ void rsqrt_nr_seed(float n2, float& r, float& h) {
    r = __uint_as_float(0x5F375A86u - (__float_as_uint(n2) >> 1));
    h = 0.5f * n2;
}
__host__ __device__ constexpr int rsqrt_nr_steps(double) { return 4; }
__host__ __device__ constexpr int rsqrt_nr_steps(float) { return 3; }
__device__ __forceinline__ double pd_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float pd_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// T = double: the reference's arithmetic.  T = float: the opt-in single-precision mode of bpltv_create(dtype = 32) --
// the same operation sequence in f32 ("spec v2f": bplo_pdhg_f32 of the oracle), with state, f, alpha and the step
// table held as float (the pointers of PdhgArgs then address float arrays): half the LDS and HBM bytes per pixel.
template <typename T, int PI, int PJ, int TI, int TJ>
__global__ __launch_bounds__(TI* TJ) void pdhg_tile_kernel(PdhgArgs A) {
    constexpr int RI = PI * TI, RJ = PJ * TJ;
    // LDS planes with guard cells so that the neighbour reads need no select:
    //   sy1: rows of RI+1, a leading zero column  -> y1(i-1,j) at li = 0 reads 0
    //   sy2: a leading zero row                   -> y2(i,j-1) at lj = 0 reads 0
    //   sxb: RI+1 trailing pad cells              -> xbar(i+1,j), xbar(i,j+1) at the region edge read
    //        finite garbage (zeros); there the result is either masked by the image-border test or
    //        lies in the halo that is never written back
    constexpr int S1 = RI + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char pdhg_smem[];
    T* smem = reinterpret_cast<T*>(pdhg_smem);
    T* sy1 = smem;                              // [RJ][S1]
    T* sy2 = smem + RJ * S1;                    // [RJ+1][RI]
    T* sxb = smem + RJ * S1 + (RJ + 1) * RI;    // [RJ][RI] + RI + 1
    const T* __restrict__ Axin = reinterpret_cast<const T*>(A.xin);
    const T* __restrict__ Ay1in = reinterpret_cast<const T*>(A.y1in);
    const T* __restrict__ Ay2in = reinterpret_cast<const T*>(A.y2in);
    T* __restrict__ Axout = reinterpret_cast<T*>(A.xout);
    T* __restrict__ Ay1out = reinterpret_cast<T*>(A.y1out);
    T* __restrict__ Ay2out = reinterpret_cast<T*>(A.y2out);
    const T* __restrict__ Af = reinterpret_cast<const T*>(A.f);

    const int tid = threadIdx.x;
    const int ti = tid % TI, tj = tid / TI;
    PDHG_DECODE_BLOCK(A, imgl, ta, tb)
    const int img = A.img0 + imgl;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    int fimg, apar;
    pdhg_data_image(img, A.O, A.Odata, fimg, apar);
    const size_t base = (size_t)img * M * N;                 // state planes: one slot per solve image
    const size_t fbase = (size_t)fimg * M * N;               // dataset planes
    const T* __restrict__ alpha = reinterpret_cast<const T*>(A.alpha) + (size_t)apar * A.astride;
    const int amode = (A.am == 1 && A.an == 1) ? 0 : ((A.am == M && A.an == N) ? 2 : 1);
    const bool first = (A.first != 0) || (BPLTV_DBG(A) & 1);

    // ---- prologue: every global load is issued before the first use (one memory round trip).
    // Out-of-image pixels read a clamped in-image address and are zeroed afterwards.
    T x[PJ][PI], y1[PJ][PI], y2[PJ][PI], f[PJ][PI], al[PJ][PI];
    size_t gidx[PJ][PI], fidx[PJ][PI], aidx[PJ][PI];
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            const int li = ti + TI * pi, lj = PJ * tj + pj;
            const int gi = min(oi + li, M - 1), gj = min(oj + lj, N - 1);
            gidx[pj][pi] = base + gi + (size_t)M * gj;
            fidx[pj][pi] = fbase + gi + (size_t)M * gj;
            size_t ai = 0;  // scalar alpha
            if (amode == 2) {
                ai = gi + (size_t)M * gj;
            } else if (amode == 1) {  // PatchOp: piecewise-constant upsampling
                const unsigned pa = ((unsigned)gi * (unsigned)A.am) / (unsigned)M;
                const unsigned pb = ((unsigned)gj * (unsigned)A.an) / (unsigned)N;
                ai = pa + (size_t)A.am * pb;
            }
            aidx[pj][pi] = ai;
        }
    if (!first) {
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                if (BPLTV_DBG(A) & 64) {  // experiment: non-temporal loads
                    x[pj][pi] = __builtin_nontemporal_load(&Axin[gidx[pj][pi]]);
                    y1[pj][pi] = __builtin_nontemporal_load(&Ay1in[gidx[pj][pi]]);
                    y2[pj][pi] = __builtin_nontemporal_load(&Ay2in[gidx[pj][pi]]);
                } else {
                    x[pj][pi] = Axin[gidx[pj][pi]];
                    y1[pj][pi] = Ay1in[gidx[pj][pi]];
                    y2[pj][pi] = Ay2in[gidx[pj][pi]];
                }
            }
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            f[pj][pi] = Af[fidx[pj][pi]];
            al[pj][pi] = alpha[aidx[pj][pi]];
        }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            const int li = ti + TI * pi, lj = PJ * tj + pj;
            const bool in = (oi + li < M) && (oj + lj < N);
            if (first) {
                x[pj][pi] = f[pj][pi];
                y1[pj][pi] = T(0);
                y2[pj][pi] = T(0);
            }
            if (!in) {
                f[pj][pi] = T(0); x[pj][pi] = T(0); y1[pj][pi] = T(0); y2[pj][pi] = T(0); al[pj][pi] = T(0);
            }
            sy1[lj * S1 + li + 1] = y1[pj][pi];
            sy2[(lj + 1) * RI + li] = y2[pj][pi];
        }
    for (int e = threadIdx.x; e < RJ; e += TI * TJ) sy1[e * S1] = T(0);
    for (int e = threadIdx.x; e < RI; e += TI * TJ) sy2[e] = T(0);
    for (int e = threadIdx.x; e < RI + 1; e += TI * TJ) sxb[RI * RJ + e] = T(0);
    __syncthreads();

    const T rho = (T)A.rho;
    const int nit = (BPLTV_DBG(A) & 4) ? 0 : A.nit;
    // Neumann border without a select in the loop: at the last image row/column (and outside the
    // image) the "neighbour" read is redirected to the pixel's own xbar cell, so the forward
    // difference is exactly +0.  The LDS offsets are computed once per launch.
    int n1[PJ][PI], n2[PJ][PI];
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            const int li = ti + TI * pi, lj = PJ * tj + pj;
            const int l = lj * RI + li;
            n1[pj][pi] = l + (((oi + li) < M - 1) ? 1 : 0);
            n2[pj][pi] = l + (((oj + lj) < N - 1) ? RI : 0);
        }
    // Halo rows do not need all the iterations.  The validity front moves in by one row per iteration: a row k rows
    // away from a region edge that is not an image border is only ever read for what it held after k iterations (k + 1
    // primal steps at the far edge, where the dual of the row above still reads its xbar); what it computes later is
    // garbage nobody uses.  A wave whose rows are all past their use stops computing and only keeps the barriers
    // company.  Enabled for the multi-pixel variants of the large images, where the loop is bound by f64 issue (the
    // freed slots go to the co-resident workgroup); with one pixel per thread -- the small batches, bound by the
    // latency of the computing waves' dependent chain -- it was measured to buy nothing.
    int my_nit = nit;
    if (PI * PJ > 1 && N > RJ) {
        const int w0 = (tid & ~63) / TI, w1 = min((tid | 63) / TI, TJ - 1);      // thread rows of this wave
        const int r0 = PJ * w0, r1 = PJ * w1 + PJ - 1;                          // pixel rows of this wave
        if (oj > 0) my_nit = min(my_nit, r1);                                   // near edge: row k needs k iterations
        if (oj + RJ < N) my_nit = min(my_nit, RJ - r0);                         // far edge: row k' rows from it needs k' + 1
    }
    // step sizes of iteration `it`: scalar loads, issued one iteration ahead
    const T* __restrict__ row = reinterpret_cast<const T*>(A.tab) + (size_t)TAB_STRIDE * A.it0;
    T tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    for (int it = 0; it < nit; ++it) {
        const T* __restrict__ nrow = row + TAB_STRIDE * ((it + 1 < nit) ? it + 1 : it);
        const T ntau = nrow[0], nsigma = nrow[1], nomega = nrow[2], ninv1ptau = nrow[3], nopw = nrow[4];
        const bool act = it < my_nit;
        T xb[PJ][PI];
        // ---- primal step: x <- prox_{tau*fidelity}(x - tau * G^T y); over-relaxation.
        // LDS reads are unconditional (clamped index) and selected afterwards: one wait for all.
        T y1m[PJ][PI], y2m[PJ][PI];
        if (act) {
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                const int li = ti + TI * pi, lj = PJ * tj + pj;
                y1m[pj][pi] = sy1[lj * S1 + li];        // (li-1)+1: guard column at li = 0
                y2m[pj][pi] = sy2[lj * RI + li];        // (lj-1)+1: guard row at lj = 0
            }
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                const int li = ti + TI * pi, lj = PJ * tj + pj;
                const int l = lj * RI + li;
                const T div = (y1m[pj][pi] - y1[pj][pi]) + (y2m[pj][pi] - y2[pj][pi]);
                const T tt = div - f[pj][pi];
                const T xo = x[pj][pi];
                const T xn = pd_fma(-tau, tt, xo) * inv1ptau;
                const T b = pd_fma(-omega, xo, opw * xn);
                x[pj][pi] = xn;
                xb[pj][pi] = b;
                sxb[l] = b;
            }
        }
        __syncthreads();
        if (act) {
        // ---- dual step: y <- proj_{|y_ij| <= alpha_ij}((y + sigma * G xbar) / (1 + sigma*rho/alpha))
        T xp1[PJ][PI], xpM[PJ][PI];
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                xp1[pj][pi] = sxb[n1[pj][pi]];
                xpM[pj][pi] = sxb[n2[pj][pi]];
            }
        T n2v[PJ][PI];
        bool any_out = false;
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                const T b = xb[pj][pi];
                const T d1 = xp1[pj][pi] - b;
                const T d2 = xpM[pj][pi] - b;
                const T a = al[pj][pi];
                T y1n = pd_fma(sigma, d1, y1[pj][pi]);
                T y2n = pd_fma(sigma, d2, y2[pj][pi]);
                if (rho != T(0)) {
                    const T den = T(1) + sigma * rho / a;
                    y1n = y1n / den;
                    y2n = y2n / den;
                }
                y1[pj][pi] = y1n;
                y2[pj][pi] = y2n;
                n2v[pj][pi] = pd_fma(y2n, y2n, y1n * y1n);
                any_out |= n2v[pj][pi] > a * a;
            }
        // projection onto the alpha-ball.  One pixel per thread: a wave whose pixels all lie inside
        // skips the rsqrt.  Several pixels per thread: the rsqrt chains of the thread's pixels are
        // issued together (independent, interleaved by the scheduler) and selected afterwards.
        if (any_out) {
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
                for (int pi = 0; pi < PI; ++pi) {
                    const T a = al[pj][pi];
                    const T v = a * rsqrt_nr(n2v[pj][pi]);
                    const bool outp = n2v[pj][pi] > a * a;
                    y1[pj][pi] = outp ? y1[pj][pi] * v : y1[pj][pi];
                    y2[pj][pi] = outp ? y2[pj][pi] * v : y2[pj][pi];
                }
        }
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                const int li = ti + TI * pi, lj = PJ * tj + pj;
                sy1[lj * S1 + li + 1] = y1[pj][pi];
                sy2[(lj + 1) * RI + li] = y2[pj][pi];
            }
        }
        tau = ntau; sigma = nsigma; omega = nomega; inv1ptau = ninv1ptau; opw = nopw;
        __syncthreads();
    }

#pragma unroll
    for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            const int li = ti + TI * pi, lj = PJ * tj + pj;
            const int gi = oi + li, gj = oj + lj;
            if (gi >= ci0 && gi < ci1 && gj >= cj0 && gj < cj1 && !(BPLTV_DBG(A) & 2)) {
                const size_t idx = base + gi + (size_t)M * gj;
                if (BPLTV_DBG(A) & 16) {  // experiment: non-temporal stores
                    __builtin_nontemporal_store(x[pj][pi], &Axout[idx]);
                    __builtin_nontemporal_store(y1[pj][pi], &Ay1out[idx]);
                    __builtin_nontemporal_store(y2[pj][pi], &Ay2out[idx]);
                } else if (!(BPLTV_DBG(A) & 32)) {
                    // write-through (sc1) stores: the state of this launch is read next by workgroups
                    // on other XCDs, so it has to reach memory anyway; writing through while the
                    // kernel still runs leaves no dirty L2 lines for the end-of-kernel release
                    // (-0.9 us per launch on MI355X, profiles/README.md)
                    __hip_atomic_store(&Axout[idx], x[pj][pi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&Ay1out[idx], y1[pj][pi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&Ay2out[idx], y2[pj][pi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    Axout[idx] = x[pj][pi];
                    Ay1out[idx] = y1[pj][pi];
                    Ay2out[idx] = y2[pj][pi];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------
// pdhg_wave_kernel: the same fused iteration with ONE WAVE PER TILE and the tile in registers -- no LDS, no barrier.
//
// Why: pdhg_tile_kernel keeps every wave of a workgroup in the same phase (two s_barrier per iteration), so on large
// images, where it is bound by f64 issue, only 0.62 of the issue slots do work.  Here a wave
// owns a 32 x (2 PJ) region by itself: lane l holds column li = l & 31 of the rows half * PJ .. half * PJ + PJ - 1
// (half = l >> 5), i.e. PJ pixels per lane, PJ independent dependency chains.  Neighbours along j are the lane's own
// registers (row pj +- 1; the two halves exchange one row per step), neighbours along i are the adjacent lanes.  Waves never wait for each other, so a SIMD's two resident waves overlap one's loads, shuffles and
// Newton chains with the other's arithmetic.  Arithmetic per pixel is the oracle's, operation for operation
// (bit-exact, tests/test_gpu_pdhg.py); the adjacent lanes' values come by DPP (wave_shr / wave_shl), the other half's row
// by one ds_bpermute per step.  The validity front shrinks by one pixel per iteration exactly as in the tile
// kernel (values at region edges that are not image borders are halo garbage and never written back).
// grid ceil(ntiles / WPB), block 64 * WPB: WPB independent waves per workgroup.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T pd_shfl(T v, int src) { return __shfl(v, src, 64); }
// value of the previous / next lane of the wave by DPP (v_mov_b32_dpp wave_shr:1 / wave_shl:1: VALU moves, no LDS
// crossbar and no wait; lane 0 resp. 63 keeps its own value).  A ds_bpermute per neighbour costs this kernel 2.3x.
__device__ __forceinline__ int pd_dpp_prev(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int pd_dpp_next(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ double pd_lane_prev(double v) {
    return __hiloint2double(pd_dpp_prev(__double2hiint(v)), pd_dpp_prev(__double2loint(v)));
}
__device__ __forceinline__ double pd_lane_next(double v) {
    return __hiloint2double(pd_dpp_next(__double2hiint(v)), pd_dpp_next(__double2loint(v)));
}
__device__ __forceinline__ float pd_lane_prev(float v) { return __int_as_float(pd_dpp_prev(__float_as_int(v))); }
__device__ __forceinline__ float pd_lane_next(float v) { return __int_as_float(pd_dpp_next(__float_as_int(v))); }

// lane - 1 with 0 for lane 0 (bound_ctrl: no previous value to keep, no move to set it up); lane + 1 with the lane's own
// value for lane 63 (bound_ctrl off: the destination keeps `v`)
__device__ __forceinline__ double pd_lane_prev_or0(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ double pd_lane_next_or_self(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xf, 0xf, false));
}
// the same shifts with the vacated lane (0 resp. 63) taking `fill` instead (pdhg_rowsw_kernel: the neighbour wave's edge)
__device__ __forceinline__ double pd_lane_prev_or(double v, double fill) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x138, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double pd_lane_next_or(double v, double fill) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x130, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float pd_lane_prev_or(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float pd_lane_next_or(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float pd_lane_prev_or0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float pd_lane_next_or_self(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// ------------------------------------------------------------------------------------------
// pdhg_rows_kernel: the fused iteration of pdhg_tile_kernel for LARGE images, laid out so that the halo costs less.
//
// A wave is one row of 64 threads along i; a thread owns PJ consecutive pixels along j (lj = PJ * tj + pj): region
// 64 x (PJ * TJ), TJ waves.  Neighbours along i are the adjacent lanes (DPP wave shifts, no LDS), neighbours along j
// inside a thread's strip are its own registers; only a strip's ends go through LDS (two planes of TJ + 1 rows: the y2
// of a strip's last pixel, the xbar of its first) -- 1 / PJ of pdhg_tile_kernel's LDS traffic, 2 KB instead of 55 KB,
// so occupancy is set by registers alone.  What that buys:
//   * 64 x 64 at T = 8 recomputes 1.78 x instead of the 2.25 x of the 48 x 48 region;
//   * halo pixels stop when nobody reads them any more (pdhg_tile_kernel does that per wave; here a wave's PJ pixel rows
//     are stopped row by row under wave-uniform branches): the first and last wave of a 64 x 64 region skip half their
//     pixel-iterations, 1.56 x recompute in all.
// Same arithmetic per pixel as pdhg_tile_kernel: bit-identical results.  Block 64 * TJ, dynamic LDS pdhg_rows_lds.
// For images of at least 64 x (PJ * TJ) pixels only (the host checks): then the image's last row / column is the region's
// last lane / pixel row (tile_span), where the Neumann difference xbar - xbar = +0 comes from the lane shift keeping the
// own value and from one select per strip -- no per-pixel border selects in the loop.
// ------------------------------------------------------------------------------------------
// CL: the per-pixel constants f and alpha live in LDS (each thread's own cells, read back every iteration) instead of
// registers -- 4 VGPRs per pixel less, which is what lets two 8-pixel workgroups share a CU.
constexpr size_t pdhg_rows_lds(int PJ, int TJ, bool CL, size_t word = sizeof(double)) {
    return word * 64 * (2 * (size_t)(TJ + 1) + (CL ? 2 * (size_t)PJ * TJ : 0));
}

template <typename T, int PJ, int TJ, bool CL>
__global__ __launch_bounds__(64 * TJ) void pdhg_rows_kernel(PdhgArgs A) {
    constexpr int RI = 64, RJ = PJ * TJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char pdhg_smem[];
    T* sy2 = reinterpret_cast<T*>(pdhg_smem);   // [TJ + 1][64]: row tj + 1 = y2 of strip tj's last pixel row, row 0 = 0
    T* sxb = sy2 + (TJ + 1) * 64;               // [TJ + 1][64]: row tj = xbar of strip tj's first pixel row, row TJ = 0
    T* sfc = sxb + (TJ + 1) * 64;               // CL: [RJ][64] f, then [RJ][64] alpha
    const T* __restrict__ Axin = reinterpret_cast<const T*>(A.xin);
    const T* __restrict__ Ay1in = reinterpret_cast<const T*>(A.y1in);
    const T* __restrict__ Ay2in = reinterpret_cast<const T*>(A.y2in);
    T* __restrict__ Axout = reinterpret_cast<T*>(A.xout);
    T* __restrict__ Ay1out = reinterpret_cast<T*>(A.y1out);
    T* __restrict__ Ay2out = reinterpret_cast<T*>(A.y2out);
    const T* __restrict__ Af = reinterpret_cast<const T*>(A.f);
    const int tid = threadIdx.x, ti = tid & 63;
    const int tj = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: uniform
    PDHG_DECODE_BLOCK(A, imgl, ta, tb)
#ifdef BPLTV_EXPERIMENTS
    if ((BPLTV_DBG(A) >> 8) > 0) {   // experiment: the initially resident workgroups start at pseudo-random phases
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin < 512u) {
            const unsigned hsh = (lin * 2654435761u) >> 16;
            const int units = (int)(hsh % (unsigned)(BPLTV_DBG(A) >> 8));   // 1024 cycles each
            for (int k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(16);
        }
    }
#endif
    const int img = A.img0 + imgl;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    int fimg, apar;
    pdhg_data_image(img, A.O, A.Odata, fimg, apar);
    const size_t base = (size_t)img * M * N;
    const size_t fbase = (size_t)fimg * M * N;
    const T* __restrict__ alpha = reinterpret_cast<const T*>(A.alpha) + (size_t)apar * A.astride;
    const int amode = (A.am == 1 && A.an == 1) ? 0 : ((A.am == M && A.an == N) ? 2 : 1);
    const bool first = (A.first != 0) || (BPLTV_DBG(A) & 1);
    const int lj0 = PJ * tj;
    const int gi = min(oi + ti, M - 1);
    const bool in_i = oi + ti < M;

    T x[PJ], y1[PJ], y2[PJ], f[PJ], al[PJ];
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = min(oj + lj0 + pj, N - 1);
        const size_t gidx = gi + (size_t)M * gj;
        size_t ai = 0;
        if (amode == 2) ai = gidx;
        else if (amode == 1) ai = ((unsigned)gi * (unsigned)A.am) / (unsigned)M + (size_t)A.am * (((unsigned)gj * (unsigned)A.an) / (unsigned)N);
        if (!first) {
            x[pj] = Axin[base + gidx];
            y1[pj] = Ay1in[base + gidx];
            y2[pj] = Ay2in[base + gidx];
        }
        f[pj] = Af[fbase + gidx];
        al[pj] = alpha[ai];
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        if (first) { x[pj] = f[pj]; y1[pj] = T(0); y2[pj] = T(0); }
        if (!(in_i && oj + lj0 + pj < N)) { f[pj] = T(0); x[pj] = T(0); y1[pj] = T(0); y2[pj] = T(0); al[pj] = T(0); }
        if (CL) {
            sfc[(lj0 + pj) * 64 + ti] = f[pj];
            sfc[(RJ + lj0 + pj) * 64 + ti] = al[pj];
        }
    }
    sy2[(tj + 1) * 64 + ti] = y2[PJ - 1];
    if (tj == 0) { sy2[ti] = T(0); sxb[TJ * 64 + ti] = T(0); }
    __syncthreads();

    const T rho = (T)A.rho;
    const int nit = A.nit;
    const bool hasD_last = oj + lj0 + PJ - 1 < N - 1;
    // iterations pixel row lj is still read for: lj of them next to a near edge that is not the image border, RJ - lj
    // next to such a far edge (see pdhg_tile_kernel)
    int lim[PJ];
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int lj = lj0 + pj;
        int l = nit;
        if (oj > 0) l = min(l, lj);
        if (oj + RJ < N) l = min(l, RJ - lj);
        lim[pj] = l;
    }
    const T* __restrict__ row = reinterpret_cast<const T*>(A.tab) + (size_t)TAB_STRIDE * A.it0;
    T tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    for (int it = 0; it < nit; ++it) {
        const T* __restrict__ nrow = row + TAB_STRIDE * ((it + 1 < nit) ? it + 1 : it);
        const T ntau = nrow[0], nsigma = nrow[1], nomega = nrow[2], ninv1ptau = nrow[3], nopw = nrow[4];
        T xb[PJ];
        // ---- primal step
        const T y2up = sy2[tj * 64 + ti];   // y2 of the pixel row above the strip (guard row of zeros at lj = 0)
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            xb[pj] = T(0);
            if (it < lim[pj]) {
                const T y1m = pd_lane_prev_or0(y1[pj]);                // lane 0: the zero guard column
                const T y2m = (pj > 0) ? y2[pj > 0 ? pj - 1 : 0] : y2up;
                const T div = (y1m - y1[pj]) + (y2m - y2[pj]);
                const T tt = div - (CL ? sfc[(lj0 + pj) * 64 + ti] : f[pj]);
                const T xo = x[pj];
                const T xn = pd_fma(-tau, tt, xo) * inv1ptau;
                xb[pj] = pd_fma(-omega, xo, opw * xn);
                x[pj] = xn;
            }
        }
        sxb[tj * 64 + ti] = xb[0];
        if (!(BPLTV_DBG(A) & 128)) __syncthreads();
        // ---- dual step
        const T xbdn = sxb[(tj + 1) * 64 + ti];   // xbar of the pixel row below the strip
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            if (it < lim[pj]) {
                const T b = xb[pj];
                const T xp1 = pd_lane_next_or_self(b);                 // lane 63: own value
                const T xpM = (pj < PJ - 1) ? xb[pj < PJ - 1 ? pj + 1 : pj] : (hasD_last ? xbdn : b);
                const T d1 = xp1 - b;
                const T d2 = xpM - b;
                const T a = CL ? sfc[(RJ + lj0 + pj) * 64 + ti] : al[pj];
                T y1n = pd_fma(sigma, d1, y1[pj]);
                T y2n = pd_fma(sigma, d2, y2[pj]);
                if (rho != T(0)) {
                    const T den = T(1) + sigma * rho / a;
                    y1n = y1n / den;
                    y2n = y2n / den;
                }
                const T n2v = pd_fma(y2n, y2n, y1n * y1n);
                if (n2v > a * a) {
                    const T v = a * rsqrt_nr(n2v);
                    y1n = y1n * v;
                    y2n = y2n * v;
                }
                y1[pj] = y1n;
                y2[pj] = y2n;
            }
        }
        sy2[(tj + 1) * 64 + ti] = y2[PJ - 1];
        tau = ntau; sigma = nsigma; omega = nomega; inv1ptau = ninv1ptau; opw = nopw;
        if (!(BPLTV_DBG(A) & 128)) __syncthreads();
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gi2 = oi + ti, gj2 = oj + lj0 + pj;
        if (gi2 >= ci0 && gi2 < ci1 && gj2 >= cj0 && gj2 < cj1 && !(BPLTV_DBG(A) & 2)) {
            const size_t idx = base + gi2 + (size_t)M * gj2;
            __hip_atomic_store(&Axout[idx], x[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay1out[idx], y1[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay2out[idx], y2[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// pdhg_rowsw_kernel: pdhg_rows_kernel with rows of 128 pixels -- two waves side by side (round 4).  Along i a 64-lane row
// with its halo of 8 keeps 48 of 64 lanes (1024 columns: 21 regions, 1344 lanes); a 128-pixel row keeps 112 of 128 (9
// regions, 1152 lanes): 14 % fewer pixel-iterations.  The two waves of a row exchange ONE column per step and direction
// through LDS -- y1 of the left wave's lane 63 for the right wave's lane 0 (primal step), xbar of the right wave's lane 0
// for the left wave's lane 63 (dual step) -- behind the two workgroup barriers the strips' row exchange needs anyway.
// Same arithmetic per pixel: bit-identical results.  Block 128 * TJ (wave = 2 * strip + half), dynamic LDS pdhg_rowsw_lds.
// ------------------------------------------------------------------------------------------
constexpr size_t pdhg_rowsw_lds(int PJ, int TJ, size_t word = sizeof(double)) {
    return word * (128 * 2 * (size_t)(TJ + 1) + 2 * (size_t)TJ * PJ);
}

template <typename T, int PJ, int TJ>
__global__ __launch_bounds__(128 * TJ) void pdhg_rowsw_kernel(PdhgArgs A) {
    constexpr int RI = 128, RJ = PJ * TJ;
    constexpr bool CL = false;
    extern __shared__ __attribute__((aligned(16))) unsigned char pdhg_smem[];
    T* sy2 = reinterpret_cast<T*>(pdhg_smem);   // [TJ + 1][128]: row tj + 1 = y2 of strip tj's last pixel row, row 0 = 0
    T* sxb = sy2 + (TJ + 1) * 128;              // [TJ + 1][128]: row tj = xbar of strip tj's first pixel row, row TJ = 0
    T* sE1 = sxb + (TJ + 1) * 128;              // [TJ][PJ]: y1 of lane 63 of the strip's left wave (for lane 0 of its right wave)
    T* sEx = sE1 + TJ * PJ;                     // [TJ][PJ]: xbar of lane 0 of the right wave (for lane 63 of the left wave)
    T* sfc = sEx;                               // (no f / alpha planes in LDS in this kernel)
    const T* __restrict__ Axin = reinterpret_cast<const T*>(A.xin);
    const T* __restrict__ Ay1in = reinterpret_cast<const T*>(A.y1in);
    const T* __restrict__ Ay2in = reinterpret_cast<const T*>(A.y2in);
    T* __restrict__ Axout = reinterpret_cast<T*>(A.xout);
    T* __restrict__ Ay1out = reinterpret_cast<T*>(A.y1out);
    T* __restrict__ Ay2out = reinterpret_cast<T*>(A.y2out);
    const T* __restrict__ Af = reinterpret_cast<const T*>(A.f);
    const int tid = threadIdx.x, l64 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: uniform
    const int hh = wv & 1, tj = wv >> 1;                        // left / right half of the 128-pixel row, strip
    const int ti = l64 + 64 * hh;
    PDHG_DECODE_BLOCK(A, imgl, ta, tb)
    const int img = A.img0 + imgl;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    int fimg, apar;
    pdhg_data_image(img, A.O, A.Odata, fimg, apar);
    const size_t base = (size_t)img * M * N;
    const size_t fbase = (size_t)fimg * M * N;
    const T* __restrict__ alpha = reinterpret_cast<const T*>(A.alpha) + (size_t)apar * A.astride;
    const int amode = (A.am == 1 && A.an == 1) ? 0 : ((A.am == M && A.an == N) ? 2 : 1);
    const bool first = (A.first != 0) || (BPLTV_DBG(A) & 1);
    const int lj0 = PJ * tj;
    const int gi = min(oi + ti, M - 1);
    const bool in_i = oi + ti < M;

    T x[PJ], y1[PJ], y2[PJ], f[PJ], al[PJ];
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = min(oj + lj0 + pj, N - 1);
        const size_t gidx = gi + (size_t)M * gj;
        size_t ai = 0;
        if (amode == 2) ai = gidx;
        else if (amode == 1) ai = ((unsigned)gi * (unsigned)A.am) / (unsigned)M + (size_t)A.am * (((unsigned)gj * (unsigned)A.an) / (unsigned)N);
        if (!first) {
            x[pj] = Axin[base + gidx];
            y1[pj] = Ay1in[base + gidx];
            y2[pj] = Ay2in[base + gidx];
        }
        f[pj] = Af[fbase + gidx];
        al[pj] = alpha[ai];
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        if (first) { x[pj] = f[pj]; y1[pj] = T(0); y2[pj] = T(0); }
        if (!(in_i && oj + lj0 + pj < N)) { f[pj] = T(0); x[pj] = T(0); y1[pj] = T(0); y2[pj] = T(0); al[pj] = T(0); }
        if (CL) {
            sfc[(lj0 + pj) * 128 + ti] = f[pj];
            sfc[(RJ + lj0 + pj) * 128 + ti] = al[pj];
        }
    }
    sy2[(tj + 1) * 128 + ti] = y2[PJ - 1];
    if (tj == 0) { sy2[ti] = T(0); sxb[TJ * 128 + ti] = T(0); }
    if (!hh && l64 == 63) {
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) sE1[tj * PJ + pj] = y1[pj];
    }
    __syncthreads();

    const T rho = (T)A.rho;
    const int nit = A.nit;
    const bool hasD_last = oj + lj0 + PJ - 1 < N - 1;
    // iterations pixel row lj is still read for: lj of them next to a near edge that is not the image border, RJ - lj
    // next to such a far edge (see pdhg_tile_kernel)
    int lim[PJ];
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int lj = lj0 + pj;
        int l = nit;
        if (oj > 0) l = min(l, lj);
        if (oj + RJ < N) l = min(l, RJ - lj);
        lim[pj] = l;
    }
    const T* __restrict__ row = reinterpret_cast<const T*>(A.tab) + (size_t)TAB_STRIDE * A.it0;
    T tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    for (int it = 0; it < nit; ++it) {
        const T* __restrict__ nrow = row + TAB_STRIDE * ((it + 1 < nit) ? it + 1 : it);
        const T ntau = nrow[0], nsigma = nrow[1], nomega = nrow[2], ninv1ptau = nrow[3], nopw = nrow[4];
        T xb[PJ];
        // ---- primal step
        const T y2up = sy2[tj * 128 + ti];   // y2 of the pixel row above the strip (guard row of zeros at lj = 0)
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            xb[pj] = T(0);
            if (it < lim[pj]) {
                // lane 0 of the left wave: the zero guard column; of the right wave: lane 63 of the left one (the shift's fill value)
                const T y1m = pd_lane_prev_or(y1[pj], hh ? sE1[tj * PJ + pj] : T(0));
                const T y2m = (pj > 0) ? y2[pj > 0 ? pj - 1 : 0] : y2up;
                const T div = (y1m - y1[pj]) + (y2m - y2[pj]);
                const T tt = div - (CL ? sfc[(lj0 + pj) * 128 + ti] : f[pj]);
                const T xo = x[pj];
                const T xn = pd_fma(-tau, tt, xo) * inv1ptau;
                xb[pj] = pd_fma(-omega, xo, opw * xn);
                x[pj] = xn;
            }
        }
        sxb[tj * 128 + ti] = xb[0];
        if (hh && l64 == 0) {
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) sEx[tj * PJ + pj] = xb[pj];
        }
        if (!(BPLTV_DBG(A) & 128)) __syncthreads();
        // ---- dual step
        const T xbdn = sxb[(tj + 1) * 128 + ti];   // xbar of the pixel row below the strip
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            if (it < lim[pj]) {
                const T b = xb[pj];
                // lane 63 of the right wave: own value; of the left wave: lane 0 of the right one
                const T xp1 = pd_lane_next_or(b, hh ? b : sEx[tj * PJ + pj]);
                const T xpM = (pj < PJ - 1) ? xb[pj < PJ - 1 ? pj + 1 : pj] : (hasD_last ? xbdn : b);
                const T d1 = xp1 - b;
                const T d2 = xpM - b;
                const T a = CL ? sfc[(RJ + lj0 + pj) * 128 + ti] : al[pj];
                T y1n = pd_fma(sigma, d1, y1[pj]);
                T y2n = pd_fma(sigma, d2, y2[pj]);
                if (rho != T(0)) {
                    const T den = T(1) + sigma * rho / a;
                    y1n = y1n / den;
                    y2n = y2n / den;
                }
                const T n2v = pd_fma(y2n, y2n, y1n * y1n);
                if (n2v > a * a) {
                    const T v = a * rsqrt_nr(n2v);
                    y1n = y1n * v;
                    y2n = y2n * v;
                }
                y1[pj] = y1n;
                y2[pj] = y2n;
            }
        }
        sy2[(tj + 1) * 128 + ti] = y2[PJ - 1];
        if (!hh && l64 == 63) {
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) sE1[tj * PJ + pj] = y1[pj];
        }
        tau = ntau; sigma = nsigma; omega = nomega; inv1ptau = ninv1ptau; opw = nopw;
        if (!(BPLTV_DBG(A) & 128)) __syncthreads();
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gi2 = oi + ti, gj2 = oj + lj0 + pj;
        if (gi2 >= ci0 && gi2 < ci1 && gj2 >= cj0 && gj2 < cj1 && !(BPLTV_DBG(A) & 2)) {
            const size_t idx = base + gi2 + (size_t)M * gj2;
            __hip_atomic_store(&Axout[idx], x[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay1out[idx], y1[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay2out[idx], y2[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// pdhg_rows2_kernel: pdhg_rows_kernel re-cut for instruction-level parallelism (round 4).  Same layout (a wave = one row
// of 64 lanes along i, PJ pixels per thread along j, DPP i-neighbours, strip ends through LDS, halo pixel rows that stop
// early), same arithmetic per pixel (bit-identical), but:
//   * interior waves (every pixel row runs all iterations -- 6 of the 8 waves of a 64 x 64 region, and every wave next
//     to an image border) run straight-line code: the dual chains of G pixels are issued together -- projection as
//     multiply by (outside ? alpha * rsqrt : 1.0), exact, instead of an exec-masked branch per pixel -- so that one
//     workgroup alone keeps a SIMD's f64 pipe busy while its neighbour loads or stores (pdhg_rows_kernel: one 25-deep
//     dependent chain per wave at a time);
//   * f lives in LDS (own cells, read back once per iteration at the top of the primal step): 2 PJ registers freed for
//     the interleaved chains;
//   * x ping-pongs between two register sets over a loop unrolled by two (no copy of the new x per pixel), the lane shift
//     for xbar(i+1) zero-fills lane 63 and the Neumann difference at the image's last column comes from a per-lane step
//     sigma * {1, 0} instead (no two moves per pixel to seed the shift).
// ------------------------------------------------------------------------------------------
constexpr size_t pdhg_rows2_lds(int PJ, int TJ, size_t word = sizeof(double)) {
    return word * 64 * (2 * (size_t)(TJ + 1) + 2 * (size_t)PJ * TJ);
}
__device__ __forceinline__ double pd_lane_next_or0(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float pd_lane_next_or0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

template <typename T, int PJ, int TJ, int G>
__global__ __launch_bounds__(64 * TJ, TJ / 2) void pdhg_rows2_kernel(PdhgArgs A) {   // two workgroups per CU: 128 VGPRs
    constexpr int RI = 64, RJ = PJ * TJ;
    static_assert(PJ % G == 0, "pixel groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char pdhg_smem[];
    T* sy2 = reinterpret_cast<T*>(pdhg_smem);   // [TJ + 1][64]: row tj + 1 = y2 of strip tj's last pixel row, row 0 = 0
    T* sxb = sy2 + (TJ + 1) * 64;               // [TJ + 1][64]: row tj = xbar of strip tj's first pixel row, row TJ = 0
    T* sf = sxb + (TJ + 1) * 64;                // [RJ][64]: f, every thread its own cells
    T* sal = sf + RJ * 64;                      // [RJ][64]: alpha, likewise
    const T* __restrict__ Axin = reinterpret_cast<const T*>(A.xin);
    const T* __restrict__ Ay1in = reinterpret_cast<const T*>(A.y1in);
    const T* __restrict__ Ay2in = reinterpret_cast<const T*>(A.y2in);
    T* __restrict__ Axout = reinterpret_cast<T*>(A.xout);
    T* __restrict__ Ay1out = reinterpret_cast<T*>(A.y1out);
    T* __restrict__ Ay2out = reinterpret_cast<T*>(A.y2out);
    const T* __restrict__ Af = reinterpret_cast<const T*>(A.f);
    const int tid = threadIdx.x, ti = tid & 63;
    const int tj = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: uniform
    PDHG_DECODE_BLOCK(A, imgl, ta, tb)
    const int img = A.img0 + imgl;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    int fimg, apar;
    pdhg_data_image(img, A.O, A.Odata, fimg, apar);
    const size_t base = (size_t)img * M * N;
    const size_t fbase = (size_t)fimg * M * N;
    const T* __restrict__ alpha = reinterpret_cast<const T*>(A.alpha) + (size_t)apar * A.astride;
    const int amode = (A.am == 1 && A.an == 1) ? 0 : ((A.am == M && A.an == N) ? 2 : 1);
    const bool first = A.first != 0;
    const int lj0 = PJ * tj;
    const int gi = min(oi + ti, M - 1);
    const bool in_i = oi + ti < M;

    T xa[PJ], xc[PJ], y1[PJ], y2[PJ];
    {
        T f[PJ], al[PJ];
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            const int gj = min(oj + lj0 + pj, N - 1);
            const size_t gidx = gi + (size_t)M * gj;
            size_t ai = 0;
            if (amode == 2) ai = gidx;
            else if (amode == 1) ai = ((unsigned)gi * (unsigned)A.am) / (unsigned)M + (size_t)A.am * (((unsigned)gj * (unsigned)A.an) / (unsigned)N);
            if (!first) {
                xa[pj] = Axin[base + gidx];
                y1[pj] = Ay1in[base + gidx];
                y2[pj] = Ay2in[base + gidx];
            }
            f[pj] = Af[fbase + gidx];
            al[pj] = alpha[ai];
        }
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            if (first) { xa[pj] = f[pj]; y1[pj] = T(0); y2[pj] = T(0); }
            if (!(in_i && oj + lj0 + pj < N)) { f[pj] = T(0); xa[pj] = T(0); y1[pj] = T(0); y2[pj] = T(0); al[pj] = T(0); }
            sf[(lj0 + pj) * 64 + ti] = f[pj];
            sal[(lj0 + pj) * 64 + ti] = al[pj];
        }
    }
    sy2[(tj + 1) * 64 + ti] = y2[PJ - 1];
    if (tj == 0) { sy2[ti] = T(0); sxb[TJ * 64 + ti] = T(0); }
    __syncthreads();

    const T rho = (T)A.rho;
    const int nit = A.nit;
    const bool hasD_last = oj + lj0 + PJ - 1 < N - 1;
    // the image's last column is lane 63 of the last region (images are at least a region wide): its forward difference
    // along i is +0.  The lane shift delivers 0 there; sigma * 0 keeps y1 = +0 exactly as xbar - xbar = +0 does.
    const T m1 = (ti == 63 && oi + 63 >= M - 1) ? T(0) : T(1);
    // iterations pixel row lj is still read for (see pdhg_rows_kernel)
    int lim[PJ];
    int lim_min = nit;
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int lj = lj0 + pj;
        int l = nit;
        if (oj > 0) l = min(l, lj);
        if (oj + RJ < N) l = min(l, RJ - lj);
        lim[pj] = l;
        lim_min = min(lim_min, l);
    }
    const bool interior = lim_min >= nit;   // wave-uniform
    const T* __restrict__ row = reinterpret_cast<const T*>(A.tab) + (size_t)TAB_STRIDE * A.it0;

    // one iteration: x from xi to xo
    auto iterate = [&](const T (&xi)[PJ], T (&xo)[PJ], int it) {
        const T tau = row[TAB_STRIDE * it + 0], sigma = row[TAB_STRIDE * it + 1], omega = row[TAB_STRIDE * it + 2],
                inv1ptau = row[TAB_STRIDE * it + 3], opw = row[TAB_STRIDE * it + 4];
        const T sig1 = sigma * m1;
        T xb[PJ];
        const T y2up = sy2[tj * 64 + ti];   // y2 of the pixel row above the strip (guard row of zeros at lj = 0)
        if (interior) {
#pragma unroll
            for (int g0 = 0; g0 < PJ; g0 += G) {
                T fv[G];
#pragma unroll
                for (int g = 0; g < G; ++g) fv[g] = sf[(lj0 + g0 + g) * 64 + ti];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int pj = g0 + g;
                    const T y1m = pd_lane_prev_or0(y1[pj]);
                    const T y2m = (pj > 0) ? y2[pj > 0 ? pj - 1 : 0] : y2up;
                    const T div = (y1m - y1[pj]) + (y2m - y2[pj]);
                    const T tt = div - fv[g];
                    const T xn = pd_fma(-tau, tt, xi[pj]) * inv1ptau;
                    xb[pj] = pd_fma(-omega, xi[pj], opw * xn);
                    xo[pj] = xn;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) {
                xb[pj] = T(0);
                xo[pj] = xi[pj];
                if (it < lim[pj]) {
                    const T y1m = pd_lane_prev_or0(y1[pj]);
                    const T y2m = (pj > 0) ? y2[pj > 0 ? pj - 1 : 0] : y2up;
                    const T div = (y1m - y1[pj]) + (y2m - y2[pj]);
                    const T tt = div - sf[(lj0 + pj) * 64 + ti];
                    const T xn = pd_fma(-tau, tt, xi[pj]) * inv1ptau;
                    xb[pj] = pd_fma(-omega, xi[pj], opw * xn);
                    xo[pj] = xn;
                }
            }
        }
        sxb[tj * 64 + ti] = xb[0];
        __syncthreads();
        const T xbdn = sxb[(tj + 1) * 64 + ti];   // xbar of the pixel row below the strip
        if (interior) {
#pragma unroll
            for (int g0 = 0; g0 < PJ; g0 += G) {
                T n2v[G], av[G];
#pragma unroll
                for (int g = 0; g < G; ++g) av[g] = sal[(lj0 + g0 + g) * 64 + ti];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int pj = g0 + g;
                    const T b = xb[pj];
                    const T xp1 = pd_lane_next_or0(b);
                    const T xpM = (pj < PJ - 1) ? xb[pj < PJ - 1 ? pj + 1 : pj] : (hasD_last ? xbdn : b);
                    const T d1 = xp1 - b;
                    const T d2 = xpM - b;
                    T y1n = pd_fma(sig1, d1, y1[pj]);
                    T y2n = pd_fma(sigma, d2, y2[pj]);
                    if (rho != T(0)) {
                        const T den = T(1) + sigma * rho / av[g];
                        y1n = y1n / den;
                        y2n = y2n / den;
                    }
                    y1[pj] = y1n;
                    y2[pj] = y2n;
                    n2v[g] = pd_fma(y2n, y2n, y1n * y1n);
                }
                bool any_out = false;
#pragma unroll
                for (int g = 0; g < G; ++g) any_out |= n2v[g] > av[g] * av[g];
                if (any_out) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int pj = g0 + g;
                        const T a = av[g];
                        const T v = (n2v[g] > a * a) ? a * rsqrt_nr(n2v[g]) : T(1);   // y * 1 = y exactly
                        y1[pj] = y1[pj] * v;
                        y2[pj] = y2[pj] * v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the groups apart: G chains in flight, not PJ
            }
        } else {
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) {
                if (it < lim[pj]) {
                    const T b = xb[pj];
                    const T xp1 = pd_lane_next_or0(b);
                    const T xpM = (pj < PJ - 1) ? xb[pj < PJ - 1 ? pj + 1 : pj] : (hasD_last ? xbdn : b);
                    const T d1 = xp1 - b;
                    const T d2 = xpM - b;
                    const T a = sal[(lj0 + pj) * 64 + ti];
                    T y1n = pd_fma(sig1, d1, y1[pj]);
                    T y2n = pd_fma(sigma, d2, y2[pj]);
                    if (rho != T(0)) {
                        const T den = T(1) + sigma * rho / a;
                        y1n = y1n / den;
                        y2n = y2n / den;
                    }
                    const T n2v = pd_fma(y2n, y2n, y1n * y1n);
                    if (n2v > a * a) {
                        const T v = a * rsqrt_nr(n2v);
                        y1n = y1n * v;
                        y2n = y2n * v;
                    }
                    y1[pj] = y1n;
                    y2[pj] = y2n;
                }
            }
        }
        sy2[(tj + 1) * 64 + ti] = y2[PJ - 1];
        __syncthreads();
    };
    int it = 0;
    for (; it + 1 < nit; it += 2) {
        iterate(xa, xc, it);
        iterate(xc, xa, it + 1);
    }
    if (it < nit) {
        iterate(xa, xc, it);
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) xa[pj] = xc[pj];
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gi2 = oi + ti, gj2 = oj + lj0 + pj;
        if (gi2 >= ci0 && gi2 < ci1 && gj2 >= cj0 && gj2 < cj1) {
            const size_t idx = base + gi2 + (size_t)M * gj2;
            __hip_atomic_store(&Axout[idx], xa[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay1out[idx], y1[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay2out[idx], y2[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// pdhg_stream_kernel: the T fused iterations as a PIPELINE OF WAVES that streams down the image (round 4).
//
// pdhg_rows_kernel loads a 64 x 64 region, iterates T times behind workgroup barriers and stores the core: its memory
// phases and its compute phase overlap only through the second workgroup of the CU, and the halo rows above and below the
// core are recomputed.  Here a workgroup owns a strip of 64 columns (lanes along i, halo T on each side that is not an
// image border, exactly as the rows kernel) and a segment of rows, and wave t (t = 1 .. nit) computes ITERATION t of the
// launch for one pixel row after the other:
//     wave 1 reads row r of the input state from HBM (prefetched PF rows ahead) and hands f(r), alpha(r) to everyone,
//     wave t reads row r of level t - 1 from an LDS ring, does the primal step of row r and -- one row behind, because the
//     dual step needs xbar of the row below -- the dual step of row r - 1, and hands row r - 1 of level t to wave t + 1,
//     wave nit stores its rows to HBM.
// Waves synchronise pairwise through two counters per level in LDS (rows produced, rows consumed): no workgroup barrier in
// the loop, loads, arithmetic and stores of a workgroup run all the time, and along j nothing is recomputed but the nit
// lead-in rows at each artificial end of a segment (the rows kernel: T halo rows of every 64).  Neighbours along i are the
// adjacent lanes (DPP), neighbours along j are the wave's own previous row (registers) and the next row of the level
// below (the ring).  Same arithmetic per pixel as every other variant: bit-identical results.
// Validity: a value of level t is wrong within t lanes of a strip edge that is not an image border (never stored: the
// core starts T >= nit lanes in) and within t rows of a segment end that is not an image border (never stored: a segment's
// region is its core plus T rows at each such end, tile_span).
// grid (strips, segments, images), block 64 * NL; images at least 64 pixels wide.
// ------------------------------------------------------------------------------------------
template <int NL, int D, int FD>
constexpr size_t pdhg_stream_lds(size_t word = sizeof(double)) {
    return word * 64 * ((size_t)(NL - 1) * D * 3 + (size_t)FD * 2) + sizeof(int) * 2 * (NL + 2);
}
__device__ __forceinline__ void pd_wait_ge(int* p, int v) {
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void pd_publish(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename T, int NL, int D, int FD, int PF, int WPE>
__global__ __launch_bounds__(64 * NL, WPE) void pdhg_stream_kernel(PdhgArgs A) {   // WPE: waves per SIMD the register budget allows
    static_assert((FD & (FD - 1)) == 0 && (D & (D - 1)) == 0, "ring depths: powers of two");
    extern __shared__ __attribute__((aligned(16))) unsigned char pdhg_smem[];
    T* ring = reinterpret_cast<T*>(pdhg_smem);            // [NL - 1][D][3][64]: output rows of levels 1 .. NL - 1
    T* fa = ring + (size_t)(NL - 1) * D * 3 * 64;         // [FD][2][64]: f and alpha rows
    int* prod = reinterpret_cast<int*>(fa + (size_t)FD * 2 * 64);   // prod[t]: rows level t has produced (t = 0: f / alpha rows)
    int* cons = prod + (NL + 2);                          // cons[t]: input rows wave t has consumed (read out of the ring below it)
    const T* __restrict__ Axin = reinterpret_cast<const T*>(A.xin);
    const T* __restrict__ Ay1in = reinterpret_cast<const T*>(A.y1in);
    const T* __restrict__ Ay2in = reinterpret_cast<const T*>(A.y2in);
    T* __restrict__ Axout = reinterpret_cast<T*>(A.xout);
    T* __restrict__ Ay1out = reinterpret_cast<T*>(A.y1out);
    T* __restrict__ Ay2out = reinterpret_cast<T*>(A.y2out);
    const T* __restrict__ Af = reinterpret_cast<const T*>(A.f);
    const int tid = threadIdx.x, ti = tid & 63;
    const int lev = __builtin_amdgcn_readfirstlane(tid >> 6) + 1;   // this wave's level: 1 .. NL (uniform)
    if (tid < 2 * (NL + 2)) prod[tid] = 0;
    __syncthreads();                                               // the only workgroup barrier
    const int nit = A.nit;
    if (lev > nit) return;                                         // a short last launch: the upper levels have nothing to do
    PDHG_DECODE_BLOCK(A, imgl, ta, tb)
    const int img = A.img0 + imgl;
    const int M = A.M, N = A.N;
    int oi, ci0, ci1, rs, cj0, cj1;
    tile_span(ta, M, 64, A.halo, oi, ci0, ci1);
    tile_span(tb, N, A.seg, A.halo, rs, cj0, cj1);                 // rows [rs, re) are processed, [cj0, cj1) are stored
    const int re = min(N, rs + A.seg);
    int fimg, apar;
    pdhg_data_image(img, A.O, A.Odata, fimg, apar);
    const size_t base = (size_t)img * M * N + (size_t)(oi + ti);
    const size_t fbase = (size_t)fimg * M * N + (size_t)(oi + ti);
    const T* __restrict__ alpha = reinterpret_cast<const T*>(A.alpha) + (size_t)apar * A.astride;
    const int amode = (A.am == 1 && A.an == 1) ? 0 : ((A.am == M && A.an == N) ? 2 : 1);
    const unsigned pa = (amode == 1) ? ((unsigned)(oi + ti) * (unsigned)A.am) / (unsigned)M : 0u;
    const bool first = A.first != 0;
    const bool last = lev == nit;                                  // this wave stores to HBM
    const bool core_i = (oi + ti >= ci0) && (oi + ti < ci1);
    const T rho = (T)A.rho;
    const T* __restrict__ row = reinterpret_cast<const T*>(A.tab) + (size_t)TAB_STRIDE * (A.it0 + lev - 1);
    const T tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    // the image's last column is lane 63 of the last strip: its forward difference along i is +0 (see pdhg_rows2_kernel)
    const T sig1 = sigma * ((ti == 63 && oi + 63 >= M - 1) ? T(0) : T(1));
    T* myring = ring + (size_t)(lev - 1) * D * 3 * 64;             // where this level's rows go (lev < nit)
    const T* inring = ring + (size_t)(lev - 2) * D * 3 * 64;       // where its input rows come from (lev > 1)
    const int nrows = re - rs;

    // level 1: its input rows come from HBM, PF rows in flight
    T pX[PF], pY1[PF], pY2[PF], pF[PF], pA[PF];
    auto fetch = [&](int q, int slot) {                            // row rs + q of the launch's input
        const int r = min(rs + q, N - 1);
        const size_t o = (size_t)M * r;
        pF[slot] = Af[fbase + o];
        if (!first) { pX[slot] = Axin[base + o]; pY1[slot] = Ay1in[base + o]; pY2[slot] = Ay2in[base + o]; }
        T a;
        if (amode == 0) a = alpha[0];
        else if (amode == 2) a = alpha[(size_t)(oi + ti) + o];
        else a = alpha[pa + (size_t)A.am * (((unsigned)r * (unsigned)A.an) / (unsigned)N)];
        pA[slot] = a;
    };
    if (lev == 1) {
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch(u, u);
    }
    T Y1p = T(0), Y2p = T(0), XBp = T(0), Xnp = T(0), Ap = T(0);   // of the previous row: y of the level below, xbar and x of this level, alpha
    // one row step; q = row index within the region, slot = q % PF (static)
    auto step = [&](int q, int slot) {
        T X, Y1, Y2, Fv, Av;
        if (lev == 1) {
            Fv = pF[slot]; Av = pA[slot];
            if (first) { X = Fv; Y1 = T(0); Y2 = T(0); }
            else { X = pX[slot]; Y1 = pY1[slot]; Y2 = pY2[slot]; }
            if (q + PF < nrows) fetch(q + PF, slot);
            if (nit > 1) {                                         // f and alpha of this row for the levels above
                if (q >= FD) pd_wait_ge(&cons[nit], q - FD + 1);   // the slot's previous row has been used by the last level
                fa[((q & (FD - 1)) * 2 + 0) * 64 + ti] = Fv;
                fa[((q & (FD - 1)) * 2 + 1) * 64 + ti] = Av;
                pd_publish(&prod[0], q + 1);
            }
        } else {
            pd_wait_ge(&prod[lev - 1], q + 1);                     // row q of the level below is in its ring
            const T* src = inring + (size_t)(q & (D - 1)) * 3 * 64;
            X = src[ti]; Y1 = src[64 + ti]; Y2 = src[128 + ti];
            Fv = fa[((q & (FD - 1)) * 2 + 0) * 64 + ti];           // (prod[0] > q follows: level 1 wrote it before its row q)
            Av = fa[((q & (FD - 1)) * 2 + 1) * 64 + ti];
            pd_publish(&cons[lev], q + 1);
        }
        // ---- primal step of row q
        const T y1m = pd_lane_prev_or0(Y1);
        const T y2m = (q > 0 || rs > 0) ? Y2p : T(0);              // row above; the image's first row has the zero guard
        const T div = (y1m - Y1) + (y2m - Y2);
        const T tt = div - Fv;
        const T xn = pd_fma(-tau, tt, X) * inv1ptau;
        const T xb = pd_fma(-omega, X, opw * xn);
        // ---- dual step of row q - 1 (needs xbar of row q)
        if (q > 0) {
            const T b = XBp;
            const T xp1 = pd_lane_next_or0(b);
            const T d1 = xp1 - b;
            const T d2 = xb - b;
            T y1n = pd_fma(sig1, d1, Y1p);
            T y2n = pd_fma(sigma, d2, Y2p);
            if (rho != T(0)) {
                const T den = T(1) + sigma * rho / Ap;
                y1n = y1n / den;
                y2n = y2n / den;
            }
            const T n2v = pd_fma(y2n, y2n, y1n * y1n);
            if (n2v > Ap * Ap) {
                const T v = Ap * rsqrt_nr(n2v);
                y1n = y1n * v;
                y2n = y2n * v;
            }
            const int qo = q - 1;                                  // the finished row of this level
            if (last) {
                const int gj = rs + qo;
                if (core_i && gj >= cj0 && gj < cj1) {
                    const size_t idx = base + (size_t)M * gj;
                    __hip_atomic_store(&Axout[idx], Xnp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&Ay1out[idx], y1n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&Ay2out[idx], y2n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (qo >= D) pd_wait_ge(&cons[lev + 1], qo - D + 1);   // the slot's previous row has been read
                T* dst = myring + (size_t)(qo & (D - 1)) * 3 * 64;
                dst[ti] = Xnp; dst[64 + ti] = y1n; dst[128 + ti] = y2n;
                pd_publish(&prod[lev], qo + 1);
            }
        }
        Y1p = Y1; Y2p = Y2; XBp = xb; Xnp = xn; Ap = Av;
    };
    int q = 0;
    for (; q + PF <= nrows; q += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(q + u, u);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (q + u < nrows) step(q + u, u);
    // the dual step of the last row: below it is the image's border (forward difference +0) or an artificial segment
    // end (the row is invalid and not stored; any finite value will do)
    {
        const T b = XBp;
        const T xp1 = pd_lane_next_or0(b);
        const T d1 = xp1 - b;
        const T d2 = b - b;
        T y1n = pd_fma(sig1, d1, Y1p);
        T y2n = pd_fma(sigma, d2, Y2p);
        if (rho != T(0)) {
            const T den = T(1) + sigma * rho / Ap;
            y1n = y1n / den;
            y2n = y2n / den;
        }
        const T n2v = pd_fma(y2n, y2n, y1n * y1n);
        if (n2v > Ap * Ap) {
            const T v = Ap * rsqrt_nr(n2v);
            y1n = y1n * v;
            y2n = y2n * v;
        }
        const int qo = nrows - 1;
        if (last) {
            const int gj = rs + qo;
            if (core_i && gj >= cj0 && gj < cj1) {
                const size_t idx = base + (size_t)M * gj;
                __hip_atomic_store(&Axout[idx], Xnp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&Ay1out[idx], y1n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&Ay2out[idx], y2n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (qo >= D) pd_wait_ge(&cons[lev + 1], qo - D + 1);
            T* dst = myring + (size_t)(qo & (D - 1)) * 3 * 64;
            dst[ti] = Xnp; dst[64 + ti] = y1n; dst[128 + ti] = y2n;
            pd_publish(&prod[lev], qo + 1);
        }
    }
}

template <typename T, int PJ, int WPB>
__global__ __launch_bounds__(64 * WPB, 2) void pdhg_wave_kernel(PdhgArgs A) {
    constexpr int RI = 32, RJ = 2 * PJ;
    const T* __restrict__ Axin = reinterpret_cast<const T*>(A.xin);
    const T* __restrict__ Ay1in = reinterpret_cast<const T*>(A.y1in);
    const T* __restrict__ Ay2in = reinterpret_cast<const T*>(A.y2in);
    T* __restrict__ Axout = reinterpret_cast<T*>(A.xout);
    T* __restrict__ Ay1out = reinterpret_cast<T*>(A.y1out);
    T* __restrict__ Ay2out = reinterpret_cast<T*>(A.y2out);
    const T* __restrict__ Af = reinterpret_cast<const T*>(A.f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int tile_g = (int)blockIdx.x * WPB + wave;
    if (tile_g >= A.ntiles) return;      // a whole wave leaves; there is no barrier in this kernel
    const int tilesPerImg = A.nTi * A.nTj;
    const int imgl = tile_g / tilesPerImg;
    const int img = A.img0 + imgl;
    const int t = tile_g - imgl * tilesPerImg;
    const int ta = t % A.nTi, tb = t / A.nTi;
    int oi, ci0, ci1, oj, cj0, cj1;
    tile_span(ta, A.M, RI, A.halo, oi, ci0, ci1);
    tile_span(tb, A.N, RJ, A.halo, oj, cj0, cj1);
    const int M = A.M, N = A.N;
    const size_t base = (size_t)img * M * N;
    const size_t fbase = (size_t)(img % A.Odata) * M * N;
    const T* __restrict__ alpha = reinterpret_cast<const T*>(A.alpha) + (size_t)(img / A.Odata) * A.astride;
    const int amode = (A.am == 1 && A.an == 1) ? 0 : ((A.am == M && A.an == N) ? 2 : 1);
    const bool first = A.first != 0;
    const int gi = oi + li, gic = min(gi, M - 1);
    T x[PJ], y1[PJ], y2[PJ], f[PJ], al[PJ];
    size_t gidx[PJ];
    // every global load is issued before the first use
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = min(oj + half * PJ + pj, N - 1);
        gidx[pj] = (size_t)gic + (size_t)M * gj;
        size_t ai = 0;
        if (amode == 2) ai = gidx[pj];
        else if (amode == 1) ai = ((unsigned)gic * (unsigned)A.am) / (unsigned)M + (size_t)A.am * (((unsigned)gj * (unsigned)A.an) / (unsigned)N);
        f[pj] = Af[fbase + gidx[pj]];
        al[pj] = alpha[ai];
    }
    if (!first) {
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            x[pj] = Axin[base + gidx[pj]];
            y1[pj] = Ay1in[base + gidx[pj]];
            y2[pj] = Ay2in[base + gidx[pj]];
        }
    }
    const bool in_i = gi < M;
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const bool in = in_i && (oj + half * PJ + pj < N);
        if (first) { x[pj] = f[pj]; y1[pj] = T(0); y2[pj] = T(0); }
        if (!in) { f[pj] = T(0); x[pj] = T(0); y1[pj] = T(0); y2[pj] = T(0); al[pj] = T(0); }
    }
    const T rho = (T)A.rho;
    // neighbour availability: lane to the right / row below inside the region AND inside the image (else the pixel's
    // own xbar is used: the forward difference is exactly +0, the Neumann border)
    const bool has_right = (li < RI - 1) && (gi < M - 1);
    const bool has_left = li > 0;
    const T* __restrict__ row = reinterpret_cast<const T*>(A.tab) + (size_t)TAB_STRIDE * A.it0;
    T tau = row[0], sigma = row[1], omega = row[2], inv1ptau = row[3], opw = row[4];
    for (int it = 0; it < A.nit; ++it) {
        const T* __restrict__ nrow = row + TAB_STRIDE * ((it + 1 < A.nit) ? it + 1 : it);
        const T ntau = nrow[0], nsigma = nrow[1], nomega = nrow[2], ninv1ptau = nrow[3], nopw = nrow[4];
        // ---- primal step.  Waves issue in order: the operations of G = 4 pixels are written stage by stage, so that
        // consecutive instructions are independent and the f64 pipe (4 cycles issue, longer latency) stays fed by the
        // two waves of a SIMD.
        constexpr int G = 4;
        static_assert(PJ % G == 0, "pixels per lane in groups of four");
        const T y2_other_last = pd_shfl(y2[PJ - 1], lane ^ 32);   // row PJ-1 of the other half (used by half 1, pj = 0)
        T xb[PJ];
#pragma unroll
        for (int g0 = 0; g0 < PJ; g0 += G) {
            T y1m[G], y2m[G], dv[G], xn[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int pj = g0 + u;
                y1m[u] = pd_lane_prev(y1[pj]);
                y2m[u] = (pj > 0) ? y2[pj - 1] : (half ? y2_other_last : T(0));
            }
#pragma unroll
            for (int u = 0; u < G; ++u) y1m[u] = has_left ? y1m[u] : T(0);
#pragma unroll
            for (int u = 0; u < G; ++u) { y1m[u] = y1m[u] - y1[g0 + u]; y2m[u] = y2m[u] - y2[g0 + u]; }
#pragma unroll
            for (int u = 0; u < G; ++u) dv[u] = y1m[u] + y2m[u];
#pragma unroll
            for (int u = 0; u < G; ++u) dv[u] = dv[u] - f[g0 + u];
#pragma unroll
            for (int u = 0; u < G; ++u) xn[u] = pd_fma(-tau, dv[u], x[g0 + u]);
#pragma unroll
            for (int u = 0; u < G; ++u) xn[u] = xn[u] * inv1ptau;
#pragma unroll
            for (int u = 0; u < G; ++u) dv[u] = opw * xn[u];
#pragma unroll
            for (int u = 0; u < G; ++u) { xb[g0 + u] = pd_fma(-omega, x[g0 + u], dv[u]); x[g0 + u] = xn[u]; }
        }
        // ---- dual step
        const T xb_other_first = pd_shfl(xb[0], lane ^ 32);       // row 0 of the other half (used by half 0, pj = PJ-1)
        T n2v[PJ];
        bool any_out = false;
#pragma unroll
        for (int g0 = 0; g0 < PJ; g0 += G) {
            T xp1[G], xpM[G], n1[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int pj = g0 + u;
                xp1[u] = pd_lane_next(xb[pj]);
                xpM[u] = (pj < PJ - 1) ? xb[pj + 1] : (half ? xb[pj] : xb_other_first);
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int pj = g0 + u, lj = half * PJ + pj;
                const bool has_down = (lj < RJ - 1) && (oj + lj < N - 1);
                xp1[u] = has_right ? xp1[u] : xb[pj];
                xpM[u] = has_down ? xpM[u] : xb[pj];
            }
#pragma unroll
            for (int u = 0; u < G; ++u) { xp1[u] = xp1[u] - xb[g0 + u]; xpM[u] = xpM[u] - xb[g0 + u]; }
#pragma unroll
            for (int u = 0; u < G; ++u) { y1[g0 + u] = pd_fma(sigma, xp1[u], y1[g0 + u]); y2[g0 + u] = pd_fma(sigma, xpM[u], y2[g0 + u]); }
            if (rho != T(0)) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const T den = T(1) + sigma * rho / al[g0 + u];
                    y1[g0 + u] = y1[g0 + u] / den;
                    y2[g0 + u] = y2[g0 + u] / den;
                }
            }
#pragma unroll
            for (int u = 0; u < G; ++u) n1[u] = y1[g0 + u] * y1[g0 + u];
#pragma unroll
            for (int u = 0; u < G; ++u) n2v[g0 + u] = pd_fma(y2[g0 + u], y2[g0 + u], n1[u]);
#pragma unroll
            for (int u = 0; u < G; ++u) any_out |= n2v[g0 + u] > al[g0 + u] * al[g0 + u];
        }
        if (any_out) {
            // The PJ Newton chains of a lane are independent.  They are written step by step ACROSS the pixels and
            // pinned (empty asm) so that the compiler neither sinks each chain under its pixel's condition -- one
            // exec-masked block per pixel, 17 dependent operations each with nothing to overlap -- nor re-serialises
            // them: the f64 pipe then always has PJ independent operations in flight.  Same operations per pixel as
            // rsqrt_nr (bit-exact).
            T r[PJ], hh[PJ];
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) rsqrt_nr_seed(n2v[pj], r[pj], hh[pj]);
#pragma unroll
            for (int k = 0; k < rsqrt_nr_steps(T(0)); ++k) {
#pragma unroll
                for (int pj = 0; pj < PJ; ++pj) {
                    const T tq = r[pj] * r[pj];
                    const T w = pd_fma(-hh[pj], tq, T(1.5));
                    r[pj] = r[pj] * w;
                }
#pragma unroll
                for (int pj = 0; pj < PJ; ++pj) asm volatile("" : "+v"(r[pj]));
            }
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) {
                const T a = al[pj];
                const T v = a * r[pj];
                const bool outp = n2v[pj] > a * a;
                y1[pj] = outp ? y1[pj] * v : y1[pj];
                y2[pj] = outp ? y2[pj] * v : y2[pj];
            }
        }
        tau = ntau; sigma = nsigma; omega = nomega; inv1ptau = ninv1ptau; opw = nopw;
    }
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
        const int gj = oj + half * PJ + pj;
        if (gi >= ci0 && gi < ci1 && gj >= cj0 && gj < cj1) {
            const size_t idx = base + gi + (size_t)M * gj;
            __hip_atomic_store(&Axout[idx], x[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay1out[idx], y1[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&Ay2out[idx], y2[pj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Two launch chains of equal launches have two stable ways of sharing the chip (tools/chain_phase.py stamps every launch):
// OUT of phase -- chain 1's launches start at 0.45-0.58 of chain 0's period, one chain's launch gap falls into the other's
// arithmetic, 5.22 ms per 5000 iterations of the reference batch -- and IN phase -- both chains launch together and
// compute together, phase 0.00 +- 0.05 for the whole sequence, 6.55 ms: the rate of a single chain.  Which one a solve
// falls into is settled in its first few launches and not by anything the host controls (8-36 % of the solves were slow
// ones, tools/step_jitter.py).  This one-wave kernel sits in chain 1's graph a few launches in.  It watches the word chain
// 0's launches rewrite, times two consecutive launch starts of chain 0 (its own clock reads cost nothing: it is the only
// wave of its launch) and holds chain 1's queue until the MIDDLE of chain 0's period; both states being stable the sequence
// stays out of phase from there.  (Letting chain 1 go right at a launch start of chain 0 did the opposite: its next launch
// is queued already and starts within a fraction of a period -- 30-40 % slow solves.)  About two periods per solve,
// <= 0.4 % of it; never longer than 200 us (chain 0 finished or not running: nothing changes the word).
// check != 0 (later in the sequence): the period is the one the first gate measured (phase[1]); chain 1 is held only if
// chain 0's next launch start comes within 15 % of a period of this kernel's own start or later than 85 % -- in phase;
// otherwise the kernel ends after half a period on average.
__global__ __launch_bounds__(64) void pdhg_phase_gate_kernel(unsigned* __restrict__ phase, int check) {
    if (threadIdx.x != 0) return;
    const long long t0 = (long long)wall_clock64();                       // 100 MHz
    long long ta[2] = {0, 0};
    unsigned c = __hip_atomic_load(phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nobs = check ? 1 : 2;
    for (int k = 0; k < nobs; ++k) {
        unsigned n = c;
        while ((n = __hip_atomic_load(phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == c) {
            if ((long long)wall_clock64() - t0 > 20000) return;
            __builtin_amdgcn_s_sleep(1);
        }
        ta[k] = (long long)wall_clock64();
        c = n;
    }
    long long per, last;
    if (check) {
        per = (long long)phase[1];
        last = ta[0];
        const long long d = last - t0;
        if (per < 100 || per > 20000 || (20 * d > 3 * per && 20 * d < 17 * per)) return;   // no period on record, or out of phase
    } else {
        per = ta[1] - ta[0];
        last = ta[1];
        phase[1] = (unsigned)per;
    }
    const long long until = last + per / 2;
    while ((long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(1);
}

__global__ __launch_bounds__(256) void pdhg_init_kernel(const double* __restrict__ f, const double* __restrict__ alpha,
                                                        int am, int an, int M, int N, int Odata, int astride, size_t total,
                                                        int init, int dual_first, double sigma, double rho,
                                                        double* __restrict__ x, double* __restrict__ y1,
                                                        double* __restrict__ y2) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t npx = (size_t)M * N;
    const int img = (int)(e / npx);
    const int q = (int)(e - (size_t)img * npx);
    const int i = q % M, j = q / M;
    const double* __restrict__ fk = f + (size_t)(img % Odata) * npx;
    const double x0 = init ? 0.0 : fk[q];
    double y1n = 0.0, y2n = 0.0;
    if (dual_first) {
        const double xi1 = (i < M - 1) ? (init ? 0.0 : fk[q + 1]) : x0;
        const double xj1 = (j < N - 1) ? (init ? 0.0 : fk[q + M]) : x0;
        const double d1 = (i < M - 1) ? xi1 - x0 : 0.0;
        const double d2 = (j < N - 1) ? xj1 - x0 : 0.0;
        const double a = alpha_at(alpha + (size_t)(img / Odata) * astride, am, an, M, N, i, j);
        y1n = __builtin_fma(sigma, d1, 0.0);
        y2n = __builtin_fma(sigma, d2, 0.0);
        if (rho != 0.0) {
            const double den = 1.0 + sigma * rho / a;
            y1n = y1n / den;
            y2n = y2n / den;
        }
        const double n2 = __builtin_fma(y2n, y2n, y1n * y1n);
        if (n2 > a * a) {
            const double v = a * rsqrt_nr(n2);
            y1n = y1n * v;
            y2n = y2n * v;
        }
    }
    x[e] = x0;
    y1[e] = y1n;
    y2[e] = y2n;
}

// The closing primal step of a dual-first run (pdhg_x_pass of the oracle; xbar is not needed any more).
// In place: a thread reads neighbouring duals and its own x only.
__global__ __launch_bounds__(256) void pdhg_xstep_kernel(const double* __restrict__ f, const double* __restrict__ y1,
                                                         const double* __restrict__ y2, const double* __restrict__ tabrow,
                                                         int M, int N, int Odata, size_t total, double* __restrict__ x) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t npx = (size_t)M * N;
    const int img = (int)(e / npx);
    const int q = (int)(e - (size_t)img * npx);
    const int i = q % M, j = q / M;
    const double tau = tabrow[0], inv1ptau = tabrow[3];
    const double y1m = (i > 0) ? y1[e - 1] : 0.0;
    const double y2m = (j > 0) ? y2[e - M] : 0.0;
    const double div = (y1m - y1[e]) + (y2m - y2[e]);
    const double t = div - f[(size_t)(img % Odata) * npx + q];
    x[e] = __builtin_fma(-tau, t, x[e]) * inv1ptau;
}

// ------------------------------------------------------------------------------------------
// Reductions: wave64 shuffle -> LDS -> one partial per block; a second single-block kernel sums
// the partials in a fixed order (bitwise reproducible, no atomics).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sh /* NT/64 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) r += sh[k];
    }
    __syncthreads();
    return r;  // valid on thread 0
}

// loss (A5): 0.5*||u - ubar||^2, per image.  grid (nblk, O), block 256.
__global__ __launch_bounds__(256) void cost_partial_kernel(const double* __restrict__ u,
                                                           const double* __restrict__ ubar, int npx,
                                                           double* __restrict__ partial) {
    __shared__ double sh[4];
    const size_t base = (size_t)blockIdx.y * npx;
    double s = 0.0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < npx; q += gridDim.x * 256) {
        const double d = u[base + q] - ubar[base + q];
        s = __builtin_fma(d, d, s);
    }
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// Same for a parameter sweep: problem blockIdx.y compares with ubar[blockIdx.y % Odata].
__global__ __launch_bounds__(256) void cost_partial_mod_kernel(const double* __restrict__ u,
                                                               const double* __restrict__ ubar, int npx,
                                                               int Odata, double* __restrict__ partial) {
    __shared__ double sh[4];
    const size_t base = (size_t)blockIdx.y * npx, bbase = (size_t)(blockIdx.y % Odata) * npx;
    double s = 0.0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < npx; q += gridDim.x * 256) {
        const double d = u[base + q] - ubar[bbase + q];
        s = __builtin_fma(d, d, s);
    }
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// per_image[k] = scale * sum_b partial[k][b];  total[0] = sum_k per_image[k].  one block of 256.
__global__ __launch_bounds__(256) void sum_final_kernel(const double* __restrict__ partial, int nblk,
                                                        int O, double scale,
                                                        double* __restrict__ per_image,
                                                        double* __restrict__ total) {
    for (int k = threadIdx.x; k < O; k += 256) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += partial[(size_t)k * nblk + b];
        per_image[k] = scale * s;
    }
    __syncthreads();
    if (threadIdx.x == 0 && total) {
        double s = 0.0;
        for (int k = 0; k < O; ++k) s += per_image[k];
        total[0] = s;
    }
}

// Primal-dual gap pieces per image: partial[(k*nblk+b)*4 + {0:||u-f||^2, 1:sum alpha|Gu|,
// 2:||f||^2, 3:||f - G^T y||^2}].  grid (nblk, O), block 256.
__global__ __launch_bounds__(256) void gap_partial_kernel(const double* __restrict__ u,
                                                          const double* __restrict__ y1,
                                                          const double* __restrict__ y2,
                                                          const double* __restrict__ f,
                                                          const double* __restrict__ alpha, int am,
                                                          int an, int M, int N,
                                                          double* __restrict__ partial) {
    __shared__ double sh[4];
    const int npx = M * N;
    const size_t base = (size_t)blockIdx.y * npx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < npx; q += gridDim.x * 256) {
        const int i = q % M, j = q / M;
        const double uk = u[base + q], fk = f[base + q];
        const double d1 = (i < M - 1) ? u[base + q + 1] - uk : 0.0;
        const double d2 = (j < N - 1) ? u[base + q + M] - uk : 0.0;
        const double a1 = (i < M - 1) ? y1[base + q] : 0.0, a1m = (i > 0) ? y1[base + q - 1] : 0.0;
        const double a2 = (j < N - 1) ? y2[base + q] : 0.0, a2m = (j > 0) ? y2[base + q - M] : 0.0;
        const double w = (a1m - a1) + (a2m - a2);
        const double r = uk - fk;
        s0 += r * r;
        s1 += alpha_at(alpha, am, an, M, N, i, j) * sqrt(d1 * d1 + d2 * d2);
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

__global__ __launch_bounds__(256) void gap_final_kernel(const double* __restrict__ partial, int nblk,
                                                        int O, double* __restrict__ gap,
                                                        double* __restrict__ gap_max) {
    for (int k = threadIdx.x; k < O; k += 256) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < nblk; ++b)
            for (int c = 0; c < 4; ++c) s[c] += partial[((size_t)k * nblk + b) * 4 + c];
        gap[k] = (0.5 * s[0] + s[1]) - (0.5 * s[2] - 0.5 * s[3]);
    }
    __syncthreads();
    if (threadIdx.x == 0 && gap_max) {
        double m = gap[0];
        for (int k = 1; k < O; ++k) m = (gap[k] > m) ? gap[k] : m;
        gap_max[0] = m;
    }
}

// dtype = 32 handles: narrowing of the kernel's inputs, widening of its result.  grid-stride, block 256.
// Parameter check on the device (bpltv_denoise_device): out[0] = bit pattern of the smallest valid entry (non-negative
// doubles order like their bits; preset to all ones), out[1] != 0 if an entry is not finite or negative -- the check
// the host entry points run on the host array.  Grid-stride, one atomic pair per workgroup.
__global__ __launch_bounds__(256) void alpha_check_kernel(const double* __restrict__ a, size_t n, unsigned long long* __restrict__ out) {
    __shared__ double smin[4];
    __shared__ int sbad[4];
    double mn = __builtin_huge_val();
    int bad = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const double v = a[e];
        if (!(v >= 0.0) || !(v < __builtin_huge_val())) bad = 1;   // NaN, negative, infinite
        else mn = fmin(mn, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_down(mn, off));
        bad |= __shfl_down(bad, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smin[w] = mn; sbad[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        atomicMin(out, (unsigned long long)__double_as_longlong(m));
        if (sbad[0] | sbad[1] | sbad[2] | sbad[3]) atomicOr(out + 1, 1ull);
    }
}

__global__ __launch_bounds__(256) void cvt_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) dst[e] = (float)src[e];
}
__global__ __launch_bounds__(256) void cvt_f32_f64_kernel(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) dst[e] = (double)src[e];
}

// FwdGradientOp / adjoint for one M x N image (operator tests, A4).
__global__ void grad_fwd_kernel(const double* __restrict__ x, int M, int N, double* __restrict__ d1,
                                double* __restrict__ d2) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * N) return;
    const int i = q % M, j = q / M;
    d1[q] = (i < M - 1) ? x[q + 1] - x[q] : 0.0;
    d2[q] = (j < N - 1) ? x[q + M] - x[q] : 0.0;
}

__global__ void grad_fwd_T_kernel(const double* __restrict__ y1, const double* __restrict__ y2, int M,
                                  int N, double* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * N) return;
    const int i = q % M, j = q / M;
    const double a = (i < M - 1) ? y1[q] : 0.0, am = (i > 0) ? y1[q - 1] : 0.0;
    const double b = (j < N - 1) ? y2[q] : 0.0, bm = (j > 0) ? y2[q - M] : 0.0;
    out[q] = (am - a) + (bm - b);
}

}  // namespace bpltv
