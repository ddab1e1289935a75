// tiling.hpp -- the host-side planning arithmetic of the PDHG path, free of HIP: 1-D tiling with halos (tile_span,
// tile_count), the block distribution of images / parameter blocks over devices (shard_range), the launch-chain
// phase rule, and the region / fusion-depth plan of a solve (plan_pdhg).  Plain C++17: compiled into libbpltv by hipcc
// (tile_span also runs on the device) and into tools/plan_host_check.cpp by g++ -fsanitize=address,undefined
// (tests/test_host_sanitize.py) -- the sanitizer coverage SURVEY section 5 asks for on the library's host-only code.
//
// What is tiled: the images of /root/reference/src/TVLearningFunctionVec.jl:45-70 (denoise), one ROF problem each.
#pragma once
#include <algorithm>
#include <cmath>

#if defined(__HIPCC__)
#define BPLTV_HD __host__ __device__
#else
#define BPLTV_HD
#endif

namespace bpltv {

// 1-D tiling with halo: region length R, halo T, image length L.  Tile a covers region
// [o, o+R) and owns (writes back) the core [c0, c1).  Image borders need no halo (Neumann).
BPLTV_HD inline void tile_span(int a, int L, int R, int T, int& o, int& c0, int& c1) {
    if (L <= R) {
        o = 0; c0 = 0; c1 = L;
        return;
    }
    const int S = R - 2 * T;
    const int cs = (a == 0) ? 0 : (R - T) + (a - 1) * S;
    int oo = (a == 0) ? 0 : cs - T;
    if (oo + R >= L) {
        oo = L - R;
        c1 = L;
    } else {
        c1 = oo + R - T;
    }
    o = oo;
    c0 = cs;
}

inline int tile_count(int L, int R, int T) {
    if (L <= R) return 1;
    if (R - 2 * T < 1) return -1;
    for (int a = 0;; ++a) {
        int o, c0, c1;
        tile_span(a, L, R, T, o, c0, c1);
        if (c1 >= L) return a + 1;
    }
}

// Block distribution of O units (images; the K parameter blocks of a sweep) over `world` shards: the first O % world
// shards get one more.
inline void shard_range(int O, int world, int rank, int* lo, int* hi) {
    const int base = O / world, rem = O % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

// May an odd launch chain run half a launch out of phase (first launch T/2 iterations, one launch more, starting in the
// other state set)?  Only when that leaves it in the same final set as the even chains, and for long sequences.
inline bool chain_out_of_phase(int niter, int T, bool from_state) {
    const int nl0 = (niter + T - 1) / T, h0 = T / 2;
    return !from_state && T >= 2 && nl0 >= 8 && ((1 + (niter - h0 + T - 1) / T) - nl0) % 2 == 1;
}

// Geometry of one PDHG kernel variant as the planner sees it (the variant table of bpltv.hip supplies these).
struct PlanVariant {
    int RI, RJ;            // region (core + halo) of one workgroup / tile
    int tiles_per_block;   // > 1: several one-wave tiles per workgroup (pdhg_wave_kernel)
    int min_image;         // 1: the image must be at least as large as the region (pdhg_rows_kernel); 2: at least as wide (M >= RI,
                           // pdhg_stream_kernel: its region along j is a segment of rows, any image height); 3: and M even
                           // (pdhg_stream2_kernel: its loader moves pairs of pixels, 16-byte pieces)
    int tmax;              // > 0: most iterations one launch can fuse (pdhg_stream_kernel: one wave per iteration)
};
inline bool plan_too_small(const PlanVariant& V, int M, int N) {
    return (V.min_image == 1 && (M < V.RI || N < V.RJ)) || (V.min_image >= 2 && M < V.RI) || (V.min_image == 3 && (M & 1));
}
struct PlanRequest {
    int M, N, nimg;        // image size, images of the solve (K * O for a sweep)
    int ncu;               // compute units of the device (0: 256)
    int maxiter;
    int tile_iters;        // params.tile_iters (0 = choose)
    int variant;           // params.reserved[0] (1-based; 0 = choose)
    int chains;            // params.reserved[1] (0 = choose)
};
struct Plan {
    int variant, T, nTi, nTj, grid, chains;
};
enum { PLAN_OK = 0, PLAN_E_VARIANT = 1, PLAN_E_MIN_IMAGE = 2, PLAN_E_TILE_ITERS = 3, PLAN_E_TILING = 4, PLAN_E_GRID = 5 };

// Variant indices (0-based) the automatic choice uses; they index the table handed in (bpltv.hip: kVariants).
constexpr int PLAN_V_TILE32 = 0, PLAN_V_TILE48 = 12, PLAN_V_ROWS64 = 18, PLAN_V_ROWS48 = 19;

// Region and fusion depth of one solve.  Fitted on MI355X; results never depend on the plan.
inline int plan_pdhg(const PlanRequest& q, const PlanVariant* tab, int ntab, Plan* pl) {
    int v = q.variant - 1;  // explicit variant (1-based), 0 = auto
    const int M = q.M, N = q.N;
    const int ncu = q.ncu > 0 ? q.ncu : 256;
    if (v < 0) {
        // Large images: 64-lane rows (pdhg_rows_kernel: 64x64 region, 8 px per thread; 64x48 when that still fits the
        // chip in one round of two workgroups per CU -- measured on 1 ... 16 x 1024^2, 4 x 512^2, 8 x 300^2, 2 x 2048^2,
        // 3 x 1100x700: 1.2-1.5 x the 48x48 tile kernel); images narrower than a region keep the 48x48 tile kernel.
        v = PLAN_V_TILE32;
        if (M > 256 || N > 256) {
            v = PLAN_V_TILE48;
            if (M >= 64 && N >= 64) {
                auto tiles = [&](int vv) { return (double)tile_count(M, tab[vv].RI, 8) * tile_count(N, tab[vv].RJ, 8) * q.nimg; };
                v = (tiles(PLAN_V_ROWS64) <= 2.0 * ncu && tiles(PLAN_V_ROWS48) <= 2.0 * ncu) ? PLAN_V_ROWS48 : PLAN_V_ROWS64;
            }
        }
    }
    if (v >= ntab) return PLAN_E_VARIANT;
    int T = q.tile_iters;
    const bool auto_variant = q.variant <= 0;
    if (T <= 0 && v == PLAN_V_TILE32) {
        // Region and fusion depth from a launch-cost model: a launch costs a fixed ~4.5 us plus T iterations; an
        // iteration of the 32x32 / 1 px kernel takes ~0.47 us while every CU holds at most one workgroup and ~0.94 us
        // per round of two co-resident workgroups beyond that, one of the 48x48 / 3 px kernel 1.36 us resp. 2.1 us.
        // Deeper fusion means fewer launches but smaller cores, i.e. more (redundant) tiles; the larger region wastes
        // fewer pixels on halos and wins once the batch no longer fits the chip with 32x32 regions.  The rows kernels
        // (64x64 / 8 px: 1.95 us alone on a CU, 3.5 us per round of two; 64x48 / 6 px: 1.6 / 2.7 us; depth 8 only:
        // their halo is one wave) join for images of at least one region.
        struct Cand { int v; double one, two; int tmin, tmax; };
        const Cand cands[4] = {{PLAN_V_TILE32, 0.47, 0.94, 2, 12}, {PLAN_V_TILE48, 1.36, 2.1, 2, 12},
                               {PLAN_V_ROWS64, 1.95, 3.5, 8, 8}, {PLAN_V_ROWS48, 1.6, 2.7, 8, 8}};
        double best = 1e300;
        for (const Cand& cd : cands) {
            if (cd.v != PLAN_V_TILE32 && !auto_variant) continue;
            if (cd.v >= ntab) continue;
            const PlanVariant& Vc = tab[cd.v];
            if (plan_too_small(Vc, M, N)) continue;
            for (int t = cd.tmin; t <= cd.tmax; ++t) {
                if ((M > Vc.RI && 2 * t >= Vc.RI) || (N > Vc.RJ && 2 * t >= Vc.RJ)) continue;   // no core left
                const int a = tile_count(M, Vc.RI, t), b = tile_count(N, Vc.RJ, t);
                if (a < 1 || b < 1) continue;
                const double tiles = (double)a * b * q.nimg;
                const double rounds = tiles / (2.0 * ncu);
                // whole rounds for the short tile workgroups; the long rows workgroups overlap the tail of a round
                const double eff = Vc.min_image ? (rounds <= 1.0 ? 1.0 : 0.5 * (std::ceil(rounds) + rounds)) : std::ceil(rounds);
                const double per_iter = (tiles <= ncu) ? cd.one : cd.two * eff;
                double cost = std::ceil((double)std::max(q.maxiter, 1) / t) * (4.5 + t * per_iter);
                // two launch chains (below: from 1.5 workgroups per CU) hide part of every launch of the 32x32 kernel:
                // 10 images at T = 8 and 5 images at T = 10 take 0.74 / 0.69 of what the single-chain model says (only
                // within one round of workgroups, and not for shallow fusion)
                if (cd.v == PLAN_V_TILE32 && q.chains != 1 && q.nimg >= 2 && 2.0 * tiles > 3.0 * ncu && tiles <= 2.0 * ncu && t >= 6) cost *= 0.72;
                if (cost < best) { best = cost; T = t; v = cd.v; }
            }
        }
    }
    if (plan_too_small(tab[v], M, N)) {
        if (!auto_variant) return PLAN_E_MIN_IMAGE;
        v = PLAN_V_TILE48;
    }
    const PlanVariant& V = tab[v];
    if (T <= 0) {
        // Large images: depth 8 (the rows kernels: one halo wave at each end of the region), or 6 for the 48x48 tile
        // kernel once the grid is many times the chip
        T = 8;
        if (v == PLAN_V_TILE48 && (double)tile_count(M, V.RI, 8) * tile_count(N, V.RJ, 8) * q.nimg > 8192.0) T = 6;
    }
    // the halo must leave a core when the image is larger than the region
    auto maxT = [](int L, int R) { return (L <= R) ? (1 << 20) : (R - 1) / 2; };
    int cap = std::min(maxT(M, V.RI), maxT(N, V.RJ));
    if (V.tmax > 0) cap = std::min(cap, V.tmax);
    if (T > cap) T = cap;
    if (T < 1) return PLAN_E_TILE_ITERS;
    pl->variant = v;
    pl->T = T;
    pl->nTi = tile_count(M, V.RI, T);
    pl->nTj = tile_count(N, V.RJ, T);
    if (pl->nTi < 1 || pl->nTj < 1) return PLAN_E_TILING;
    const long long grid = (long long)pl->nTi * pl->nTj * q.nimg;
    if (grid > 0x7FFFFFFFll) return PLAN_E_GRID;   // workgroups of one launch are counted in an int
    pl->grid = (int)grid;
    // Independent image groups ("chains") of the launch graph: two chains once the batch is well beyond one workgroup
    // per CU (10 x 128^2: 7.2e5 -> 8.5e5 it/s; 8 / 32 / 64 images +18 / 26 / 16 %; large images +2 %); smaller batches
    // are faster as one chain.  Chain 0 runs on the handle's own stream, chain 1 on a second one: those two hardware
    // queues overlap; a third does not.
    int ch = q.chains;
    if (ch <= 0) ch = (2 * (long)pl->grid > 3 * (long)ncu && q.nimg >= 2) ? 2 : 1;
    if (ch > q.nimg) ch = q.nimg;
    pl->chains = ch;
    return PLAN_OK;
}

}  // namespace bpltv
