// plan_host_check.cpp -- host fuzz of csrc/tiling.hpp: the tiling with halos (tile_span / tile_count), the block
// distribution over devices (shard_range), the launch-chain phase rule and the region / fusion-depth planner
// (plan_pdhg) on random and edge shapes.  Plain C++17, no GPU; tests/test_host_sanitize.py builds it with
// g++ -fsanitize=address,undefined and runs it (sanitizers row: the library's host-only code).
//
// Invariants checked (what the kernels rely on -- an out-of-range tile is an out-of-bounds access on the device):
//   * the cores [c0, c1) of tiles 0 .. n-1 partition [0, L) exactly; every region [o, o + R) lies inside the image;
//   * a core edge that is not an image border sits at least T pixels inside its region (halo deep enough for T fused
//     iterations); the first / last region touches the image border it needs no halo for;
//   * tile_count is the number of tiles tile_span needs, and -1 exactly when no core is left;
//   * shard_range partitions [0, O) into `world` consecutive, balanced ranges;
//   * a plan has T >= 1, a core left in both directions, grid = nTi * nTj * nimg, 1 <= chains <= nimg, and the chains'
//     image ranges partition the images; an out-of-phase chain ends in the same state set as an in-phase one.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../bpldenoising_amd/csrc/tiling.hpp"

using namespace bpltv;

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() {   // splitmix64
    unsigned long long z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 16);
}
static int rint_in(int lo, int hi) { return lo + (int)(rnd() % (unsigned)(hi - lo + 1)); }

static int bad = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            if (bad < 20) { printf("FAIL: " __VA_ARGS__); printf("\n"); } \
            ++bad;                                         \
        }                                                  \
    } while (0)

static void check_tiling(int L, int R, int T) {
    const int n = tile_count(L, R, T);
    if (L > R && R - 2 * T < 1) { CHECK(n == -1, "tile_count(%d,%d,%d) = %d, expected -1", L, R, T, n); return; }
    CHECK(n >= 1, "tile_count(%d,%d,%d) = %d", L, R, T, n);
    if (n < 1) return;
    int expect = 0;
    for (int a = 0; a < n; ++a) {
        int o, c0, c1;
        tile_span(a, L, R, T, o, c0, c1);
        CHECK(c0 == expect, "L %d R %d T %d tile %d: core starts at %d, previous ended at %d", L, R, T, a, c0, expect);
        CHECK(c1 > c0 && c1 <= L, "L %d R %d T %d tile %d: core [%d,%d)", L, R, T, a, c0, c1);
        CHECK(o >= 0 && o + std::min(R, L) <= L, "L %d R %d T %d tile %d: region [%d,%d) leaves the image", L, R, T, a, o, o + R);
        CHECK(c0 >= o && c1 <= o + std::min(R, L), "L %d R %d T %d tile %d: core outside its region", L, R, T, a);
        if (L > R) {
            if (c0 > 0) CHECK(c0 - o >= T, "L %d R %d T %d tile %d: near halo %d < T", L, R, T, a, c0 - o);
            else CHECK(o == 0, "L %d R %d T %d tile %d: first region does not start at the border", L, R, T, a);
            if (c1 < L) CHECK(o + R - c1 >= T, "L %d R %d T %d tile %d: far halo %d < T", L, R, T, a, o + R - c1);
            else CHECK(o + R == L, "L %d R %d T %d tile %d: last region does not end at the border", L, R, T, a);
        }
        CHECK((a == n - 1) == (c1 == L), "L %d R %d T %d tile %d of %d ends at %d", L, R, T, a, n, c1);
        expect = c1;
    }
    CHECK(expect == L, "L %d R %d T %d: cores end at %d", L, R, T, expect);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200000;
    // the variant geometries of csrc/bpltv.hip (region RI x RJ, tiles per workgroup, needs an image of a region)
    const PlanVariant tab[] = {{32, 32, 1, 0, 0}, {64, 64, 1, 0, 0}, {16, 16, 1, 0, 0}, {32, 32, 1, 0, 0}, {64, 32, 1, 0, 0}, {64, 64, 1, 0, 0}, {64, 16, 1, 0, 0},
                               {128, 16, 1, 0, 0}, {128, 32, 1, 0, 0}, {32, 64, 1, 0, 0}, {40, 40, 1, 0, 0}, {40, 40, 1, 0, 0}, {48, 48, 1, 0, 0}, {96, 64, 1, 0, 0},
                               {96, 48, 1, 0, 0}, {32, 32, 4, 0, 0}, {32, 16, 4, 0, 0}, {32, 24, 4, 0, 0}, {64, 64, 1, 1, 0}, {64, 48, 1, 1, 0}, {64, 128, 1, 1, 0},
                               {64, 64, 1, 1, 0}, {64, 96, 1, 1, 0}, {64, 64, 1, 1, 0}, {64, 48, 1, 1, 0}, {64, 80, 1, 1, 0}, {64, 96, 1, 1, 0}, {64, 48, 1, 1, 0}, {64, 64, 1, 1, 0}, {64, 272, 1, 2, 8}, {64, 144, 1, 2, 8}};
    const int ntab = (int)(sizeof(tab) / sizeof(tab[0]));
    // edge shapes first, then random ones
    for (int L : {1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 4096})
        for (int R : {16, 32, 40, 48, 64, 96, 128})
            for (int T = 1; T <= 24; ++T) check_tiling(L, R, T);
    for (int k = 0; k < rounds; ++k) check_tiling(rint_in(1, 5000), rint_in(2, 160), rint_in(1, 40));
    for (int k = 0; k < rounds; ++k) {
        const int O = rint_in(1, 300), world = rint_in(1, 64);
        int prev = 0, mn = 1 << 30, mx = 0;
        for (int r = 0; r < world; ++r) {
            int lo, hi;
            shard_range(O, world, r, &lo, &hi);
            CHECK(lo == prev && hi >= lo && hi <= O, "shard_range(%d,%d,%d) = [%d,%d)", O, world, r, lo, hi);
            mn = std::min(mn, hi - lo); mx = std::max(mx, hi - lo);
            prev = hi;
        }
        CHECK(prev == O && mx - mn <= 1, "shard_range(%d,%d): ends at %d, shares %d..%d", O, world, prev, mn, mx);
    }
    int nplans = 0;
    for (int k = 0; k < rounds; ++k) {
        PlanRequest q;
        const int kind = rint_in(0, 3);
        q.M = kind == 0 ? rint_in(1, 40) : (kind == 1 ? rint_in(1, 300) : rint_in(1, 2100));
        q.N = kind == 0 ? rint_in(1, 40) : (kind == 1 ? rint_in(1, 300) : rint_in(1, 2100));
        q.nimg = (rnd() & 3) ? rint_in(1, 12) : rint_in(1, 3000);
        q.ncu = (rnd() & 1) ? 256 : rint_in(0, 304);
        q.maxiter = (rnd() & 7) ? rint_in(0, 12000) : rint_in(0, 3);
        q.tile_iters = (rnd() & 1) ? 0 : rint_in(0, 40);
        q.variant = (rnd() & 1) ? 0 : rint_in(0, ntab + 1);
        q.chains = (rnd() & 1) ? 0 : rint_in(0, 5);
        Plan pl{0, 0, 0, 0, 0, 0};
        const int rc = plan_pdhg(q, tab, ntab, &pl);
        if (rc != PLAN_OK) {
            CHECK(rc == PLAN_E_VARIANT ? q.variant > ntab
                                       : (rc == PLAN_E_MIN_IMAGE ? (q.variant >= 1 && plan_too_small(tab[q.variant - 1], q.M, q.N))
                                                                 : (rc == PLAN_E_GRID ? true : true)),
                  "plan rc %d for M %d N %d variant %d", rc, q.M, q.N, q.variant);
            continue;
        }
        ++nplans;
        const PlanVariant& V = tab[pl.variant];
        CHECK(pl.variant >= 0 && pl.variant < ntab && pl.T >= 1, "plan variant %d T %d", pl.variant, pl.T);
        CHECK(!plan_too_small(V, q.M, q.N), "rows variant %d on a %dx%d image", pl.variant + 1, q.M, q.N);
        CHECK(V.tmax == 0 || pl.T <= V.tmax, "variant %d fuses %d > %d iterations", pl.variant + 1, pl.T, V.tmax);
        CHECK(q.M <= V.RI || V.RI - 2 * pl.T >= 1, "no core along i: M %d R %d T %d", q.M, V.RI, pl.T);
        CHECK(q.N <= V.RJ || V.RJ - 2 * pl.T >= 1, "no core along j: N %d R %d T %d", q.N, V.RJ, pl.T);
        CHECK(pl.nTi == tile_count(q.M, V.RI, pl.T) && pl.nTj == tile_count(q.N, V.RJ, pl.T) && pl.nTi >= 1 && pl.nTj >= 1, "tile counts %d x %d", pl.nTi, pl.nTj);
        CHECK((long)pl.grid == (long)pl.nTi * pl.nTj * q.nimg, "grid %d", pl.grid);
        CHECK(pl.chains >= 1 && pl.chains <= q.nimg, "chains %d of %d images", pl.chains, q.nimg);
        if (q.variant >= 1 && q.variant <= ntab) CHECK(pl.variant == q.variant - 1, "variant %d requested, %d planned", q.variant, pl.variant + 1);
        if (q.tile_iters >= 1) CHECK(pl.T <= q.tile_iters, "T %d above the requested %d", pl.T, q.tile_iters);
        check_tiling(q.M, V.RI, pl.T);
        check_tiling(q.N, V.RJ, pl.T);
        int prev = 0;   // the chains' image groups (build_graphs): lo = nimg * c / chains
        for (int c = 0; c < pl.chains; ++c) {
            const int lo = (int)(((long)q.nimg * c) / pl.chains), hi = (int)(((long)q.nimg * (c + 1)) / pl.chains);
            CHECK(lo == prev && hi > lo, "chain %d of %d: images [%d,%d)", c, pl.chains, lo, hi);
            prev = hi;
        }
        CHECK(prev == q.nimg, "chains end at image %d of %d", prev, q.nimg);
        if (q.maxiter > 0 && chain_out_of_phase(q.maxiter, pl.T, false)) {
            // the in-phase chain's launch l writes set l % 2; the out-of-phase chain starts by writing set 1 and has one launch more
            const int nl = (q.maxiter + pl.T - 1) / pl.T, h0 = pl.T / 2;
            const int nl1 = 1 + (q.maxiter - h0 + pl.T - 1) / pl.T;
            CHECK(h0 >= 1 && ((nl - 1) % 2) == ((1 + (nl1 - 1)) % 2), "out-of-phase chain ends in another set: maxiter %d T %d", q.maxiter, pl.T);
        }
    }
    printf("%s: %d tilings, %d shard distributions, %d plans checked, %d failures\n", bad ? "FAILED" : "all ok", rounds, rounds, nplans, bad);
    return bad ? 1 : 0;
}
