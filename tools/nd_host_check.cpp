// nd_host_check.cpp -- host check of csrc/nd_symbolic.hpp: on small grids the multifrontal factorisation that
// follows the tree (tools/nd_ref.hpp) must solve random SPD stencil systems to rounding, for both stencils, odd
// shapes and several leaf sizes.  Build: g++ -O2 -std=c++17 tools/nd_host_check.cpp -o tools/_bin/nd_host_check
// (tests/test_nd_host.py builds and runs it; no GPU).  `nd_host_check v 1024 [leaf]` prints the structure of a
// large grid (fronts per level, fill, flop).
#include <cstdio>
#include <cstdlib>
#include <string>
#include "nd_ref.hpp"

int main(int argc, char** argv) {
    if (argc == 4 && std::string(argv[1]) == "bytes") {   // workspace per image of the device solver for an M x N grid
        const int M = atoi(argv[2]), N = atoi(argv[3]);
        for (int sr = 0; sr < 2; ++sr) {
            const bpltv::NdTree T = bpltv::nd_build(M, N, sr ? bpltv::nd_stencil_sr() : bpltv::nd_stencil_tv(), getenv("BPLTV_ND_LEAF") ? atoi(getenv("BPLTV_ND_LEAF")) : 32);
            printf("bytes_per_image %s %lld\n", sr ? "sr" : "tv", (long long)(8 * (T.fac_doubles + T.ws_doubles[0] + T.ws_doubles[1] + T.uv_doubles + T.n)));
        }
        return 0;
    }
    const int shapes[][2] = {{1, 1}, {3, 1}, {1, 7}, {2, 2}, {5, 4}, {16, 16}, {33, 17}, {40, 64}, {7, 130}, {96, 50}, {128, 128}};
    int bad = 0;
    for (int sr = 0; sr < 2; ++sr)
        for (const auto& sh : shapes)
            for (int leaf : {1, 8, 32, 100}) {
                const int M = sh[0], N = sh[1];
                if ((long)M * N > 6000 && leaf != 32) continue;
                const bpltv::NdStencil st = sr ? bpltv::nd_stencil_sr() : bpltv::nd_stencil_tv();
                const bpltv::NdTree T = bpltv::nd_build(M, N, st, leaf);
                std::vector<int> cnt(T.n, 0);   // every pixel is a pivot exactly once
                for (const auto& v : T.nodes)
                    for (int k = 0; k < v.p; ++k) cnt[T.pix[v.piv_off + k]]++;
                for (int g = 0; g < T.n; ++g)
                    if (cnt[g] != 1) {
                        printf("FAIL %dx%d sr=%d leaf=%d: pixel %d is a pivot %d times\n", M, N, sr, leaf, g, cnt[g]);
                        ++bad;
                        break;
                    }
                if (T.nodes[0].b != 0) { printf("FAIL root has a boundary\n"); ++bad; }
                const auto P = ndref::random_spd(T, 7 + M + 3 * N + leaf, (M * N > 50) ? 1e6 : 0.0);
                const ndref::Factor F = ndref::factor(T, P);
                if (F.fail) {
                    printf("FAIL %dx%d sr=%d leaf=%d: non-positive pivot in node %d\n", M, N, sr, leaf, F.fail - 1);
                    ++bad;
                    continue;
                }
                std::vector<double> xt(T.n), rhs, x;
                for (int g = 0; g < T.n; ++g) xt[g] = std::sin(0.37 * g) + 0.1 * (g % 7);
                ndref::matvec(T, P, xt, rhs);
                x = rhs;
                ndref::solve(T, F, x);
                double err = 0.0, nrm = 0.0;
                for (int g = 0; g < T.n; ++g) { err = std::fmax(err, std::fabs(x[g] - xt[g])); nrm = std::fmax(nrm, std::fabs(xt[g])); }
                const bool ok = err <= 1e-9 * nrm;
                if (!ok) ++bad;
                if (!ok || argc > 1)
                    printf("%s %4dx%-4d %s leaf %3d: nodes %6zu levels %2d max front %4d fill %.3g flop %.3g  err %.2e\n", ok ? "ok  " : "FAIL", M,
                           N, sr ? "sr" : "tv", leaf, T.nodes.size(), T.levels(), T.max_f, (double)T.fac_doubles, T.flops(), err / nrm);
            }
    if (argc > 2) {
        const int M = atoi(argv[2]), leaf = argc > 3 ? atoi(argv[3]) : 32;
        for (int sr = 0; sr < 2; ++sr) {
            const bpltv::NdTree T = bpltv::nd_build(M, M, sr ? bpltv::nd_stencil_sr() : bpltv::nd_stencil_tv(), leaf);
            printf("%dx%d %s leaf %d: nodes %zu levels %d max front %d max p %d factor %.1f MB  workspace %.1f + %.1f MB  flop %.3g  index %.1f MB\n",
                   M, M, sr ? "sr" : "tv", leaf, T.nodes.size(), T.levels(), T.max_f, T.max_p, T.fac_doubles * 8e-6, T.ws_doubles[0] * 8e-6,
                   T.ws_doubles[1] * 8e-6, T.flops(), (T.pix.size() + T.cmap.size() + 4 * T.orig.size()) * 4e-6);
            for (int l = 0; l < T.levels(); ++l) {
                long sp = 0, sb = 0, mf = 0;
                double fl = 0, fill = 0;
                for (int q = T.lvl_start[l]; q < T.lvl_start[l + 1]; ++q) {
                    const auto& v = T.nodes[q];
                    sp += v.p; sb += v.b; mf = std::max<long>(mf, v.p + v.b);
                    fl += (double)v.p * v.p * v.p / 3 + (double)v.p * v.p * v.b + (double)v.p * v.b * v.b;
                    fill += (double)(v.p + v.b) * v.p;
                }
                const int cntl = T.lvl_start[l + 1] - T.lvl_start[l];
                printf("  level %2d: %6d fronts, mean p %6.1f b %6.1f, max f %5ld, flop %.3g, factor %.1f MB\n", l, cntl, (double)sp / cntl,
                       (double)sb / cntl, mf, fl, fill * 8e-6);
            }
        }
    }
    printf(bad ? "nd_host_check: %d FAILED\n" : "nd_host_check: all ok\n", bad);
    return bad ? 1 : 0;
}
