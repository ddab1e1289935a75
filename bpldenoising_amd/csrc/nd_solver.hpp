// nd_solver.hpp -- host side of the nested-dissection (multifrontal) Cholesky: tree (nd_symbolic.hpp) on the device,
// workspace for a group of images, and the level-by-level launch sequences of nd_kernels.hpp.
//
// One object serves one grid shape and stencil (TV: offsets 0, 1, M-1, M; sum of regularisers: seven diagonals) and
// up to `cap` images per call; the matrix is handed over as the diagonals the assemble kernels write (BandDiags
// convention: plane t holds A[c + off_t][c] at planes[t * tot + img * n + c]).  Replaces the sparse LU behind Julia's
// `\` at /root/reference/src/TVLearningFunctionVec.jl:131,248 and /root/reference/src/SumRegsLearningFunction.jl:324,394.
// Cost for a 1024 x 1024 image: 4e10 flop and 0.55 GB of factor (banded Cholesky: 1.1e12 flop, 8.6 GB).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "nd_kernels.hpp"
#include "nd_symbolic.hpp"

namespace bpltv {

// dynamic LDS of the split substitution kernels (nd_fwd_rows_kernel: pivot vector + 8 partial rows of 128; nd_bwd_cols_kernel:
// boundary vector + a reduction line)
inline size_t nd_rows_lds(int pmax) { return sizeof(double) * ((size_t)pmax + 8 * HB2_NB); }
inline size_t nd_cols_lds(int bmax) { return sizeof(double) * ((size_t)bmax + 64); }

struct NdSolver {
    NdTree T;
    struct Level {
        int n0 = 0, n1 = 0;
        bool small = true;
        int pmax = 0, bmax = 0, fmax = 0, MPmax = 0, bcmax = 0;   // bcmax: largest child boundary
        int fpmax = 0;                 // largest factor block (f p entries) of the level
        bool has_child = false;
        long long fac0 = 0, fac_len = 0, ws_len = 0;
        int wave = -1;                 // >= 0: instance of nd_front_wave_kernel (a wave per front) that holds every front of the level
        bool skinny = false;           // <= 16 pivots and <= 128 rows, with children: nd_front_skinny_kernel (only the pivot block column in LDS)
        bool skinny2 = false;          // large-regime level of fronts with <= 32 pivots and <= 256 rows: nd_front_skinny2_kernel (two block columns)
    };
    std::vector<Level> lv;
    bool built = false;
    int cap = 0;                       // images the workspace holds
    NdNodeDev* d_nodes = nullptr;
    int *d_pix = nullptr, *d_cmap = nullptr, *d_inv = nullptr;
    int4* d_orig = nullptr;
    double *fac = nullptr, *ws[2] = {nullptr, nullptr}, *yv = nullptr, *uv = nullptr;
    bool lu = false;                   // LU variant (nd_kernels.hpp): the fronts' upper triangles, transposed, beside the lower
    double *facU = nullptr, *wsU[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    std::string err;
    double factor_flop = 0.0;          // per image (multiply-add = 2)
    bool wave_fronts = true;           // Cholesky, fronts of <= 64 rows and <= 32 pivots: nd_front_wave_kernel (false: tools, A/B timing)
    bool skinny_fronts = true;         // Cholesky, levels of fronts with <= 16 pivots that are too tall for the wave kernel: nd_front_skinny_kernel
    int skinny_min = 0, skinny2_min = 256; // ... only on levels of at least so many (front, image) pairs: one launch with a long serial chain per
                                       // front against five short launches pays once the level fills the chip (bpltv.hip sets the measured thresholds)
    bool staged_solve = true;          // small levels with f p <= NDS_STAGE: substitutions with the factor block staged in LDS

    // the instances of nd_front_wave_kernel<F, P>, smallest first within a pivot class
    struct WaveInst { int F, P; void (*fn)(NdArgs); };
    static const WaveInst* wave_insts(int& n) {
        static const WaveInst tab[] = {
            {40, 8, nd_front_wave_kernel<40, 8>},   {56, 8, nd_front_wave_kernel<56, 8>},   {64, 8, nd_front_wave_kernel<64, 8>},
            {24, 16, nd_front_wave_kernel<24, 16>}, {32, 16, nd_front_wave_kernel<32, 16>}, {48, 16, nd_front_wave_kernel<48, 16>},
            {64, 16, nd_front_wave_kernel<64, 16>}, {48, 32, nd_front_wave_kernel<48, 32>}, {64, 32, nd_front_wave_kernel<64, 32>},
        };
        n = (int)(sizeof(tab) / sizeof(tab[0]));
        return tab;
    }
    static int pick_wave(int fmax, int pmax) {
        int n;
        const WaveInst* t = wave_insts(n);
        for (int i = 0; i < n; ++i)
            if (t[i].F >= fmax && t[i].P >= pmax) return i;
        return -1;
    }

#define NDCHK(call)                                                                               \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            err = std::string(#call) + " failed: " + hipGetErrorString(e_);                       \
            return e_ == hipErrorOutOfMemory ? 5 : 2;                                             \
        }                                                                                         \
    } while (0)

    // bytes of device memory per image of the workspace (factor, two update-matrix workspaces, vectors)
    size_t bytes_per_image() const {
        return sizeof(double) * (size_t)((lu ? 2 : 1) * (T.fac_doubles + T.ws_doubles[0] + T.ws_doubles[1]) + T.uv_doubles + T.n);
    }
    size_t index_bytes() const {
        return sizeof(NdNodeDev) * T.nodes.size() + sizeof(int) * (T.pix.size() + T.cmap.size()) + sizeof(int4) * T.orig.size();
    }

    // symbolic phase + upload of the index arrays.  Returns 0, 2 (HIP error) or 5 (out of memory).
    // lu_variant: block LU without pivoting for a structurally symmetric, numerically non-symmetric matrix.
    int build(int M, int N, const NdStencil& st, int leaf_pix = 0, bool lu_variant = false) {
        release();
        lu = lu_variant;
        T = nd_build(M, N, st, leaf_pix > 0 ? leaf_pix : 32);
        if (!T.ok) { err = "nested dissection, symbolic phase: " + T.err; return 6; }
        factor_flop = 2.0 * T.flops();
        const int L = T.levels();
        lv.assign(L, Level());
        for (int l = 0; l < L; ++l) {
            Level& a = lv[l];
            a.n0 = T.lvl_start[l]; a.n1 = T.lvl_start[l + 1];
            a.fac0 = T.nodes[a.n0].fac_off;
            for (int q = a.n0; q < a.n1; ++q) {
                const NdNode& v = T.nodes[q];
                a.pmax = std::max(a.pmax, v.p); a.bmax = std::max(a.bmax, v.b); a.fmax = std::max(a.fmax, v.p + v.b);
                a.fpmax = std::max(a.fpmax, (v.p + v.b) * v.p);
                const int MP = nd_up16(nd_up16(v.p) + v.b);
                a.MPmax = std::max(a.MPmax, MP);
                a.fac_len = v.fac_off + (long long)(v.p + v.b) * v.p - a.fac0;
                a.ws_len = v.u_off + (long long)v.b * v.b;
                for (int ci = 0; ci < 2; ++ci)
                    if (v.child[ci] >= 0) { a.has_child = true; a.bcmax = std::max(a.bcmax, T.nodes[v.child[ci]].b); }
            }
            // in LDS only while the pivot block is at most three block columns wide: wave 0 factors them one after the
            // other there; wider pivot blocks (the root of a 128-wide image) go through bcr_potrf_lds_body (large regime)
            a.small = a.MPmax <= 128 && a.pmax <= 48;
            if (a.small && !lu) a.wave = pick_wave(a.fmax, a.pmax);
            a.skinny = a.small && !lu && a.wave < 0 && a.has_child && a.pmax <= 16 && 16 + nd_up16(a.bmax) <= 128;
            a.skinny2 = !a.small && !lu && a.has_child && a.pmax <= 32 && 32 + nd_up16(a.bmax) <= 256;
        }
        std::vector<NdNodeDev> nd(T.nodes.size());
        for (size_t q = 0; q < T.nodes.size(); ++q) {
            const NdNode& v = T.nodes[q];
            nd[q] = NdNodeDev{v.p, v.b, v.piv_off, v.cmap_off, v.orig_off, v.orig_cnt, v.child[0], v.child[1], v.fac_off, v.u_off, v.uv_off, -1, 0};
        }
        // parent -> child maps of the large-regime fronts (gather-form assembly, nd_gather_kernel)
        std::vector<int> inv;
        for (int l = 0; l < L; ++l) {
            if (lv[l].small && !lv[l].skinny) continue;
            for (int q = lv[l].n0; q < lv[l].n1; ++q) {
                const NdNode& v = T.nodes[q];
                if (v.child[0] < 0 && v.child[1] < 0) continue;
                const int f = v.p + v.b;
                nd[q].inv_off = (int)inv.size();
                inv.resize(inv.size() + 2 * (size_t)f, -1);
                for (int ci = 0; ci < 2; ++ci) {
                    if (v.child[ci] < 0) continue;
                    const NdNode& ch = T.nodes[v.child[ci]];
                    int* dst = inv.data() + nd[q].inv_off + (size_t)ci * f;
                    for (int k = 0; k < ch.b; ++k) dst[T.cmap[ch.cmap_off + k]] = k;
                }
            }
        }
        static_assert(sizeof(NdOrig) == sizeof(int4), "NdOrig is uploaded as int4");
        NDCHK(hipMalloc((void**)&d_nodes, nd.size() * sizeof(NdNodeDev)));
        NDCHK(hipMalloc((void**)&d_pix, std::max<size_t>(1, T.pix.size()) * sizeof(int)));
        NDCHK(hipMalloc((void**)&d_cmap, std::max<size_t>(1, T.cmap.size()) * sizeof(int)));
        NDCHK(hipMalloc((void**)&d_orig, std::max<size_t>(1, T.orig.size()) * sizeof(int4)));
        NDCHK(hipMalloc((void**)&d_inv, std::max<size_t>(1, inv.size()) * sizeof(int)));
        if (!inv.empty()) NDCHK(hipMemcpy(d_inv, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice));
        NDCHK(hipMemcpy(d_nodes, nd.data(), nd.size() * sizeof(NdNodeDev), hipMemcpyHostToDevice));
        NDCHK(hipMemcpy(d_pix, T.pix.data(), T.pix.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!T.cmap.empty()) NDCHK(hipMemcpy(d_cmap, T.cmap.data(), T.cmap.size() * sizeof(int), hipMemcpyHostToDevice));
        NDCHK(hipMemcpy(d_orig, T.orig.data(), T.orig.size() * sizeof(int4), hipMemcpyHostToDevice));
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_front_small_kernel<true, NDS_T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)nd_small_lds(128)));
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_front_skinny2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)nd_skinny2_lds(256)));
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_potrf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bcr_potrf_lds(HB2_NB)));
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_front_small_lu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)nd_small_lds(128)));
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_getri_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)nd_getri_lds()));
        if (nd_large_lds(T.max_f) > 160 * 1024) { err = "front too large for the substitution kernels"; return 6; }
        {   // the split substitutions of fronts with many boundary rows stage a pivot / boundary vector in dynamic LDS
            int bmax_all = 0;
            for (const NdNode& v : T.nodes) bmax_all = std::max(bmax_all, v.b);
            const size_t rows_lds = nd_rows_lds(T.max_p), cols_lds = nd_cols_lds(bmax_all);
            if (rows_lds > 160 * 1024 || cols_lds > 160 * 1024) { err = "pivot block or boundary too long for the split substitution kernels"; return 6; }
            NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_fwd_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds));
            NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_bwd_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cols_lds));
        }
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_fwd_large_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)nd_large_lds(T.max_f)));
        NDCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_bwd_large_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)nd_large_lds(T.max_f)));
        built = true;
        return 0;
    }

    // workspace for `nimg` images per call.  Nothing stays allocated on failure.
    int alloc(int nimg, hipStream_t st) {
        stream = st;
        if (nimg <= cap) return 0;
        free_ws();
        const int rc = alloc_ws(nimg);
        if (rc) free_ws();
        return rc;
    }
    int alloc_ws(int nimg) {
        NDCHK(hipMalloc((void**)&fac, (size_t)nimg * T.fac_doubles * sizeof(double)));
        for (int s = 0; s < 2; ++s) NDCHK(hipMalloc((void**)&ws[s], std::max<size_t>(1, (size_t)nimg * T.ws_doubles[s]) * sizeof(double)));
        NDCHK(hipMalloc((void**)&yv, (size_t)nimg * T.n * sizeof(double)));
        NDCHK(hipMalloc((void**)&uv, std::max<size_t>(1, (size_t)nimg * T.uv_doubles) * sizeof(double)));
        if (lu) {
            NDCHK(hipMalloc((void**)&facU, (size_t)nimg * T.fac_doubles * sizeof(double)));
            for (int s = 0; s < 2; ++s) NDCHK(hipMalloc((void**)&wsU[s], std::max<size_t>(1, (size_t)nimg * T.ws_doubles[s]) * sizeof(double)));
        }
        cap = nimg;
        return 0;
    }
    void free_ws() {
        for (void* p : {(void*)fac, (void*)ws[0], (void*)ws[1], (void*)yv, (void*)uv, (void*)facU, (void*)wsU[0], (void*)wsU[1]})
            if (p) (void)hipFree(p);
        fac = ws[0] = ws[1] = yv = uv = facU = wsU[0] = wsU[1] = nullptr;
        cap = 0;
    }
    void release() {
        free_ws();
        for (void* p : {(void*)d_nodes, (void*)d_pix, (void*)d_cmap, (void*)d_orig, (void*)d_inv})
            if (p) (void)hipFree(p);
        d_nodes = nullptr; d_pix = d_cmap = d_inv = nullptr; d_orig = nullptr;
        built = false;
    }

    // Factor the matrices of `nimg` images.  planes: plane 0 of the first image of the group; tot: doubles per plane
    // (of the whole batch); d_fail[img]: node + 1 of the first non-positive pivot (not cleared here).
    int factor(const double* planes, size_t tot, int nimg, int* d_fail) {
        if (!built || nimg > cap) { err = "nd solver: not built or workspace too small"; return 2; }
        NdArgs A;
        A.nodes = d_nodes; A.pix = d_pix; A.cmap = d_cmap; A.orig = d_orig; A.inv = d_inv;
        if (lu) { err = "nd solver: built for LU, use factor_lu"; return 2; }
        A.planes = planes; A.planes2 = planes; A.tot = tot; A.n = T.n;
        A.fac = fac; A.fac2 = fac; A.fac_stride = T.fac_doubles; A.fail = d_fail;
        A.ws_mine2 = nullptr; A.ws_child2 = nullptr;
        const int L = T.levels();
        for (int l = L - 1; l >= 0; --l) {
            const Level& a = lv[l];
            A.ws_mine = ws[l & 1]; A.ws_mine_stride = T.ws_doubles[l & 1];
            A.ws_child = ws[(l + 1) & 1]; A.ws_child_stride = T.ws_doubles[(l + 1) & 1];
            const int cnt = a.n1 - a.n0;
            if (a.small) {
                A.node0 = a.n0;
                if (skinny_fronts && a.skinny && (long long)cnt * nimg >= skinny_min) {
                    hipLaunchKernelGGL(nd_front_skinny_kernel, dim3(cnt, nimg), dim3(256), nd_skinny_lds(16 + nd_up16(a.bmax)), stream, A);
                } else if (wave_fronts && a.wave >= 0) {
                    int n;
                    hipLaunchKernelGGL(wave_insts(n)[a.wave].fn, dim3(cnt, nimg), dim3(64), 0, stream, A);
                } else if (a.MPmax <= 48) hipLaunchKernelGGL((nd_front_small_kernel<false, 128>), dim3(cnt, nimg), dim3(128), nd_small_lds(a.MPmax), stream, A);
                else if (a.MPmax <= 64) hipLaunchKernelGGL((nd_front_small_kernel<false, NDS_T>), dim3(cnt, nimg), dim3(NDS_T), nd_small_lds(a.MPmax), stream, A);
                else hipLaunchKernelGGL((nd_front_small_kernel<true, NDS_T>), dim3(cnt, nimg), dim3(NDS_T), nd_small_lds(a.MPmax), stream, A);
                continue;
            }
            if (skinny_fronts && a.skinny2 && (long long)cnt * nimg >= skinny2_min) {
                A.node0 = a.n0;
                hipLaunchKernelGGL(nd_front_skinny2_kernel, dim3(cnt, nimg), dim3(256), nd_skinny2_lds(32 + nd_up16(a.bmax)), stream, A);
                continue;
            }
            for (int q0 = a.n0; q0 < a.n1; q0 += 32768) {     // grid.y <= 65535
                const int qn = std::min(32768, a.n1 - q0);
                A.node0 = q0;
                hipLaunchKernelGGL(nd_gather_kernel, dim3(std::max(1, std::min((a.fmax + 4 * NDG_C - 1) / (4 * NDG_C), 128)), qn, nimg), dim3(256), 0, stream, A);
                hipLaunchKernelGGL(nd_orig_kernel, dim3(qn, nimg), dim3(256), 0, stream, A);
                const int npan = (a.pmax + HB2_NB - 1) / HB2_NB;
                for (int k = 0; k < npan; ++k) {
                    hipLaunchKernelGGL(nd_potrf_kernel, dim3(qn, nimg), dim3(BCR_PT), bcr_potrf_lds(std::min(HB2_NB, nd_up16(a.pmax - HB2_NB * k))), stream, A, k);
                    const int below = a.fmax - HB2_NB * k;
                    const int ntile = (below + 63) / 64;
                    if (ntile > 0) hipLaunchKernelGGL(nd_trsm_kernel, dim3(ntile, qn, nimg), dim3(BG_T), 0, stream, A, k, 0);
                    if (a.pmax > HB2_NB * (k + 1)) {
                        const int ntr = (a.fmax - HB2_NB * (k + 1) + 63) / 64;
                        hipLaunchKernelGGL(nd_syrk_kernel, dim3(ntr * (ntr + 1) / 2, qn, nimg), dim3(BG_T), 0, stream, A, k);
                    }
                }
                if (a.bmax > 0) {
                    const int nt = (a.bmax + 63) / 64;
                    hipLaunchKernelGGL(nd_schur_kernel, dim3(nt * (nt + 1) / 2, qn, nimg), dim3(BG_T), 0, stream, A);
                }
            }
        }
        NDCHK(hipGetLastError());
        return 0;
    }

    // LU variant.  planesL: diagonals of the lower triangle (A[c + off][c] at pixel c, plane 0 = main diagonal), planesU: of the
    // upper triangle (A[c][c + off] at pixel c; its plane 0 is not read).  d_fail[img]: node + 1 of the first broken pivot.
    int factor_lu(const double* planesL, const double* planesU, size_t tot, int nimg, int* d_fail) {
        if (!built || !lu || nimg > cap) { err = "nd solver: not built for LU or workspace too small"; return 2; }
        const int L = T.levels();
        for (int l = L - 1; l >= 0; --l) {
            const Level& a = lv[l];
            // side 0 writes the lower triangles (FL, update matrices' lower parts), side 1 the transposed upper ones
            auto args = [&](int side) {
                NdArgs A;
                A.nodes = d_nodes; A.pix = d_pix; A.cmap = d_cmap; A.orig = d_orig; A.inv = d_inv;
                A.planes = side ? planesU : planesL; A.planes2 = side ? planesL : planesU; A.tot = tot; A.n = T.n;
                A.fac = side ? facU : fac; A.fac2 = side ? fac : facU; A.fac_stride = T.fac_doubles; A.fail = d_fail;
                double* const* wm = side ? wsU : ws;
                double* const* wo = side ? ws : wsU;
                A.ws_mine = wm[l & 1]; A.ws_mine_stride = T.ws_doubles[l & 1];
                A.ws_child = wm[(l + 1) & 1]; A.ws_child_stride = T.ws_doubles[(l + 1) & 1];
                A.ws_mine2 = wo[l & 1]; A.ws_child2 = wo[(l + 1) & 1];
                A.node0 = a.n0;
                return A;
            };
            const int cnt = a.n1 - a.n0;
            if (a.small) {
                hipLaunchKernelGGL(nd_front_small_lu_kernel, dim3(cnt, nimg), dim3(NDS_T), nd_small_lds(a.MPmax), stream, args(0));
                continue;
            }
            for (int q0 = a.n0; q0 < a.n1; q0 += 32768) {
                const int qn = std::min(32768, a.n1 - q0);
                NdArgs AL = args(0), AU = args(1);
                AL.node0 = AU.node0 = q0;
                const dim3 gg(std::max(1, std::min((a.fmax + 4 * NDG_C - 1) / (4 * NDG_C), 128)), qn, nimg);
                hipLaunchKernelGGL(nd_gather_kernel, gg, dim3(256), 0, stream, AL);
                hipLaunchKernelGGL(nd_gather_kernel, gg, dim3(256), 0, stream, AU);
                hipLaunchKernelGGL(nd_orig_kernel, dim3(qn, nimg), dim3(256), 0, stream, AL);
                hipLaunchKernelGGL(nd_orig_kernel, dim3(qn, nimg), dim3(256), 0, stream, AU);
                const int npan = (a.pmax + HB2_NB - 1) / HB2_NB;
                for (int k = 0; k < npan; ++k) {
                    hipLaunchKernelGGL(nd_getri_kernel, dim3(qn, nimg), dim3(NDG_T), nd_getri_lds(), stream, AL, k);
                    const int below = a.fmax - HB2_NB * k;
                    const int ntile = (below + 63) / 64;
                    if (ntile > 0) hipLaunchKernelGGL(nd_trsm_kernel, dim3(ntile, qn, nimg), dim3(BG_T), 0, stream, AL, k, 1);
                    if (a.pmax > HB2_NB * (k + 1)) {
                        const int ntr = (a.fmax - HB2_NB * (k + 1) + 63) / 64;
                        hipLaunchKernelGGL(nd_syrk_kernel, dim3(ntr * (ntr + 1) / 2, qn, nimg), dim3(BG_T), 0, stream, AL, k);
                        hipLaunchKernelGGL(nd_syrk_kernel, dim3(ntr * (ntr + 1) / 2, qn, nimg), dim3(BG_T), 0, stream, AU, k);
                    }
                }
                if (a.bmax > 0) {
                    const int nt = (a.bmax + 63) / 64;
                    hipLaunchKernelGGL(nd_schur_kernel, dim3(nt * (nt + 1) / 2, qn, nimg), dim3(BG_T), 0, stream, AL);
                    hipLaunchKernelGGL(nd_schur_kernel, dim3(nt * (nt + 1) / 2, qn, nimg), dim3(BG_T), 0, stream, AU);
                }
            }
        }
        NDCHK(hipGetLastError());
        return 0;
    }

    // vec <- A^-1 vec for `nimg` images ([nimg][n]); acc += solution when not null.
    int solve(double* vec, double* acc, int nimg) {
        if (!built || nimg > cap) { err = "nd solver: not built or workspace too small"; return 2; }
        NdSolveArgs S;
        S.nodes = d_nodes; S.pix = d_pix; S.cmap = d_cmap; S.fac = fac; S.fac2 = lu ? facU : fac; S.lu = lu ? 1 : 0; S.fac_stride = T.fac_doubles;
        S.vec = vec; S.y = yv; S.uv = uv; S.uv_stride = T.uv_doubles; S.acc = nullptr; S.n = T.n;
        const int L = T.levels();
        for (int l = L - 1; l >= 0; --l) {
            const Level& a = lv[l];
            for (int q0 = a.n0; q0 < a.n1; q0 += (a.small ? 1 << 20 : 32768)) {
                S.node0 = q0;
                const int qn = std::min(a.small ? 1 << 20 : 32768, a.n1 - q0);
                if (a.small && staged_solve && a.fpmax <= NDS_STAGE) hipLaunchKernelGGL(nd_fwd_staged_kernel, dim3(qn, nimg), dim3(64), sizeof(double) * a.fpmax, stream, S);
                else if (a.small) hipLaunchKernelGGL(nd_fwd_small_kernel, dim3(qn, nimg), dim3(64), 0, stream, S);
                else {
                    const int split = a.bmax >= 256 ? 1 : 0;   // many boundary rows: L21 y by row blocks, behind the pivot sweep
                    hipLaunchKernelGGL(nd_fwd_large_kernel, dim3(qn, nimg), dim3(NDL_T), nd_large_lds(a.fmax), stream, S, split);
                    if (split)
                        hipLaunchKernelGGL(nd_fwd_rows_kernel, dim3((a.bmax + HB2_NB - 1) / HB2_NB, qn, nimg), dim3(NDL_T),
                                           nd_rows_lds(a.pmax), stream, S);
                }
            }
        }
        S.acc = acc;
        for (int l = 0; l < L; ++l) {
            const Level& a = lv[l];
            for (int q0 = a.n0; q0 < a.n1; q0 += (a.small ? 1 << 20 : 32768)) {
                S.node0 = q0;
                const int qn = std::min(a.small ? 1 << 20 : 32768, a.n1 - q0);
                if (a.small && staged_solve && a.fpmax <= NDS_STAGE) hipLaunchKernelGGL(nd_bwd_staged_kernel, dim3(qn, nimg), dim3(64), sizeof(double) * a.fpmax, stream, S);
                else if (a.small) hipLaunchKernelGGL(nd_bwd_small_kernel, dim3(qn, nimg), dim3(64), 0, stream, S);
                else {
                    const int split = a.bmax >= 256 ? 1 : 0;
                    if (split)
                        hipLaunchKernelGGL(nd_bwd_cols_kernel, dim3((a.pmax + HB2_NB - 1) / HB2_NB, qn, nimg), dim3(NDL_T),
                                           nd_cols_lds(a.bmax), stream, S);
                    hipLaunchKernelGGL(nd_bwd_large_kernel, dim3(qn, nimg), dim3(NDL_T), nd_large_lds(a.fmax), stream, S, split);
                }
            }
        }
        NDCHK(hipGetLastError());
        return 0;
    }
#undef NDCHK
};

}  // namespace bpltv
