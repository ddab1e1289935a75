// hb_band_solver.hpp -- host side of the HBM-resident banded Cholesky (kernels: adjoint_hbm_kernels.hpp).
//
// One object = workspace + launch sequences for O symmetric positive definite band matrices of order n and
// bandwidth bw given as a few diagonals (BandDiags): the reduced adjoint systems of the TV model for images wider
// than the LDS window (bw = M) and of the sum-of-regularisers model (bw = 2M: the centred stencil couples columns
// j and j + 2).  Replaces the sparse LU behind Julia's `\` at /root/reference/src/TVLearningFunctionVec.jl:131,248
// and /root/reference/src/SumRegsLearningFunction.jl:164,250,324,394.
//
// Twisted (two-sided) factorisation: the dependent chain of a banded Cholesky is one 128-column panel after the
// other (diagonal block -> triangular solve -> first update tiles -> next diagonal block, ~80-115 us each on
// MI355X however many CUs are idle), 8192 panels for a 1024 x 1024 image.  Eliminating from both ends at once
// -- top columns in natural order, bottom columns in reversed order, as 2 O independent problems in the same
// launches -- halves that depth for the factorisation and for every substitution at no extra arithmetic; the
// two Schur complements meet in a dense middle block of ~bw columns that is factored last by the same kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "adjoint_hbm_kernels.hpp"

namespace bpltv {

__global__ void hb_fail_merge_kernel(const int* __restrict__ side, const int* __restrict__ mid, int sides, int O,
                                     int* __restrict__ out) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= O) return;
    int f = 0;
    for (int s = 0; s < sides; ++s)
        if (f == 0 && side[img * sides + s] != 0) f = side[img * sides + s];
    if (f == 0 && mid && mid[img] != 0) f = mid[img];
    if (f != 0 && out[img] == 0) out[img] = f;
}

struct HbBandSolver {
    // shape
    int bw = 0, n = 0, O = 0;
    bool twisted = false;
    int sides = 1;        // problems per image in the side array
    int m = 0, nm = 0;    // twisted: eliminated columns per side, middle columns
    int np = 0;           // rows of a side problem (m + bw; n when not twisted)
    // device memory
    double* band = nullptr;      // [O*sides][np][bw+1]
    double* mid = nullptr;       // [O][nm][bw+1]
    double* buf = nullptr;       // side problems: Linv | LinvT | P x2
    double* bufm = nullptr;      // middle problems
    double* vec = nullptr;       // twisted solve: vs | ys | xs ([O*2][np] each) | vm | ym | xm ([O][nm] each)
    int* fail = nullptr;         // [O*sides + O]
    hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // C -> T, T -> U2, prefill
    hipEvent_t ev1[2] = {nullptr, nullptr};           // U1 of even / odd panels
    hipEvent_t ev2 = nullptr;                         // U2
    bool prefilled = false;      // the band has been zero-filled ahead of factor() (prefill_async)
    bool value_sync = false, sync_err = false;
    unsigned* sig[4] = {nullptr, nullptr, nullptr, nullptr};   // counters in signal memory (value_sync)
    unsigned seq[4] = {0, 0, 0, 0};
    int rw_force = 0;            // 32 | 128: rows per tile workgroup of the substitutions (0: by size)
    bool single_stream = false;  // profiling aid: rocprofv3 --pmc cannot follow two streams
    // set before alloc() by the owner (bpltv_set_option "hb_sync" / "hb_single_stream" / "hb_rw"; the unit tools read
    // their own environment): 0 = automatic
    int opt_sync = 0, opt_single_stream = 0, opt_rw = 0;
    void options_from_env() {   // tools/ only (lu_unit, nd_unit): BPLTV_HB_SYNC=event|value, BPLTV_HB_SINGLE_STREAM=1, BPLTV_HB_RW
        const char* e4 = getenv("BPLTV_HB_SYNC");
        opt_sync = e4 ? (e4[0] == 'e' ? 1 : (e4[0] == 'v' ? 2 : 0)) : 0;
        const char* e1 = getenv("BPLTV_HB_SINGLE_STREAM");
        opt_single_stream = (e1 && e1[0] == '1') ? 1 : 0;
        const char* e6 = getenv("BPLTV_HB_RW");
        opt_rw = e6 ? atoi(e6) : 0;
    }
    std::string err;

    struct Bufs {
        int npanel;
        double *Linv, *LinvT, *P;
    };
    static size_t bufs_doubles(int nprob, int ncol, int bw) {
        const size_t npanel = (size_t)(ncol + HB2_NB - 1) / HB2_NB, bwp = (size_t)(bw + 63) / 64 * 64;
        return (size_t)nprob * (2 * npanel * HB2_NB * HB2_NB + 2 * bwp * HB2_NB);
    }
    static Bufs carve(double* base, int nprob, int ncol) {
        Bufs b;
        b.npanel = (ncol + HB2_NB - 1) / HB2_NB;
        const size_t blk = (size_t)nprob * HB2_NB * HB2_NB;
        b.Linv = base;
        b.LinvT = base + blk * b.npanel;
        b.P = b.LinvT + blk * b.npanel;     // two buffers of nprob * bwp * 128 doubles (panel parity)
        return b;
    }

    size_t bytes_needed(int bw_, int n_, int O_) const {
        Shape s = shape(bw_, n_);
        const size_t W = (size_t)bw_ + 1;
        size_t d = (size_t)O_ * s.sides * s.np * W + bufs_doubles(O_ * s.sides, s.sides == 2 ? s.m : n_, bw_);
        if (s.sides == 2) d += (size_t)O_ * s.nm * W + bufs_doubles(O_, s.nm, bw_) + (size_t)O_ * (6 * (size_t)s.np + 3 * (size_t)s.nm);
        return d * sizeof(double);
    }

    struct Shape { int sides, m, nm, np; };
    static Shape shape(int bw_, int n_) {
        Shape s;
        const long half = ((long)n_ - bw_) / 2;
        s.m = (int)(half > 0 ? (half / HB2_NB) * HB2_NB : 0);
        // worth it from a few panels per side; the middle must hold both trailing windows: nm >= bw
        if (s.m >= 4 * HB2_NB) {
            s.sides = 2;
            s.nm = n_ - 2 * s.m;
            s.np = s.m + bw_;
        } else {
            s.sides = 1; s.m = 0; s.nm = 0; s.np = n_;
        }
        return s;
    }

#define HBCHK(call)                                                                               \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            err = std::string(#call) + " failed: " + hipGetErrorString(e_);                       \
            return e_ == hipErrorOutOfMemory ? 5 : 2;                                             \
        }                                                                                         \
    } while (0)

    // returns 0, 2 (HIP error) or 5 (out of memory); `err` holds the message.  Nothing stays allocated on failure.
    int alloc(int bw_, int n_, int O_, hipStream_t st, bool allow_twist = true) {
        const int rc = alloc_impl(bw_, n_, O_, st, allow_twist);
        if (rc) release();
        return rc;
    }
    int alloc_impl(int bw_, int n_, int O_, hipStream_t st, bool allow_twist) {
        bw = bw_; n = n_; O = O_; stream = st;
        Shape s = allow_twist ? shape(bw, n) : Shape{1, 0, 0, n_};
        twisted = s.sides == 2; sides = s.sides; m = s.m; nm = s.nm; np = s.np;
        const size_t W = (size_t)bw + 1;
        single_stream = opt_single_stream != 0;
        rw_force = (opt_rw == 32 || opt_rw == 128) ? opt_rw : 0;
        {   // cross-stream dependencies by stream memory operations where the device has them (opt_sync = 1: HIP events
            // instead; rocprofv3 needs that -- it stalls every stream memory operation)
            int dev = 0, can = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) can = 0;
            value_sync = can != 0 && opt_sync != 1;
            if (value_sync)
                for (auto& q : sig) {
                    if (hipExtMallocWithFlags((void**)&q, 8, hipMallocSignalMemory) != hipSuccess || hipMemset(q, 0, 8) != hipSuccess) {
                        (void)hipGetLastError();
                        value_sync = false;
                        break;
                    }
                }
            if (opt_sync == 2 && !value_sync) {   // asked for explicitly: no silent fallback to events
                err = "hb_sync = value: stream memory operations (hipStreamWaitValue32 on signal memory) are not available on this device";
                return 2;
            }
        }
        HBCHK(hipMalloc((void**)&band, (size_t)O * sides * np * W * sizeof(double)));
        HBCHK(hipMalloc((void**)&buf, bufs_doubles(O * sides, twisted ? m : n, bw) * sizeof(double)));
        // hb3_chain_kernel writes only the nonzero triangles of L11^-1 and L11^-T
        HBCHK(hipMemsetAsync(buf, 0, bufs_doubles(O * sides, twisted ? m : n, bw) * sizeof(double), stream));
        HBCHK(hipMalloc((void**)&fail, (size_t)O * 3 * sizeof(int)));
        if (twisted) {
            HBCHK(hipMalloc((void**)&mid, (size_t)O * nm * W * sizeof(double)));
            HBCHK(hipMalloc((void**)&bufm, bufs_doubles(O, nm, bw) * sizeof(double)));
            HBCHK(hipMemsetAsync(bufm, 0, bufs_doubles(O, nm, bw) * sizeof(double), stream));
            HBCHK(hipMalloc((void**)&vec, (size_t)O * (6 * (size_t)np + 3 * (size_t)nm) * sizeof(double)));
        }
        HBCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hb3_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bcr_potrf_lds(HB2_NB)));
        {   // the bulk of the trailing update yields to the dependent chain on the main stream
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            HBCHK(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, greatest));
            HBCHK(hipStreamCreateWithPriority(&stream3, hipStreamNonBlocking, least));
        }
        for (auto& e : ev) HBCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : ev1) HBCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HBCHK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
        return 0;
    }

    void release() {
        for (void* p : {(void*)band, (void*)mid, (void*)buf, (void*)bufm, (void*)vec, (void*)fail})
            if (p) (void)hipFree(p);
        band = mid = buf = bufm = vec = nullptr; fail = nullptr;
        for (auto& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (auto& e : ev1) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (auto& q : sig) { if (q) (void)hipFree(q); q = nullptr; }
        if (ev2) (void)hipEventDestroy(ev2);
        ev2 = nullptr;
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream3) (void)hipStreamDestroy(stream3);
        stream2 = stream3 = nullptr;
    }

    // Right-looking blocked Cholesky of `nprob` band problems of `nrow` rows, eliminating columns [0, nelim), as a
    // three-stream pipeline over the panels of 128 columns (k = panel index):
    //   main stream   C(k)  = hb3_chain_kernel: diagonal block k minus panel k-1's contribution, Cholesky, inverse.
    //                 The dependent chain of the whole factorisation is C(0), C(1), ... back to back.
    //   stream2       T(k)  = hb2_trsm_kernel (after C(k)): the panel P = L21;
    //                 U1(k) = update part 1 (block column k+1 below its diagonal block and diagonal block k+2), plus
    //                         the copies of panel k-1 into the band (C(k) has read the entries they overwrite)
    //   stream3       U2(k) = update part 2 (the bulk, block columns >= k+2), after T(k), lowest priority.
    // C(k) needs U1(k-2) (and through it everything older); U1(k) needs U2(k-1) (same tiles); the rest is stream order.
    // So the triangular solve and both updates of panel k overlap with C(k+1).  Diagonal block k+1 is not updated in
    // the band (C(k+1) does it in LDS) except behind the last panel, where part 0 runs.  The panel buffer P
    // alternates by panel parity: slot k is read until U1(k+1)'s copies, rewritten by T(k+2) behind them.
    // Cross-stream dependencies of the pipeline: `post(which, s)` marks a point of stream s, `await(s, ticket)` makes
    // s wait for it.  Two implementations: stream memory operations on four counters in signal memory
    // (hipStreamWriteValue32 / hipStreamWaitValue32; the default), or HIP events (record / hipStreamWaitEvent).
    // A kernel behind an event wait starts ~12 us after its last dependency ended, also when the event had completed
    // long before (kernel timeline, tools/trace_window.py); behind a value wait the pipeline of eight 1024^2 images
    // runs 4 % faster (0.522 -> 0.500 s per gradient).
    enum { SIG_C = 0, SIG_T = 1, SIG_U1 = 2, SIG_U2 = 3 };
    struct Ticket { int which; unsigned value; hipEvent_t e; };
    Ticket post(int which, hipStream_t s, hipEvent_t e) {
        Ticket t{which, 0u, e};
        if (value_sync) {
            t.value = ++seq[which];
            sync_err |= hipStreamWriteValue32(s, sig[which], t.value, 0) != hipSuccess;
        } else {
            sync_err |= hipEventRecord(e, s) != hipSuccess;
        }
        return t;
    }
    void await(hipStream_t s, const Ticket& t) {
        if (value_sync) sync_err |= hipStreamWaitValue32(s, sig[t.which], t.value, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess;
        else sync_err |= hipStreamWaitEvent(s, t.e, 0) != hipSuccess;
    }

    int factor_problems(double* B, int nprob, int nrow, int nelim, const Bufs& hb, int* d_fail) {
        const int nt = (bw + 63) / 64, bwp = nt * 64;
        const int g0 = hb2_update_tiles(nt, 0), g1 = hb2_update_tiles(nt, 1), g2 = hb2_update_tiles(nt, 2);
        hipStream_t sA = single_stream ? stream : stream2, sB = single_stream ? stream : stream3;
        const size_t pblk = (size_t)nprob * bwp * HB2_NB;
        const double* nul = nullptr;
        const bool ms = !single_stream;
        int k = 0;
        const double* Plast = nul;   // the newest panel, not yet copied into the band
        int k0last = 0;
        Ticket tU1[2] = {}, tU2 = {};
        bool haveU2 = false;
        sync_err = false;
        for (int k0 = 0; k0 < nelim; k0 += HB2_NB, ++k) {
            double* Pp = hb.P + (size_t)(k & 1) * pblk;
            const bool last = k0 + HB2_NB >= nelim;
            if (k >= 2 && ms) await(stream, tU1[k & 1]);   // U1(k-2)
            hipLaunchKernelGGL(hb3_chain_kernel, dim3(nprob), dim3(BCR_PT), bcr_potrf_lds(HB2_NB), stream, B, bw, nrow, k0,
                               hb.npanel, hb.Linv, hb.LinvT, d_fail);
            if (k0 + HB2_NB >= nrow) break;   // nothing below this panel
            if (ms) await(sA, post(SIG_C, stream, ev[0]));
            hipLaunchKernelGGL(hb2_trsm_kernel, dim3(2 * nt * nprob), dim3(BG_T), 0, sA, B, bw, nrow, k0, hb.npanel, hb.Linv, Pp,
                               bwp, nprob, 0);
            if (ms && g2 > 0) await(sB, post(SIG_T, sA, ev[1]));
            if (last)   // no chain kernel follows: the next diagonal block is updated in the band
                hipLaunchKernelGGL(hb2_update_kernel, dim3(g0 * nprob), dim3(BG_T), 0, sA, B, bw, nrow, k0, (const double*)Pp, bwp, 0,
                                   nprob, nul, nul, 0);
            if (ms && haveU2) await(sA, tU2);     // U2(k-1)
            // (+ nt workgroups per problem that only copy, when there is a panel to copy)
            hipLaunchKernelGGL(hb2_update_kernel, dim3((g1 + (Plast ? nt : 0)) * nprob), dim3(BG_T), 0, sA, B, bw, nrow, k0, (const double*)Pp,
                               bwp, 1, nprob, nul, Plast, k0last);
            Plast = Pp; k0last = k0;
            if (ms) tU1[k & 1] = post(SIG_U1, sA, ev1[k & 1]);
            if (g2 > 0) {
                hipLaunchKernelGGL(hb2_update_kernel, dim3(g2 * nprob), dim3(BG_T), 0, sB, B, bw, nrow, k0, (const double*)Pp, bwp, 2,
                                   nprob, nul, nul, 0);
                if (ms) { tU2 = post(SIG_U2, sB, ev2); haveU2 = true; }
            }
        }
        if (Plast) {   // the newest panel's copy (every chain kernel that read its band entries is behind sA's last wait,
                       // or is the last launch of the main stream: wait for it)
            if (ms) await(sA, post(SIG_C, stream, ev[0]));
            hipLaunchKernelGGL(hb2_update_kernel, dim3(nt * nprob), dim3(BG_T), 0, sA, B, bw, nrow, k0last, nul, bwp, 5, nprob, nul, Plast,
                               k0last);
        }
        if (ms) {   // join: both side streams are complete before the caller continues
            await(stream, post(SIG_U1, sA, ev1[0]));
            await(stream, post(SIG_U2, sB, ev2));
        }
        if (sync_err) { err = "stream synchronisation call failed in factor_problems"; return 2; }
        HBCHK(hipGetLastError());
        return 0;
    }

    // Zero-fill of the band on the second stream, to be called BEFORE the work that produces the matrix (the PDHG
    // solve of an evaluate): 69 GB for config 5's share of one GPU, 16 ms at 4.3 TB/s, off the critical path.
    // factor() then only writes the few diagonals.
    int prefill_async() {
        if (!band || prefilled) return 0;
        HBCHK(hipMemsetAsync(band, 0, (size_t)O * sides * np * ((size_t)bw + 1) * sizeof(double), stream2));
        HBCHK(hipEventRecord(ev[2], stream2));
        prefilled = true;
        return 0;
    }

    // band <- A (diagonals D), factor; d_fail_out[img] receives 1 + the first failing column (0: none).
    int factor(const BandDiags& D, int* d_fail_out) {
        const size_t W = (size_t)bw + 1;
        HBCHK(hipMemsetAsync(fail, 0, (size_t)O * 3 * sizeof(int), stream));
        if (value_sync && *std::max_element(seq, seq + 4) > 0xF0000000u) {   // counters near wrap-around: start over
            HBCHK(hipStreamSynchronize(stream));
            HBCHK(hipStreamSynchronize(stream2));
            HBCHK(hipStreamSynchronize(stream3));
            for (int q = 0; q < 4; ++q) {
                HBCHK(hipMemset(sig[q], 0, 8));
                seq[q] = 0;
            }
        }
        if (prefilled) {   // the band is zero: write only the diagonals
            HBCHK(hipStreamWaitEvent(stream, ev[2], 0));
            const unsigned gb = (unsigned)std::min<size_t>(((size_t)np + 255) / 256, 65536);
            hipLaunchKernelGGL(hb_init_diag_kernel, dim3(gb, O * sides), dim3(256), 0, stream, D, bw, n, sides, np, band);
            prefilled = false;
        } else {
            const unsigned ib = (unsigned)std::min<size_t>(((size_t)np * W + 255) / 256, 65536);
            hipLaunchKernelGGL(hb_init_kernel, dim3(ib, O * sides), dim3(256), 0, stream, D, bw, n, sides, np, band);
        }
        const Bufs hb = carve(buf, O * sides, twisted ? m : n);
        int rc = factor_problems(band, O * sides, np, twisted ? m : n, hb, fail);
        if (rc) return rc;
        if (twisted) {
            const unsigned mb = (unsigned)std::min<size_t>(((size_t)nm * W + 255) / 256, 65536);
            hipLaunchKernelGGL(hb_mid_gather_kernel, dim3(mb, O), dim3(256), 0, stream, D, bw, n, m, nm, np, band, mid);
            const Bufs hm = carve(bufm, O, nm);
            rc = factor_problems(mid, O, nm, nm, hm, fail + 2 * O);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(hb_fail_merge_kernel, dim3((O + 63) / 64), dim3(64), 0, stream, fail, twisted ? fail + 2 * O : nullptr, sides,
                           O, d_fail_out);
        HBCHK(hipGetLastError());
        return 0;
    }

    // rows per tile workgroup of a substitution launch: the fewer, the fewer bytes a CU has to stream for its
    // workgroup (every workgroup reads the 64 KB inverse block besides its share of the tile).
    // Measured on 1024^2 images (adjoint per gradient, 32 against 128 rows): 1 image 0.395 / 0.434 s, 2 images
    // 0.401 / 0.444 s, 4 images 0.455 / 0.460 s, 8 images 0.536 / 0.497 s (64 rows: 0.525 s) -- with 16 problems per
    // launch the substitutions are bound by HBM bandwidth and the extra reads of the inverse block cost more than
    // the shorter workgroups gain.
    int subst_rw(int nprob) const {
        if (rw_force) return rw_force;
        return (long)hb2_subst_grid(bw, 32) * nprob <= 288 ? 32 : 128;
    }
    template <int RW>
    void fwd_rw(const double* B, const Bufs& hb, int nprob, int nrow, int nelim, double* x, double* y) {
        const unsigned chunks = hb2_subst_grid(bw, RW);
        for (int k0 = 0; k0 < nelim; k0 += HB2_NB)
            hipLaunchKernelGGL(hb2_fwd_kernel<RW>, dim3(chunks, nprob), dim3(BS_T), 0, stream, B, hb.Linv, bw, nrow, k0, hb.npanel, x, y, 0);
    }
    template <int RW>
    void bwd_rw(const double* B, const Bufs& hb, int nprob, int nrow, int nelim, double* y, double* x, double* acc) {
        const unsigned chunks = hb2_subst_grid(bw, RW);
        // rows of the trailing window (partial factorisation): their solution is given, push it to the earlier rows
        for (int k0 = ((nrow - 1) / HB2_NB) * HB2_NB; k0 >= nelim; k0 -= HB2_NB)
            hipLaunchKernelGGL(hb2_bwd_kernel<RW>, dim3(chunks, nprob), dim3(BS_T), 0, stream, B, hb.LinvT, bw, nrow, k0, hb.npanel, y, x,
                               (double*)nullptr, 1, 2);
        for (int k0 = ((nelim - 1) / HB2_NB) * HB2_NB; k0 >= 0; k0 -= HB2_NB)
            hipLaunchKernelGGL(hb2_bwd_kernel<RW>, dim3(chunks, nprob), dim3(BS_T), 0, stream, B, hb.LinvT, bw, nrow, k0, hb.npanel, y, x, acc,
                               0, 2);
    }
    // forward / backward sweeps of `nprob` problems over the blocks of columns [0, nelim): one launch per block of
    // 128 columns.  (Groups of four blocks per launch -- every workgroup solving the 512 x 512 triangle of the group
    // redundantly, 14 operand blocks streamed through its registers, then applying it to its own rows -- were built and
    // measured: 10 us (forward) and 20 us (backward) per block against 7-9 us here.  One CU streams about 50 GB/s,
    // so what a block needs has to stay spread over many CUs, as these launches do.)
    void fwd(const double* B, const Bufs& hb, int nprob, int nrow, int nelim, double* x, double* y) {
        switch (subst_rw(nprob)) {
            case 32: fwd_rw<32>(B, hb, nprob, nrow, nelim, x, y); break;
            default: fwd_rw<128>(B, hb, nprob, nrow, nelim, x, y);
        }
    }
    void bwd(const double* B, const Bufs& hb, int nprob, int nrow, int nelim, double* y, double* x, double* acc) {
        switch (subst_rw(nprob)) {
            case 32: bwd_rw<32>(B, hb, nprob, nrow, nelim, y, x, acc); break;
            default: bwd_rw<128>(B, hb, nprob, nrow, nelim, y, x, acc);
        }
    }

    // v <- A^-1 v, accv += solution (accv may be null).  scratch: [O][n] doubles (not twisted: holds y).
    void solve(double* v, double* accv, double* scratch) {
        if (!twisted) {
            const Bufs hb = carve(buf, O, n);
            fwd(band, hb, O, n, n, v, scratch);
            bwd(band, hb, O, n, n, scratch, v, accv);
            return;
        }
        const Bufs hb = carve(buf, 2 * O, m), hm = carve(bufm, O, nm);
        const size_t sv = (size_t)2 * O * np, sm = (size_t)O * nm;
        double *vs = vec, *ys = vec + sv, *xs = vec + 2 * sv, *vm = vec + 3 * sv, *ym = vm + sm, *xm = vm + 2 * sm;
        const unsigned gn = (unsigned)((n + 255) / 256), gp = (unsigned)((np + 255) / 256), gm = (unsigned)((nm + 255) / 256);
        hipLaunchKernelGGL(hb_tw_vec_kernel, dim3(gp, O), dim3(256), 0, stream, 0, bw, n, m, nm, np, v, vs, xs, vm, (double*)nullptr);
        fwd(band, hb, 2 * O, np, m, vs, ys);
        hipLaunchKernelGGL(hb_tw_vec_kernel, dim3(gm, O), dim3(256), 0, stream, 1, bw, n, m, nm, np, v, vs, xs, vm, (double*)nullptr);
        fwd(mid, hm, O, nm, nm, vm, ym);
        bwd(mid, hm, O, nm, nm, ym, xm, nullptr);
        hipLaunchKernelGGL(hb_tw_vec_kernel, dim3(gm, O), dim3(256), 0, stream, 2, bw, n, m, nm, np, v, vs, xs, xm, (double*)nullptr);
        bwd(band, hb, 2 * O, np, m, ys, xs, nullptr);
        hipLaunchKernelGGL(hb_tw_vec_kernel, dim3(gn, O), dim3(256), 0, stream, 3, bw, n, m, nm, np, v, vs, xs, xm, accv);
    }
#undef HBCHK
};

}  // namespace bpltv
