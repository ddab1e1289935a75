// bpltv.hip -- libbpltv: handle, launch logic and the C ABI declared in include/bpltv.h.
//
// Drop-in boundary: /root/reference/src/TVLearningFunctionVec.jl:14-27 (tv_op_learning_function),
// :45-70 (denoise), /root/reference/src/BPLDenoising.jl:41-82 (TVDenoise).  The product path is
// GPU only: there is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/bpltv.h"
#include "adjoint_hbm_kernels.hpp"
#include "hb_band_solver.hpp"
#include "hb_lu_solver.hpp"
#include <thread>
#include "nd_solver.hpp"
#include "adjoint_bcr_kernels.hpp"
#include "adjoint_kernels.hpp"
#include "pdhg_kernels.hpp"
#include "sumregs_kernels.hpp"
#include "multi_gpu.hpp"

using namespace bpltv;

namespace {

// ------------------------------------------------------------------------------------------
// PDHG kernel variants: (PI, PJ) pixels per thread, (TI, TJ) threads; region = PI*TI x PJ*TJ.
// ------------------------------------------------------------------------------------------
struct Variant {
    int RI, RJ, threads;
    void (*launch)(const PdhgArgs&, int grid, hipStream_t);
    const void* func;  // kernel symbol, for hipGraphAddKernelNode
    size_t lds;
    const char* name;
    // the single-precision instantiation (handles created with dtype = 32)
    void (*launch32)(const PdhgArgs&, int grid, hipStream_t);
    const void* func32;
    size_t lds32;
    int tiles_per_block;   // 1: one workgroup per tile (pdhg_tile_kernel); > 1: that many one-wave tiles per workgroup
    int min_image;         // 1: the image must be at least as large as the region (pdhg_rows_kernel); 2: at least as wide
    int tmax = 0;          // > 0: most iterations a launch can fuse (pdhg_stream_kernel)
};

// grid of a launch of `tiles` tiles: (nTi, nTj, images) when the kernel decodes blockIdx that way (PdhgArgs::grid3d)
inline dim3 pdhg_grid(const PdhgArgs& a, int tiles) {
    return a.grid3d ? dim3(a.nTi, a.nTj, tiles / (a.nTi * a.nTj)) : dim3(tiles);
}
// whether a launch of `nimg` images can use the 3-D grid (one tile per workgroup, grid.y / grid.z limits)
inline int pdhg_grid3d_ok(int nTj, int nimg, int tiles_per_block) { return (tiles_per_block == 1 && nTj <= 65535 && nimg <= 65535) ? 1 : 0; }

template <typename T, int PI, int PJ, int TI, int TJ>
void launch_variant(const PdhgArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = pdhg_lds_bytes(PI * TI, PJ * TJ, sizeof(T));
    hipLaunchKernelGGL((pdhg_tile_kernel<T, PI, PJ, TI, TJ>), pdhg_grid(a, grid), dim3(TI * TJ), lds, s, a);
}

#define VAR(PI, PJ, TI, TJ)                                                                    \
    { PI * TI, PJ * TJ, TI * TJ, &launch_variant<double, PI, PJ, TI, TJ>,                       \
      reinterpret_cast<const void*>(&pdhg_tile_kernel<double, PI, PJ, TI, TJ>),                 \
      pdhg_lds_bytes(PI * TI, PJ * TJ), #PI "x" #PJ "px_" #TI "x" #TJ "thr",                    \
      &launch_variant<float, PI, PJ, TI, TJ>,                                                   \
      reinterpret_cast<const void*>(&pdhg_tile_kernel<float, PI, PJ, TI, TJ>),                  \
      pdhg_lds_bytes(PI * TI, PJ * TJ, sizeof(float)), 1, 0 }
// register tiles: one wave per 32 x (2 PJ) region, WPB waves per workgroup (pdhg_wave_kernel)
template <typename T, int PJ, int WPB>
void launch_wave_variant(const PdhgArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pdhg_wave_kernel<T, PJ, WPB>), dim3((grid + WPB - 1) / WPB), dim3(64 * WPB), 0, s, a);
}
#define VARW(PJ, WPB)                                                                           \
    { 32, 2 * PJ, 64 * WPB, &launch_wave_variant<double, PJ, WPB>,                              \
      reinterpret_cast<const void*>(&pdhg_wave_kernel<double, PJ, WPB>), 0, "wave_32x" #PJ "x2",  \
      &launch_wave_variant<float, PJ, WPB>,                                                     \
      reinterpret_cast<const void*>(&pdhg_wave_kernel<float, PJ, WPB>), 0, WPB, 0 }
// 64-lane rows, PJ pixels per thread along j, TJ waves (pdhg_rows_kernel): region 64 x (PJ * TJ); CL: f and alpha in LDS
template <typename T, int PJ, int TJ, bool CL>
void launch_rows_variant(const PdhgArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pdhg_rows_kernel<T, PJ, TJ, CL>), pdhg_grid(a, grid), dim3(64 * TJ), pdhg_rows_lds(PJ, TJ, CL, sizeof(T)), s, a);
}
#define VARR(PJ, TJ, CL)                                                                        \
    { 64, PJ * TJ, 64 * TJ, &launch_rows_variant<double, PJ, TJ, CL>,                           \
      reinterpret_cast<const void*>(&pdhg_rows_kernel<double, PJ, TJ, CL>), pdhg_rows_lds(PJ, TJ, CL),  \
      "rows_64x" #PJ "px_" #TJ "waves" #CL, &launch_rows_variant<float, PJ, TJ, CL>,            \
      reinterpret_cast<const void*>(&pdhg_rows_kernel<float, PJ, TJ, CL>), pdhg_rows_lds(PJ, TJ, CL, sizeof(float)), 1, 1 }
// rows of 128 pixels, two waves side by side (pdhg_rowsw_kernel): region 128 x (PJ * TJ), 128 * TJ threads
template <typename T, int PJ, int TJ>
void launch_rowsw_variant(const PdhgArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pdhg_rowsw_kernel<T, PJ, TJ>), pdhg_grid(a, grid), dim3(128 * TJ), pdhg_rowsw_lds(PJ, TJ, sizeof(T)), s, a);
}
#define VARRW(PJ, TJ)                                                                           \
    { 128, PJ * TJ, 128 * TJ, &launch_rowsw_variant<double, PJ, TJ>,                            \
      reinterpret_cast<const void*>(&pdhg_rowsw_kernel<double, PJ, TJ>), pdhg_rowsw_lds(PJ, TJ), \
      "rows_128x" #PJ "px_" #TJ "strips", &launch_rowsw_variant<float, PJ, TJ>,                 \
      reinterpret_cast<const void*>(&pdhg_rowsw_kernel<float, PJ, TJ>), pdhg_rowsw_lds(PJ, TJ, sizeof(float)), 1, 1 }
// the same re-cut for instruction-level parallelism (pdhg_rows2_kernel): G dual chains of an interior wave in flight
template <typename T, int PJ, int TJ, int G>
void launch_rows2_variant(const PdhgArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pdhg_rows2_kernel<T, PJ, TJ, G>), pdhg_grid(a, grid), dim3(64 * TJ), pdhg_rows2_lds(PJ, TJ, sizeof(T)), s, a);
}
#define VARR2(PJ, TJ, G)                                                                        \
    { 64, PJ * TJ, 64 * TJ, &launch_rows2_variant<double, PJ, TJ, G>,                           \
      reinterpret_cast<const void*>(&pdhg_rows2_kernel<double, PJ, TJ, G>), pdhg_rows2_lds(PJ, TJ),  \
      "rows2_64x" #PJ "px_" #TJ "waves_g" #G, &launch_rows2_variant<float, PJ, TJ, G>,          \
      reinterpret_cast<const void*>(&pdhg_rows2_kernel<float, PJ, TJ, G>), pdhg_rows2_lds(PJ, TJ, sizeof(float)), 1, 1 }
// a pipeline of waves streaming down a 64-column strip (pdhg_stream_kernel): SEG rows per segment core, NL levels
template <typename T, int NL, int D, int FD, int PF, int WPE>
void launch_stream_variant(const PdhgArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pdhg_stream_kernel<T, NL, D, FD, PF, WPE>), pdhg_grid(a, grid), dim3(64 * NL), (pdhg_stream_lds<NL, D, FD>(sizeof(T))), s, a);
}
#define VARS(SEG, NL, D, FD, PF, WPE)                                                               \
    { 64, SEG + 2 * NL, 64 * NL, &launch_stream_variant<double, NL, D, FD, PF, WPE>,                  \
      reinterpret_cast<const void*>(&pdhg_stream_kernel<double, NL, D, FD, PF, WPE>), pdhg_stream_lds<NL, D, FD>(),  \
      "stream_64x" #SEG "rows_" #NL "levels_d" #D, &launch_stream_variant<float, NL, D, FD, PF, WPE>, \
      reinterpret_cast<const void*>(&pdhg_stream_kernel<float, NL, D, FD, PF, WPE>), pdhg_stream_lds<NL, D, FD>(sizeof(float)), 1, 2, NL }
const Variant kVariants[] = {
    VAR(1, 1, 32, 32),  // 1: 32x32 region, 1 px/thread   (small images, shallow blocking)
    VAR(2, 2, 32, 32),  // 2: 64x64 region, 4 px/thread
    VAR(1, 1, 16, 16),  // 3: 16x16 region
    VAR(2, 2, 16, 16),  // 4: 32x32 region, 256 threads
    VAR(2, 1, 32, 32),  // 5: 64x32 region
    VAR(4, 4, 16, 16),  // 6: 64x64 region, 256 threads, 16 px/thread
    VAR(1, 1, 64, 16),  // 7: 64x16 region (512 B rows)
    VAR(2, 1, 64, 16),  // 8: 128x16 region
    VAR(2, 2, 64, 16),  // 9: 128x32 region
    VAR(1, 2, 32, 32),  // 10: 32x64 region
    VAR(1, 2, 40, 20),  // 11: 40x40 region, 800 threads, 2 px/thread (core 24 at T = 8: 5x5 tiles per 128^2 image)
    VAR(2, 2, 20, 20),  // 12: 40x40 region, 400 threads, 4 px/thread
    VAR(1, 3, 48, 16),  // 13: 48x48 region, 768 threads, 3 px/thread   (default for images larger than 256)
    VAR(3, 2, 32, 32),  // 14: 96x64 region, 1024 threads, 6 px/thread: 150 KB of LDS, redundancy 1.60 at T = 8
    VAR(2, 3, 48, 16),  // 15: 96x48 region, 768 threads, 6 px/thread: 112 KB of LDS, redundancy 1.80 at T = 8
    VARW(16, 4),        // 16: register tiles, one wave per 32x32 region (16 px per lane), no LDS, no barriers
    VARW(8, 4),         // 17: ... per 32x16 region (8 px per lane)
    VARW(12, 4),        // 18: ... per 32x24 region (12 px per lane)
    VARR(8, 8, false),  // 19: 64-lane rows, 64x64 region, 8 px per thread, 512 threads
    VARR(6, 8, false),  // 20: ... 64x48 region, 6 px per thread
    VARR(8, 16, false), // 21: ... 64x128 region, 8 px per thread, 1024 threads
    VARR(4, 16, false), // 22: ... 64x64 region, 4 px per thread, 1024 threads
    VARR(6, 16, false), // 23: ... 64x96 region, 6 px per thread, 1024 threads
    VARR(8, 8, true),   // 24: as 19 with f and alpha in LDS
    VARR(6, 8, true),   // 25: as 20 ...
    VARR(10, 8, true),  // 26: 64x80 region, 10 px per thread, 512 threads
    VARR(12, 8, true),  // 27: 64x96 region, 12 px per thread, 512 threads
    VARR(12, 4, true),  // 28: 64x48 region, 12 px per thread, 256 threads
    VARR(16, 4, true),  // 29: 64x64 region, 16 px per thread, 256 threads
    VARR2(8, 8, 1),     // 30: rows2, 64x64 region, 8 px per thread; one dual chain at a time (the trims alone)
    VARR2(8, 8, 2),     // 31: ... two dual chains in flight
    VARS(256, 8, 4, 32, 2, 4),   // 32: streaming pipeline of 8 waves, segments of 256 rows, rings of 4 rows (76 KB of LDS), 128 VGPRs: two workgroups per CU
    VARS(256, 8, 2, 16, 2, 6),   // 33: ... rings of 2 rows (38 KB), 80 VGPRs: three workgroups per CU
    VARRW(8, 8),        // 34: rows of 128 pixels (two waves side by side), 128x64 region, 8 px per thread, 1024 threads
    VARRW(8, 4),        // 35: ... 128x32 region, 512 threads
    VARRW(6, 8),        // 36: ... 128x48 region, 6 px per thread, 1024 threads
    // (round 4, 8 x 1024^2 pixel map, 480 iterations: 32 / 33 run 2.9-3.0e4 / 3.2-3.3e4 it/s against 4.1-4.2e4 of variant 19; four or
    //  six rows of prefetch (spills) 3.0e4; segments of 128 / 352 / 512 rows 3.2e4 / 3.0e4 / 2.2e4; the same pipeline fed by a
    //  loader wave through LDS-DMA 2.2-3.3e4 -- DESIGN.md section 4.1)
    // (round 4, 8 x 1024^2 pixel map, 480 iterations: variant 19 4.09-4.22e4 it/s, 30 4.09-4.15e4, 31 4.02e4, four chains
    //  3.83e4 (spills); 64x48 / 6 px with two or three chains 3.51e4 against variant 20's 3.58e4 -- fewer moves and more
    //  chains in flight buy nothing: DESIGN.md section 4.1)
    // (64x36 / 3 px and 64x48 / 4 px with 64-thread rows -- whole halo waves that stop early -- were measured too: 86
    //  resp. 110 VGPRs, one workgroup per CU, 1.88e4 / 2.14e4 it/s on 8 x 1024^2 against 2.55e4 for variant 13)
    // (64x48 / 3 px, 48x48 / 4 px, 56x54 / 3 px, 48x48 with the 3 px along i, 64x32 / 2 px were measured on
    //  8 x 1024^2 as well: none beats variant 13)
};

struct GraphKey {
    int maxiter, T, variant, am, an, chains;
    double rho, tau0, sigma0;
    int accel, dbg, nimg;
    const void* state;
    const void* tab;   // step table (one per (maxiter, steps, L, dual-first shift): TabKey)
    int from_state;    // 1: the sequence starts from a prepared state (params.init / order), not from x = f, y = 0
    bool operator<(const GraphKey& o) const {
        return std::tie(maxiter, T, variant, am, an, chains, rho, tau0, sigma0, accel, dbg, nimg, state, tab, from_state) <
               std::tie(o.maxiter, o.T, o.variant, o.am, o.an, o.chains, o.rho, o.tau0, o.sigma0, o.accel, o.dbg, o.nimg, o.state, o.tab, o.from_state);
    }
};

struct TabKey {
    int maxiter, accel;
    double tau0, sigma0;
    double L;    // operator-norm estimate the steps are divided by: sqrt(8) (TV), sqrt(18) (sum of regularisers),
                 // or params.opnorm
    int shift;   // 1: row k carries sigma_{k+1} (dual-first order: the dual step of iteration k+1 follows the
                 // primal step of iteration k inside the fused kernel)
    bool operator<(const TabKey& o) const {
        return std::tie(maxiter, accel, tau0, sigma0, L, shift) < std::tie(o.maxiter, o.accel, o.tau0, o.sigma0, o.L, o.shift);
    }
};

struct SrGraphKey {
    int maxiter, T, am, an, accel, variant;
    double rho, tau0, sigma0;
    const void* tab;
    bool operator<(const SrGraphKey& o) const {
        return std::tie(maxiter, T, am, an, accel, variant, rho, tau0, sigma0, tab) <
               std::tie(o.maxiter, o.T, o.am, o.an, o.accel, o.variant, o.rho, o.tau0, o.sigma0, o.tab);
    }
};

}  // namespace

struct MultiState;   // shards, worker threads and the RCCL communicator of a multi-device handle (below)

// The streams of a device, shared by all its handles.  HIP maps streams onto a few hardware queues (four by default) as they
// are created; with a stream pair per handle, the two launch chains of a second handle's solve can land on ONE queue and
// run one after the other -- measured: the 5000-iteration denoise of the reference batch 5.8 -> 11.1 ms on a handle
// created while another one is alive (the reference's workflow: a training set and a validation set), back to 6.0 ms with
// GPU_MAX_HW_QUEUES=8.  So the first handle of a device creates the main stream and the chain streams back to back
// (neighbouring queues), later handles reuse them, the last one to go destroys them.  The library is quiescent when an ABI
// call returns, so handles used one after the other never meet on a stream; handles driven from several host threads at
// once are ordered by the streams they share (each keeps its own events).
struct DeviceStreams {
    hipStream_t main = nullptr;
    std::vector<hipStream_t> chains;
    int refs = 0;
};
static std::mutex g_streams_mu;
static std::map<int, DeviceStreams> g_streams;

static hipError_t device_streams_acquire(int device, hipStream_t* main) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    DeviceStreams& ds = g_streams[device];
    if (ds.refs == 0) {
        hipError_t e = hipStreamCreateWithFlags(&ds.main, hipStreamNonBlocking);
        if (e != hipSuccess) { ds.main = nullptr; return e; }
        hipStream_t cs = nullptr;
        e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);      // the second launch chain, next to the main stream
        if (e != hipSuccess) { (void)hipStreamDestroy(ds.main); ds.main = nullptr; return e; }
        ds.chains.push_back(cs);
    }
    ++ds.refs;
    *main = ds.main;
    return hipSuccess;
}
// chain stream c (0-based) of the device, created on first use
static hipError_t device_streams_chain(int device, size_t c, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    DeviceStreams& ds = g_streams[device];
    while (ds.chains.size() <= c) {
        hipStream_t cs = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        ds.chains.push_back(cs);
    }
    *out = ds.chains[c];
    return hipSuccess;
}
static void device_streams_release(int device) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    auto it = g_streams.find(device);
    if (it == g_streams.end() || it->second.refs <= 0) return;
    if (--it->second.refs == 0) {
        for (auto cs : it->second.chains) (void)hipStreamDestroy(cs);
        if (it->second.main) (void)hipStreamDestroy(it->second.main);
        g_streams.erase(it);
    }
}


// bpltv_set_option (include/bpltv.h): aids for tests and measurements; nothing here changes a result.
struct HandleOptions {
    double adjoint_budget_mb = 0.0;   // > 0: HBM the adjoint's factor workspace may take (forces image groups)
    int sr_force_lu = 0;              // 1: the LU variant of the nested dissection on the symmetric sum-of-regularisers systems too
    int nd_leaf = 0;                  // > 0: leaf size (pixels) of the nested-dissection tree; 0 = 32
    int nd_wave = 1;                  // 0: the workgroup-per-front kernel on every small level (cross-check of nd_front_wave_kernel)
    int nd_skinny = 1;                // 0: fronts of <= 32 pivots through nd_front_small_kernel / the large-regime kernels (cross-check of nd_front_skinny*_kernel)
    int nd_skinny_min = -1, nd_skinny2_min = -1;   // >= 0: (front, image) pairs a level needs for those kernels (-1: the defaults below)
    int nd_staged = 1;                // 0: substitutions of the small levels by the column-loop kernels (cross-check of nd_*_staged_kernel: same bits)
    int hb_sync = 0;                  // HBM band cross-check solver: 0 automatic, 1 HIP events, 2 stream memory operations (fails if unavailable)
    int hb_single_stream = 0;         // ... 1: its three streams folded into one (rocprofv3 --pmc)
    int hb_rw = 0;                    // ... 32 / 128: row width of its trailing update (0 automatic)
};

struct bpltv_handle {
    MultiState* multi = nullptr;   // non-null: this handle only fans out to its shards (multi_gpu.hpp)
    int M = 0, N = 0, O = 0, device = 0, ncu = 0;
    size_t npx = 0, tot = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr;   // what the launch chains wait for: everything enqueued on `stream` before the fork
    std::unique_ptr<ShardWorker> launcher;   // persistent host thread that launches the second chain's graph (lazy)
    HandleOptions opt;              // bpltv_set_option: test / measurement aids, all 0 = automatic
    bool has_data = false;
    // dataset + state
    double *d_ubar = nullptr, *d_f = nullptr;
    double* d_state[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    // current solve context: default = the O dataset images; a parameter sweep swaps in K*O slots
    double* (*cur_state)[3] = nullptr;
    int cur_nimg = 0, cur_astride = 0;
    double* d_sweep[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    size_t sweep_cap = 0;  // images
    double* d_sweep_cost = nullptr;
    int result_buf = 0;  // which state set holds the last result
    bool has_result = false;
    bool has_per_image = false;  // d_perimg (cost) and d_red (gradient partials) hold the last evaluate's rows
    double* d_alpha = nullptr;
    size_t alpha_cap = 0;
    int last_am = 1, last_an = 1;
    double alpha_min = 0.0;       // smallest entry of the last uploaded parameter (validated on the host)
    double* d_partial = nullptr;  // [1 + am*an]
    size_t partial_cap = 0;
    double* d_red = nullptr;      // reduction scratch
    size_t red_cap = 0;
    double* d_perimg = nullptr;   // [O] cost per image / gap per image
    double* d_scalar = nullptr;   // [4]
    std::map<TabKey, double*> tabs;
    // dtype = 32 (opt-in): float twins of everything pdhg_tile_kernel reads and writes; the result is widened into
    // the double state buffers after the solve, so that loss, gap, adjoint and the copies out are the f64 code
    int dtype = 64;
    float* f32_state[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    size_t f32_state_cap = 0;      // images
    float *f32_f = nullptr, *f32_alpha = nullptr;
    size_t f32_alpha_cap = 0;
    bool f32_f_valid = false;
    std::map<TabKey, float*> tabs32;
    std::map<GraphKey, std::vector<hipGraphExec_t>> graphs;  // one exec per chain
    std::vector<hipStream_t> chain_streams;   // the device's (DeviceStreams), not owned
    std::vector<hipEvent_t> chain_events;
    unsigned* d_phase = nullptr;              // the word chain 0's launches rewrite (PDHG_PHASE_STAMP, pdhg_phase_gate_kernel)
    hipStream_t capture_stream = nullptr;     // the handle's own: stream capture only (nothing ever runs on it); a capture on a shared stream
                                              // would collide with another handle of the device capturing from its own thread
    // adjoint workspace (lazy)
    bool adj_ready = false;   // common workspace
    bool band_ready = false;  // banded Cholesky workspace (LDS window or HBM band)
    bool bcr_ready = false;   // block cyclic reduction workspace
    double* d_bcr = nullptr;  // 7 block arrays [Oc][N][MP*MP]: Linv, LinvT, C, XA, XAT, XB, XBT
    int bcr_MP = 0;
    int bcr_cap = 0;          // images the block arrays hold (the adjoint runs in groups of at most that many)
    NdSolver nd;              // nested-dissection (multifrontal) Cholesky: wide images (nd_solver.hpp)
    NdSolver nd_sr;           // the same for the 13-point stencil of the sum-of-regularisers model
    double* d_coef = nullptr;   // 8 planes
    double* d_band4 = nullptr;  // 4 planes
    double* d_L = nullptr;
    double *d_invF = nullptr, *d_invB = nullptr;  // inverted 64x64 diagonal blocks of L, two layouts (x2 sides)
    double *d_L1 = nullptr, *d_dump = nullptr, *d_Lm = nullptr, *d_spill = nullptr;  // twisted factorisation
    bool adj_twisted = false;
    bool adj_hbm = false;  // M too wide for the LDS window: band factored in place in HBM (hb_band_solver.hpp)
    HbBandSolver hb;
    double *d_p = nullptr, *d_r = nullptr, *d_gpix = nullptr;
    double* d_resn = nullptr;
    int* d_fail = nullptr;
    double *d_u2 = nullptr, *d_ubar2 = nullptr;  // staging for bpltv_gradient
    // sum-of-regularisers model (sumregs_kernels.hpp): state and adjoint workspace, allocated on first use
    double* d_sr[2][7] = {{nullptr}, {nullptr}};   // x, yf1, yf2, yb1, yb2, yc1, yc2; two sets (ping-pong)
    bool sr_ready = false, sr_adj_ready = false, sr_band_ready = false;
    int last_slices = 1;                            // parameter slices of the last evaluate: 1 (TV) or 3
    int sr_result_buf = 0;
    bool sr_has_result = false;
    bool last_is_sr = false;                        // the last solve was the sum-of-regularisers model
    double *d_srcoef = nullptr, *d_srdiag = nullptr, *d_srw = nullptr, *d_srgpix = nullptr;
    HbBandSolver hb_sr;
    HbLuSolver lu_sr;             // non-symmetric row-scaled system of sumregs_gradient_reg with a patch parameter: banded LU (cross-check)
    NdSolver nd_sr_lu;            // ... the same system by nested dissection (LU variant; the default)
    bool lu_sr_ready = false;
    double* d_srdiagU = nullptr;  // its upper diagonals (7 planes)
    std::map<SrGraphKey, std::vector<hipGraphExec_t>> sr_graphs;   // one graph per launch chain
    bpltv_stats_t st;
    std::string err;
};

struct MultiState {
    std::vector<bpltv_t*> shard;                         // ordinary single-device handles
    std::vector<int> lo, hi, dev;                        // images [lo, hi) of shard k live on HIP device dev[k]
    std::vector<std::unique_ptr<ShardWorker>> worker;    // one persistent host thread per shard
    std::vector<ncclComm_t> comm;                        // ncclCommInitAll; empty when a device repeats
    int maxloc = 0;                                      // largest shard (rows per rank of the all-gather)
    std::vector<double*> d_rows, d_all;                  // all-gather send / receive buffers per shard
    std::vector<size_t> rows_cap;                        // doubles per row buffer, per shard (a failed allocation on one
                                                         // shard leaves the others' buffers and capacities consistent)
    // Parameter sweeps have a second data-parallel axis (the K parameter blocks): REPLICAS -- one handle per device the
    // caller asked for, each holding ALL O images -- let a dataset with fewer images than devices (the reference's default
    // num_samples = 1, /root/reference/src/BPLDenoising.jl:313; cameraman_128_10 holds one pair) use every device.
    // Created on the first sweep that splits the parameters (multi_sweep), filled from the shards' resident data.
    int dtype = 64;
    std::vector<int> req_dev;                            // every requested device, also those beyond min(nshards, O)
    std::vector<bpltv_t*> rep;                           // replica r lives on req_dev[r]; rep[0] == shard[0] when one shard holds everything
    std::vector<ShardWorker*> rep_worker;                // worker[r] for r < shards, an owned one beyond
    std::vector<std::unique_ptr<ShardWorker>> rep_owned;
    bool rep_data = false;                               // the replicas hold the current dataset
    int sweep_split = 0;                                 // bpltv_set_option "sweep_split": 0 automatic, 1 images, 2 parameters
    bool rep_borrowed0() const { return !rep.empty() && !shard.empty() && rep[0] == shard[0]; }
};

namespace {

int set_err(bpltv_t* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return set_err(h, BPLTV_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                  \
    } while (0)

int ensure(bpltv_t* h, double** p, size_t* cap, size_t need) {
    if (*cap >= need) return BPLTV_OK;
    if (*p) HIPCHK(h, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(h, hipMalloc((void**)p, need * sizeof(double)));
    *cap = need;
    return BPLTV_OK;
}

void fill_table(const TabKey& k, std::vector<double>& tab) {
    // oracle/bpltv_oracle.c: bplo_step_table_L (same operations, same order)
    const double L = k.L;
    double tau = k.tau0 / L, sigma = k.sigma0 / L;
    const double gamma = 1.0;
    tab.assign((size_t)TAB_STRIDE * (k.maxiter > 0 ? k.maxiter : 1), 0.0);
    for (int it = 0; it < k.maxiter; ++it) {
        const double omega = k.accel ? 1.0 / std::sqrt(1.0 + 2.0 * gamma * tau) : 1.0;
        double* r = &tab[(size_t)TAB_STRIDE * it];
        r[0] = tau;
        r[1] = sigma;
        r[2] = omega;
        r[3] = 1.0 / (1.0 + tau);
        r[4] = 1.0 + omega;
        if (k.accel) {
            tau = tau * omega;
            sigma = sigma / omega;
        }
        if (k.shift) r[1] = sigma;   // sigma_{it+1}
    }
}

// operator-norm estimate of a solve: params.opnorm, or the model's bound (sqrt(8) TV, sqrt(18) sum of regularisers)
double opnorm_of(const bpltv_params& p, double L2_default) { return p.opnorm > 0.0 ? p.opnorm : std::sqrt(L2_default); }

int get_table(bpltv_t* h, const bpltv_params& p, double** out, double L2 = 8.0, int shift = 0) {
    TabKey k{p.maxiter, p.accel ? 1 : 0, p.tau0, p.sigma0, opnorm_of(p, L2), shift};
    auto it = h->tabs.find(k);
    if (it != h->tabs.end()) {
        *out = it->second;
        return BPLTV_OK;
    }
    std::vector<double> tab;
    fill_table(k, tab);
    double* d = nullptr;
    HIPCHK(h, hipMalloc((void**)&d, tab.size() * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->tabs[k] = d;
    *out = d;
    return BPLTV_OK;
}

// ---- dtype = 32: float twins of the PDHG kernel's operands -------------------------------------------------------
void cvt_to_f32(bpltv_t* h, const double* src, float* dst, size_t n) {
    const unsigned nb = (unsigned)std::min<size_t>((n + 255) / 256, 65535);
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(nb), dim3(256), 0, h->stream, src, dst, n);
}
void cvt_to_f64(bpltv_t* h, const float* src, double* dst, size_t n) {
    const unsigned nb = (unsigned)std::min<size_t>((n + 255) / 256, 65535);
    hipLaunchKernelGGL(cvt_f32_f64_kernel, dim3(nb), dim3(256), 0, h->stream, src, dst, n);
}
void drop_graphs(bpltv_t* h);
// step table rounded to float (the oracle's bplo_pdhg_f32 rounds the same f64 table)
int get_table32(bpltv_t* h, const bpltv_params& p, float** out) {
    TabKey k{p.maxiter, p.accel ? 1 : 0, p.tau0, p.sigma0, opnorm_of(p, 8.0), 0};
    auto it = h->tabs32.find(k);
    if (it != h->tabs32.end()) {
        *out = it->second;
        return BPLTV_OK;
    }
    std::vector<double> tab;
    fill_table(k, tab);
    std::vector<float> t32(tab.begin(), tab.end());
    float* d = nullptr;
    HIPCHK(h, hipMalloc((void**)&d, t32.size() * sizeof(float)));
    HIPCHK(h, hipMemcpyAsync(d, t32.data(), t32.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->tabs32[k] = d;
    *out = d;
    return BPLTV_OK;
}
// buffers for the current solve context (cur_nimg images) and fresh float copies of f and of the parameter
int f32_prepare(bpltv_t* h) {
    if (h->f32_state_cap < (size_t)h->cur_nimg) {
        drop_graphs(h);   // captured kernels hold the old pointers
        for (int s = 0; s < 2; ++s)
            for (int c = 0; c < 3; ++c) {
                if (h->f32_state[s][c]) HIPCHK(h, hipFree(h->f32_state[s][c]));
                h->f32_state[s][c] = nullptr;
                HIPCHK(h, hipMalloc((void**)&h->f32_state[s][c], (size_t)h->cur_nimg * h->npx * sizeof(float)));
            }
        h->f32_state_cap = (size_t)h->cur_nimg;
    }
    if (!h->f32_f) HIPCHK(h, hipMalloc((void**)&h->f32_f, h->tot * sizeof(float)));
    if (!h->f32_f_valid) {
        cvt_to_f32(h, h->d_f, h->f32_f, h->tot);
        h->f32_f_valid = true;
    }
    if (h->f32_alpha_cap < h->alpha_cap) {
        drop_graphs(h);
        if (h->f32_alpha) HIPCHK(h, hipFree(h->f32_alpha));
        h->f32_alpha = nullptr;
        HIPCHK(h, hipMalloc((void**)&h->f32_alpha, h->alpha_cap * sizeof(float)));
        h->f32_alpha_cap = h->alpha_cap;
    }
    cvt_to_f32(h, h->d_alpha, h->f32_alpha, h->alpha_cap);
    HIPCHK(h, hipGetLastError());
    return BPLTV_OK;
}
// operand pointers of the PDHG kernel for this handle's dtype (float arrays travel in PdhgArgs' double* fields)
inline double* pdhg_state(bpltv_t* h, int set, int c) {
    return h->dtype == 32 ? reinterpret_cast<double*>(h->f32_state[set][c]) : h->cur_state[set][c];
}
inline const double* pdhg_f(bpltv_t* h) { return h->dtype == 32 ? reinterpret_cast<const double*>(h->f32_f) : h->d_f; }
inline const double* pdhg_alpha(bpltv_t* h) {
    return h->dtype == 32 ? reinterpret_cast<const double*>(h->f32_alpha) : h->d_alpha;
}
// the solve's result (set `buf`) widened into the double state buffers
int f32_widen(bpltv_t* h, int buf) {
    for (int c = 0; c < 3; ++c) cvt_to_f64(h, h->f32_state[buf][c], h->cur_state[buf][c], (size_t)h->cur_nimg * h->npx);
    HIPCHK(h, hipGetLastError());
    return BPLTV_OK;
}

void drop_graphs(bpltv_t* h) {
    for (auto& kv : h->graphs)
        for (auto e : kv.second) (void)hipGraphExecDestroy(e);
    h->graphs.clear();
}

int upload_alpha(bpltv_t* h, const double* alpha, int am, int an) {
    if (!alpha || am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "alpha: null pointer or empty shape");
    if (am > h->M || an > h->N)
        return set_err(h, BPLTV_E_ARG, "alpha shape %dx%d exceeds image %dx%d", am, an, h->M, h->N);
    const size_t need = (size_t)am * an;
    // The reference is defined for alpha >= 0 (alpha = 0: u = f); NaN/Inf or a negative ball radius has no
    // meaning on this path and would propagate silently through 5000 iterations.
    double amin = alpha[0];
    for (size_t e = 0; e < need; ++e) {
        if (!std::isfinite(alpha[e]) || alpha[e] < 0.0)
            return set_err(h, BPLTV_E_ARG, "alpha[%zu] = %g: parameters must be finite and >= 0", e, alpha[e]);
        if (alpha[e] < amin) amin = alpha[e];
    }
    h->alpha_min = amin;
    if (h->alpha_cap < need) {
        drop_graphs(h);  // captured kernels hold the old pointer
        int rc = ensure(h, &h->d_alpha, &h->alpha_cap, need);
        if (rc) return rc;
    }
    if (h->partial_cap < need + 1) {
        int rc = ensure(h, &h->d_partial, &h->partial_cap, need + 1);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_alpha, alpha, need * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->last_am = am;
    h->last_an = an;
    h->last_slices = 1;
    return BPLTV_OK;
}

// The same for a parameter that already lives in HBM (bpltv_denoise_device): copied device to device, checked by
// alpha_check_kernel (one 16-byte read back).
int upload_alpha_device(bpltv_t* h, const double* d_alpha, int am, int an) {
    if (!d_alpha || am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "alpha: null pointer or empty shape");
    if (am > h->M || an > h->N)
        return set_err(h, BPLTV_E_ARG, "alpha shape %dx%d exceeds image %dx%d", am, an, h->M, h->N);
    const size_t need = (size_t)am * an;
    if (h->alpha_cap < need) {
        drop_graphs(h);
        int rc = ensure(h, &h->d_alpha, &h->alpha_cap, need);
        if (rc) return rc;
    }
    if (h->partial_cap < need + 1) {
        int rc = ensure(h, &h->d_partial, &h->partial_cap, need + 1);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_alpha, d_alpha, need * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    unsigned long long* chk_d = reinterpret_cast<unsigned long long*>(h->d_scalar + 2);
    HIPCHK(h, hipMemsetAsync(chk_d, 0xFF, sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipMemsetAsync(chk_d + 1, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(alpha_check_kernel, dim3((unsigned)std::min<size_t>((need + 255) / 256, 1024)), dim3(256), 0, h->stream, h->d_alpha, need, chk_d);
    HIPCHK(h, hipGetLastError());
    unsigned long long chk_h[2] = {0, 1};
    HIPCHK(h, hipMemcpyAsync(chk_h, chk_d, sizeof(chk_h), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (chk_h[1] != 0) return set_err(h, BPLTV_E_ARG, "alpha (device array): parameters must be finite and >= 0");
    double chk[1];
    std::memcpy(chk, chk_h, sizeof(double));
    h->alpha_min = chk[0];
    h->last_am = am;
    h->last_an = an;
    h->last_slices = 1;
    return BPLTV_OK;
}

// Region, fusion depth and launch chains of one solve: plan_pdhg (tiling.hpp -- plain C++, fuzzed under the sanitizers
// by tools/plan_host_check.cpp) over the geometry of the variant table above.
int make_plan(bpltv_t* h, const bpltv_params& p, Plan* pl) {
    static const std::vector<PlanVariant> geom = [] {
        std::vector<PlanVariant> g;
        for (const Variant& V : kVariants) g.push_back(PlanVariant{V.RI, V.RJ, V.tiles_per_block, V.min_image, V.tmax});
        return g;
    }();
    PlanRequest q{h->M, h->N, h->cur_nimg, h->ncu, p.maxiter, p.tile_iters, p.reserved[0], p.reserved[1]};
    const int rc = plan_pdhg(q, geom.data(), (int)geom.size(), pl);
    switch (rc) {
        case PLAN_OK: return BPLTV_OK;
        case PLAN_E_VARIANT: return set_err(h, BPLTV_E_ARG, "unknown kernel variant %d", p.reserved[0]);
        case PLAN_E_MIN_IMAGE: {
            const Variant& V = kVariants[p.reserved[0] - 1];
            return set_err(h, BPLTV_E_ARG, "kernel variant %d needs an image of at least %dx%d pixels", p.reserved[0], V.RI, V.RJ);
        }
        case PLAN_E_TILE_ITERS: return set_err(h, BPLTV_E_ARG, "tile_iters must be >= 1");
        case PLAN_E_GRID: return set_err(h, BPLTV_E_UNSUPPORTED, "%d problems of %dx%d pixels need more than 2^31 tiles per launch", h->cur_nimg, h->M, h->N);
        default: return set_err(h, BPLTV_E_ARG, "cannot tile %dx%d with T=%d", h->M, h->N, pl->T);
    }
}

// Build one hipGraph per chain (image group): maxiter iterations as a linear launch sequence.
// The chains are replayed concurrently, each on its own stream (= its own hardware queue).
int build_graphs(bpltv_t* h, const bpltv_params& p, const Plan& pl, const double* d_tab, int niter, bool from_state,
                 std::vector<hipGraphExec_t>* out) {
    const Variant& V = kVariants[pl.variant];
    const int tilesPerImg = pl.nTi * pl.nTj;
    int rc = BPLTV_OK;
    for (int c = 0; c < pl.chains && rc == BPLTV_OK; ++c) {
        const int lo = (int)(((long)h->cur_nimg * c) / pl.chains), hi = (int)(((long)h->cur_nimg * (c + 1)) / pl.chains);
        if (hi <= lo) continue;
        hipGraph_t g = nullptr;
        HIPCHK(h, hipGraphCreate(&g, 0));
        hipGraphNode_t prev = nullptr;
        int cur = from_state ? 1 : 0;   // a prepared start lives in set 1; launch 0 always writes set 0
        // Odd chains run half a launch out of phase: their first launch fuses T/2 iterations only, so that one chain's
        // launch gaps fall into the other chain's arithmetic instead of both idling and both computing together (chains
        // of equal size otherwise stay in lockstep).  Such a chain has one launch more; it starts by writing set 1, so
        // that every chain ends in the same set.  Results do not depend on how the iterations are cut into launches.
        const int h0 = pl.T / 2;
        const bool stagger = (c & 1) && chain_out_of_phase(niter, pl.T, from_state);
        int step = stagger ? h0 : pl.T;
        // two chains: chain 0 stamps its launches, chain 1 falls in at the middle of chain 0's period after its 8th launch,
        // (pdhg_phase_gate_kernel; not in a serial replay's graphs, reserved[2] & 1: a timing aid)
        // (LDS-tile kernels only: launches of ~10 us; the row kernels' launches of ~100 us showed one kind of step only;
        // and launches of at most two workgroups per CU: longer ones -- sweeps of many problems -- are not launch-bound)
        const bool phased = pl.chains == 2 && h->d_phase != nullptr && !(p.reserved[2] & 1) && niter / pl.T >= 64 && V.RI <= 48 &&
                            (long)tilesPerImg * h->cur_nimg <= 4L * (h->ncu > 0 ? h->ncu : 256);
        int nlaunch = 0;
        for (int it = 0; it < niter; it += step, step = pl.T) {
            if (phased && c == 1 && (nlaunch == 8 || nlaunch == 40 || nlaunch == 160)) {   // 40, 160: a check, in case the sequence fell back
                unsigned* ph = h->d_phase;
                int check = nlaunch == 8 ? 0 : 1;
                void* gargs[] = {(void*)&ph, (void*)&check};
                hipKernelNodeParams gp;
                std::memset(&gp, 0, sizeof(gp));
                gp.func = reinterpret_cast<void*>(&pdhg_phase_gate_kernel);
                gp.gridDim = dim3(1); gp.blockDim = dim3(64); gp.sharedMemBytes = 0; gp.kernelParams = gargs; gp.extra = nullptr;
                hipGraphNode_t gate = nullptr;
                if (hipGraphAddKernelNode(&gate, g, prev ? &prev : nullptr, prev ? 1 : 0, &gp) == hipSuccess) prev = gate;
                else (void)hipGetLastError();
            }
            ++nlaunch;
            PdhgArgs a;
            a.f = pdhg_f(h); a.alpha = pdhg_alpha(h); a.tab = d_tab; a.rho = p.rho;
            a.am = h->last_am; a.an = h->last_an;
            a.M = h->M; a.N = h->N; a.O = h->cur_nimg;
            a.Odata = h->O; a.astride = h->cur_astride;
            a.nTi = pl.nTi; a.nTj = pl.nTj; a.halo = pl.T; a.seg = V.RJ;
            a.img0 = lo;
            a.phase = (phased && c == 0) ? h->d_phase : nullptr;
#ifdef BPLTV_EXPERIMENTS
            a.dbg = p.reserved[3];
#endif
            const int nxt = (it == 0) ? (stagger ? 1 : 0) : 1 - cur;
            a.first = (it == 0 && !from_state) ? 1 : 0;
            a.xin = pdhg_state(h, cur, 0); a.y1in = pdhg_state(h, cur, 1); a.y2in = pdhg_state(h, cur, 2);
            a.xout = pdhg_state(h, nxt, 0); a.y1out = pdhg_state(h, nxt, 1); a.y2out = pdhg_state(h, nxt, 2);
            a.it0 = it;
            a.nit = std::min(step, niter - it);
            void* kargs[] = {&a};
            hipKernelNodeParams kp;
            std::memset(&kp, 0, sizeof(kp));
            kp.func = const_cast<void*>(h->dtype == 32 ? V.func32 : V.func);
            a.ntiles = tilesPerImg * (hi - lo);
            a.xcd = (p.reserved[2] & 2) ? 1 : 0;
            a.grid3d = a.xcd ? 0 : pdhg_grid3d_ok(pl.nTj, hi - lo, V.tiles_per_block);
            kp.gridDim = a.grid3d ? pdhg_grid(a, a.ntiles) : dim3((tilesPerImg * (hi - lo) + V.tiles_per_block - 1) / V.tiles_per_block);
            kp.blockDim = dim3(V.threads);
            kp.sharedMemBytes = (unsigned)(h->dtype == 32 ? V.lds32 : V.lds);
            kp.kernelParams = kargs;
            kp.extra = nullptr;
            hipGraphNode_t node = nullptr;
            hipError_t e = hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
            if (e != hipSuccess) {
                rc = set_err(h, BPLTV_E_HIP, "hipGraphAddKernelNode: %s", hipGetErrorString(e));
                break;
            }
            prev = node;
            cur = nxt;
        }
        if (rc == BPLTV_OK) {
            hipGraphExec_t ex = nullptr;
            hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
            if (e != hipSuccess) rc = set_err(h, BPLTV_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
            else out->push_back(ex);
        }
        (void)hipGraphDestroy(g);
    }
    if (rc != BPLTV_OK) {
        for (auto e : *out) (void)hipGraphExecDestroy(e);
        out->clear();
    }
    return rc;
}

// Enqueue PDHG iterations [it0, it1) on the stream.  *buf: state set holding the current iterate
// (ignored when it0 == 0), updated to the set holding the result.
int enqueue_pdhg(bpltv_t* h, const bpltv_params& p, const Plan& pl, const double* d_tab, int it0, int it1,
                 int* buf, int* launches, bool from_state = false) {
    const Variant& V = kVariants[pl.variant];
    PdhgArgs a;
    a.f = pdhg_f(h);
    a.alpha = pdhg_alpha(h);
    a.tab = d_tab;
    a.rho = p.rho;
    a.am = h->last_am;
    a.an = h->last_an;
    a.M = h->M; a.N = h->N; a.O = h->cur_nimg;
    a.Odata = h->O; a.astride = h->cur_astride;
    a.nTi = pl.nTi; a.nTj = pl.nTj; a.halo = pl.T; a.seg = V.RJ;
    a.img0 = 0;
    a.ntiles = pl.grid;
    a.xcd = (p.reserved[2] & 2) ? 1 : 0;
    a.grid3d = a.xcd ? 0 : pdhg_grid3d_ok(pl.nTj, h->cur_nimg, V.tiles_per_block);
#ifdef BPLTV_EXPERIMENTS
    a.dbg = p.reserved[3];
#endif
    int cur = *buf;
    for (int it = it0; it < it1; it += pl.T) {
        const int nit = std::min(pl.T, it1 - it);
        const int nxt = (it == 0) ? 0 : 1 - cur;
        a.first = (it == 0 && !from_state) ? 1 : 0;
        a.xin = pdhg_state(h, cur, 0); a.y1in = pdhg_state(h, cur, 1); a.y2in = pdhg_state(h, cur, 2);
        a.xout = pdhg_state(h, nxt, 0); a.y1out = pdhg_state(h, nxt, 1); a.y2out = pdhg_state(h, nxt, 2);
        a.it0 = it;
        a.nit = nit;
        (h->dtype == 32 ? V.launch32 : V.launch)(a, pl.grid, h->stream);
        cur = nxt;
        ++*launches;
    }
    HIPCHK(h, hipGetLastError());
    *buf = cur;
    return BPLTV_OK;
}

int compute_gap(bpltv_t* h, double* gap_host /*O or null*/, double* gap_max_host) {
    h->has_per_image = false;   // d_perimg is about to hold the gaps
    const int nblk = 8;
    int rc = ensure(h, &h->d_red, &h->red_cap, (size_t)h->O * nblk * 4);
    if (rc) return rc;
    if (h->last_is_sr) {   // three-dual model: the seven state planes of the last sum-of-regularisers solve
        SrState S;
        for (int c = 0; c < 7; ++c) S.pl[c] = h->d_sr[h->sr_result_buf][c];
        hipLaunchKernelGGL(sr_gap_partial_kernel, dim3(nblk, h->O), dim3(256), 0, h->stream, S, h->d_f, h->d_alpha, h->last_am,
                           h->last_an, h->M, h->N, h->d_red);
    } else {
        const int b = h->result_buf;
        hipLaunchKernelGGL(gap_partial_kernel, dim3(nblk, h->O), dim3(256), 0, h->stream, h->d_state[b][0],
                           h->d_state[b][1], h->d_state[b][2], h->d_f, h->d_alpha, h->last_am, h->last_an, h->M,
                           h->N, h->d_red);
    }
    hipLaunchKernelGGL(gap_final_kernel, dim3(1), dim3(256), 0, h->stream, h->d_red, nblk, h->O, h->d_perimg,
                       h->d_scalar);
    HIPCHK(h, hipGetLastError());
    if (gap_host)
        HIPCHK(h, hipMemcpyAsync(gap_host, h->d_perimg, sizeof(double) * h->O, hipMemcpyDeviceToHost, h->stream));
    if (gap_max_host)
        HIPCHK(h, hipMemcpyAsync(gap_max_host, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BPLTV_OK;
}

// Replay the launch chains of one solve: chain 0 on the handle's stream, chain c on chain_streams[c - 1], all forked
// behind what the handle's stream holds now and joined back into it.  threaded: chain 1 is launched from the handle's
// persistent launcher thread (worth it from ~128 launches; short sequences are launched from this thread).  Every
// chain that was started is joined before an error is reported, so nothing is left running behind a failed call.
int launch_chains(bpltv_t* h, const std::vector<hipGraphExec_t>& ex, bool threaded) {
    while (h->chain_streams.size() + 1 < ex.size()) {   // chain 0 runs on the handle's own stream
        hipStream_t cs = nullptr;
        hipEvent_t ce = nullptr;
        HIPCHK(h, device_streams_chain(h->device, h->chain_streams.size(), &cs));   // shared by the handles of the device
        if (hipEventCreateWithFlags(&ce, hipEventDisableTiming) != hipSuccess)
            return set_err(h, BPLTV_E_HIP, "hipEventCreateWithFlags failed (launch chains)");
        h->chain_streams.push_back(cs);
        h->chain_events.push_back(ce);
    }
    HIPCHK(h, hipEventRecord(h->fork_ev, h->stream));
    std::vector<hipError_t> cerr(ex.size(), hipSuccess);
    std::vector<char> started(ex.size(), 0);
    h->st.launch_host_ms[0] = h->st.launch_host_ms[1] = 0.0;
    auto launch_chain = [&](size_t c) {
        hipError_t e = hipSuccess;
        const auto t0 = std::chrono::steady_clock::now();
        if (c == 0) {
            e = hipGraphLaunch(ex[0], h->stream);
        } else {
            hipStream_t cs = h->chain_streams[c - 1];
            e = hipStreamWaitEvent(cs, h->fork_ev, 0);
            if (e == hipSuccess) { started[c] = 1; e = hipGraphLaunch(ex[c], cs); }
            if (e == hipSuccess) e = hipEventRecord(h->chain_events[c - 1], cs);
        }
        if (c < 2) h->st.launch_host_ms[c] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        cerr[c] = e;
    };
    bool posted = false;
    if (threaded && ex.size() >= 2) {
        if (!h->launcher) {
            try { h->launcher.reset(new ShardWorker(h->device)); } catch (...) { h->launcher.reset(); }
        }
        if (h->launcher) {
            h->launcher->post([&launch_chain]() -> int { launch_chain(1); return 0; });
            posted = true;
        }
    }
    launch_chain(0);
    for (size_t c = posted ? 2 : 1; c < ex.size(); ++c) launch_chain(c);
    if (posted) (void)h->launcher->wait();
    // join every chain that got as far as its stream, whatever happened to the others
    hipError_t first = hipSuccess;
    for (size_t c = 0; c < ex.size(); ++c) {
        if (c >= 1 && started[c]) {
            hipError_t e = (cerr[c] == hipSuccess) ? hipStreamWaitEvent(h->stream, h->chain_events[c - 1], 0)
                                                   : hipStreamSynchronize(h->chain_streams[c - 1]);   // a half-launched chain: drain it
            if (cerr[c] == hipSuccess) cerr[c] = e;
        }
        if (cerr[c] != hipSuccess && first == hipSuccess) first = cerr[c];
    }
    if (first != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipGetLastError();
        return set_err(h, BPLTV_E_HIP, "launch chains: %s", hipGetErrorString(first));
    }
    return BPLTV_OK;
}

int run_pdhg(bpltv_t* h, const bpltv_params& p) {
    h->has_per_image = false;
    if (!h->has_data) return set_err(h, BPLTV_E_NODATA, "bpltv_set_data has not been called");
    if (p.maxiter < 0) return set_err(h, BPLTV_E_ARG, "maxiter < 0");
    if (p.rho != 0.0 && !(h->alpha_min > 0.0))
        return set_err(h, BPLTV_E_ARG, "rho != 0 divides by alpha: every parameter entry must be > 0 (min = %g)", h->alpha_min);
    Plan pl;
    int rc = make_plan(h, p, &pl);
    if (rc) return rc;
    // params.init / order (the choices of op_denoise_pdps the reference does not pin): the sequence starts from a
    // prepared state (pdhg_init_kernel) instead of x = f, y = 0; dual-first order = the dual step of iteration 0 in
    // that kernel, maxiter - 1 fused iterations on a table whose row k carries sigma_{k+1}, and the closing primal
    // step (pdhg_xstep_kernel).  Checker: bplo_pdhg_opts.
    const bool from_state = (p.init != 0 || p.order != 0);
    const int main_iters = p.maxiter - (p.order ? 1 : 0);
    if (from_state && h->dtype == 32)
        return set_err(h, BPLTV_E_UNSUPPORTED, "params.init / params.order are implemented for dtype = 64 handles");
    double* d_tab = nullptr;
    if (h->dtype == 32) {
        float* t32 = nullptr;
        rc = get_table32(h, p, &t32);
        if (!rc && p.maxiter > 0) rc = f32_prepare(h);
        d_tab = reinterpret_cast<double*>(t32);
    } else {
        rc = get_table(h, p, &d_tab, 8.0, p.order ? 1 : 0);
    }
    if (rc) return rc;
    h->st.tile_iters = pl.T;
    h->st.launch_chains = 1;
    h->st.pdhg_variant = pl.variant + 1;
    h->st.tiles = pl.grid;
    h->st.region_i = kVariants[pl.variant].RI;
    h->st.region_j = kVariants[pl.variant].RJ;
    h->st.launches = 0;
    h->st.iterations = 0;
    h->st.graph_used = 0;
    h->st.last_gap = -1.0;
    int buf = 0, launches = 0;
    if (p.maxiter == 0) {  // u = x0 = f (params.init = 1: 0)
        for (int c = 0; c < 3; ++c) {
            if (c == 0 && !p.init) {
                for (int r = 0; r < h->cur_nimg / h->O; ++r)
                    HIPCHK(h, hipMemcpyAsync(h->cur_state[0][0] + (size_t)r * h->tot, h->d_f, h->tot * sizeof(double),
                                             hipMemcpyDeviceToDevice, h->stream));
            } else {
                HIPCHK(h, hipMemsetAsync(h->cur_state[0][c], 0, (size_t)h->cur_nimg * h->npx * sizeof(double), h->stream));
            }
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->result_buf = 0;
        h->has_result = true;
        h->st.pdhg_ms = 0.0;
        return BPLTV_OK;
    }
    const bool chunked = p.check_every > 0;
    const size_t total = (size_t)h->cur_nimg * h->npx;
    const unsigned gtot = (unsigned)((total + 255) / 256);
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    if (from_state) {
        hipLaunchKernelGGL(pdhg_init_kernel, dim3(gtot), dim3(256), 0, h->stream, h->d_f, h->d_alpha, h->last_am, h->last_an,
                           h->M, h->N, h->O, h->cur_astride, total, p.init ? 1 : 0, p.order ? 1 : 0,
                           p.sigma0 / opnorm_of(p, 8.0), p.rho, h->cur_state[1][0], h->cur_state[1][1], h->cur_state[1][2]);
        HIPCHK(h, hipGetLastError());
        buf = 1;
    }
    if (!chunked) {
        bool done = main_iters == 0;
        if (p.use_graph && !done) {
            GraphKey key{main_iters, pl.T, pl.variant, h->last_am, h->last_an, pl.chains, p.rho, p.tau0, p.sigma0, p.accel ? 1 : 0, p.reserved[3] | ((p.reserved[2] & 3) << 16), h->cur_nimg, (const void*)pdhg_state(h, 0, 0), (const void*)d_tab, from_state ? 1 : 0};
            auto it = h->graphs.find(key);
            const int nl = (main_iters + pl.T - 1) / pl.T;
            if (it == h->graphs.end() && h->graphs.size() >= 16) {  // bounded cache
                drop_graphs(h);
                it = h->graphs.end();
            }
            if (it == h->graphs.end() && nl <= 50000) {  // longer sequences are launched eagerly
                std::vector<hipGraphExec_t> ex;
                if (build_graphs(h, p, pl, d_tab, main_iters, from_state, &ex) == BPLTV_OK && !ex.empty()) {
                    h->graphs[key] = ex;
                    it = h->graphs.find(key);
                }
                (void)hipGetLastError();
            }
            if (it != h->graphs.end()) {
                const std::vector<hipGraphExec_t>& ex = it->second;
                if (ex.size() == 1 || (p.reserved[2] & 1)) {
                    // reserved[2] = 1: replay the chains one after the other (no kernels in flight
                    // together) -- used by bench.py to time an isolated launch, as rocprofv3 sees it
                    for (size_t c = 0; c < ex.size(); ++c) HIPCHK(h, hipGraphLaunch(ex[c], h->stream));
                } else {
                    // fork: every chain waits for what is on the handle's stream now (fork_ev is recorded behind
                    // pdhg_init_kernel when the sequence starts from a prepared state; ev[0] -- the timing start -- sits in
                    // front of it).  hipGraphLaunch walks the graph on the calling thread (a few us of host time per
                    // kernel node), so the second chain is launched from the handle's launcher thread -- launched one after
                    // the other from this thread the second chain starts when the first is half done and nothing
                    // overlaps (measured: 6.83e5 it/s against 8.2e5 on the 10 x 128^2 batch).
                    rc = launch_chains(h, ex, nl >= 128 && !(p.reserved[2] & 8));   // reserved[2] & 8 (timing aid): both chains launched from the calling thread
                    if (rc) return rc;
                }
                h->st.launch_chains = (int)ex.size();
                buf = (nl - 1) % 2 == 0 ? 0 : 1;  // launch 0 writes set 0, launch l writes set l%2
                launches = nl * (int)ex.size() + (chain_out_of_phase(main_iters, pl.T, from_state) ? (int)ex.size() / 2 : 0);
                h->st.graph_used = 1;
                done = true;
            }
        }
        if (!done) {
            rc = enqueue_pdhg(h, p, pl, d_tab, 0, main_iters, &buf, &launches, from_state);
            if (rc) return rc;
        }
        h->st.iterations = p.maxiter;
        if (h->dtype == 32) {
            rc = f32_widen(h, buf);
            if (rc) return rc;
        }
    } else {
        int it = 0;
        while (it < main_iters) {
            const int it1 = std::min(main_iters, it + p.check_every);
            rc = enqueue_pdhg(h, p, pl, d_tab, it, it1, &buf, &launches, from_state);
            if (rc) return rc;
            it = it1;
            h->result_buf = buf;
            if (h->dtype == 32) {   // the gap kernels read the double state
                rc = f32_widen(h, buf);
                if (rc) return rc;
            }
            double gmax = 0.0;
            rc = compute_gap(h, nullptr, &gmax);
            if (rc) return rc;
            h->st.last_gap = gmax;
            if (p.gap_tol > 0.0 && gmax <= p.gap_tol) break;
        }
        h->st.iterations = it + ((p.order && it == main_iters) ? 1 : 0);
    }
    if (p.order && h->st.iterations == p.maxiter) {   // dual-first: the primal step of the last iteration
        hipLaunchKernelGGL(pdhg_xstep_kernel, dim3(gtot), dim3(256), 0, h->stream, h->d_f, h->cur_state[buf][1], h->cur_state[buf][2],
                           d_tab + (size_t)TAB_STRIDE * (p.maxiter - 1), h->M, h->N, h->O, total, h->cur_state[buf][0]);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    h->st.pdhg_ms = ms;
    h->st.launches = launches;
    h->result_buf = buf;
    h->has_result = true;
    h->last_is_sr = false;
    const bool amap = (h->last_am == h->M && h->last_an == h->N) && !(h->M == 1 && h->N == 1);
    h->st.bytes_per_px_iter = (amap ? 64.0 : 56.0) * (h->dtype == 32 ? 0.5 : 1.0);
    h->st.algorithmic_bytes = h->st.bytes_per_px_iter * (double)h->npx * h->cur_nimg * h->st.iterations;
    return BPLTV_OK;
}

int compute_cost(bpltv_t* h, const double* d_u, const double* d_ubar, double* d_out /*device scalar*/) {
    const int nblk = 16;
    int rc = ensure(h, &h->d_red, &h->red_cap, (size_t)h->O * nblk * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(cost_partial_kernel, dim3(nblk, h->O), dim3(256), 0, h->stream, d_u, d_ubar, (int)h->npx,
                       h->d_red);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, h->stream, h->d_red, nblk, h->O, 0.5, h->d_perimg,
                       d_out);
    HIPCHK(h, hipGetLastError());
    return BPLTV_OK;
}

size_t adj_factor_lds(int M, int NB) {  // ring (bw+NB) x (bw+1) + panel NB x (bw+NB)
    return sizeof(double) * ((size_t)(M + NB) * (M + 1) + (size_t)NB * (M + NB));
}

// hipMalloc a list of buffers; on failure every buffer of the list is freed again and its pointer cleared, so a
// failed workspace allocation leaves nothing behind (a retry starts from scratch).
struct AllocReq { void** p; size_t bytes; };
int alloc_all(bpltv_t* h, std::initializer_list<AllocReq> reqs, const char* what) {
    for (const AllocReq& r : reqs) *r.p = nullptr;
    for (const AllocReq& r : reqs) {
        const hipError_t e = hipMalloc(r.p, r.bytes);
        if (e != hipSuccess) {
            *r.p = nullptr;
            for (const AllocReq& q : reqs)
                if (*q.p) { (void)hipFree(*q.p); *q.p = nullptr; }
            (void)hipGetLastError();
            return set_err(h, e == hipErrorOutOfMemory ? BPLTV_E_NOMEM : BPLTV_E_HIP, "%s: hipMalloc of %.1f MB failed: %s", what, r.bytes / 1e6,
                           hipGetErrorString(e));
        }
    }
    return BPLTV_OK;
}

int adj_alloc(bpltv_t* h) {
    if (h->adj_ready) return BPLTV_OK;
    const size_t tot = h->tot;
    const int rc = alloc_all(h, {{(void**)&h->d_coef, 8 * tot * sizeof(double)},
                                 {(void**)&h->d_band4, 4 * tot * sizeof(double)},
                                 {(void**)&h->d_p, tot * sizeof(double)},
                                 {(void**)&h->d_r, tot * sizeof(double)},
                                 {(void**)&h->d_gpix, tot * sizeof(double)},
                                 {(void**)&h->d_resn, 4 * (size_t)h->O * (1 + RESN_BLK) * sizeof(double)},
                                 {(void**)&h->d_fail, (size_t)h->O * sizeof(int)}}, "adjoint workspace");
    if (rc) return rc;
    h->adj_ready = true;
    return BPLTV_OK;
}

// HBM the factor workspace of the adjoint may take: bpltv_set_option "adjoint_budget_mb" (a test aid: forces the
// gradient to run in image groups), else what is free now plus what this handle already holds for it, minus a 2 GB reserve.
size_t adj_budget(const bpltv_t* h, size_t held) {
    if (h->opt.adjoint_budget_mb > 0.0) return (size_t)(h->opt.adjoint_budget_mb * 1e6);
    size_t freeb = 0, totalb = 0;
    (void)hipMemGetInfo(&freeb, &totalb);
    const size_t reserve = 2ull << 30;
    return freeb + held > reserve ? freeb + held - reserve : 0;
}
// images per group so that `per_image` bytes each fit the budget: all O when they fit, at least one
int adj_group(bpltv_t* h, size_t per_image, size_t held, const char* what, int* Oc) {
    const size_t budget = adj_budget(h, held);
    size_t g = per_image ? budget / per_image : (size_t)h->O;
    if (g < 1)
        return set_err(h, BPLTV_E_NOMEM, "adjoint gradient (%s): the factor workspace of ONE %dx%d image needs %.2f GB of HBM, %.2f GB available",
                       what, h->M, h->N, per_image / 1e9, budget / 1e9);
    *Oc = (int)std::min<size_t>(std::min<size_t>(g, (size_t)h->O), 32768);   // images are a grid dimension of the solver kernels
    return BPLTV_OK;
}

// Block cyclic reduction (adjoint_bcr_kernels.hpp) applies to M <= 128, N >= 2; its seven block arrays
// take 7*N*MP^2 doubles per image (117 MB for 128^2; the odd blocks' slots stay unused because level 0
// runs in operator form).  Workspace for groups of *Oc images.
bool bcr_applicable(const bpltv_t* h) { return h->M <= BS_MP && h->N >= 2; }

int bcr_alloc(bpltv_t* h, int* Oc) {
    const int MP = (h->M + 15) / 16 * 16;
    const size_t per = BcrArrays::doubles(h->M, h->N, 1, MP) * sizeof(double);
    int rc = adj_group(h, per, (size_t)h->bcr_cap * per, "block cyclic reduction", Oc);
    if (rc) return rc;
    if (h->bcr_cap >= *Oc) return BPLTV_OK;
    if (h->d_bcr) (void)hipFree(h->d_bcr);
    h->d_bcr = nullptr; h->bcr_cap = 0; h->bcr_ready = false;
    rc = alloc_all(h, {{(void**)&h->d_bcr, (size_t)*Oc * per}}, "adjoint gradient (block cyclic reduction)");
    if (rc) return rc;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&bcr_potrf_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bcr_potrf_lds(BS_MP)));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&bcr0_schur_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bcr0_schur_lds(BS_MP)));
    h->bcr_MP = MP;
    h->bcr_cap = *Oc;
    h->bcr_ready = true;
    return BPLTV_OK;
}

// Nested-dissection Cholesky (nd_solver.hpp): tree built once per handle, workspace for groups of *Oc images.
int nd_alloc(bpltv_t* h, NdSolver& nd, const NdStencil& st, const char* what, int* Oc, bool lu = false) {
    if (!nd.built) {
        const int rc = nd.build(h->M, h->N, st, h->opt.nd_leaf, lu);
        if (rc) { const std::string m = nd.err; nd.release(); return set_err(h, rc, "adjoint gradient (%s): %s", what, m.c_str()); }
    }
    const size_t per = nd.bytes_per_image();
    int rc = adj_group(h, per, (size_t)nd.cap * per, what, Oc);
    if (rc) return rc;
    rc = nd.alloc(*Oc, h->stream);
    if (rc) return set_err(h, rc, "adjoint gradient (%s): %s", what, nd.err.c_str());
    nd.wave_fronts = h->opt.nd_wave != 0;
    nd.skinny_fronts = h->opt.nd_skinny != 0;
    nd.skinny_min = h->opt.nd_skinny_min >= 0 ? h->opt.nd_skinny_min : 0;
    nd.skinny2_min = h->opt.nd_skinny2_min >= 0 ? h->opt.nd_skinny2_min : 256;   // 10 x 128^2: its levels of 160 / 80 fronts lose 4 %, 40 x 128^2 (640 / 320) gains
    nd.staged_solve = h->opt.nd_staged != 0;
    return BPLTV_OK;
}

int band_alloc(bpltv_t* h) {
    if (h->band_ready) return BPLTV_OK;
    const size_t tot = h->tot;
    const size_t W = (size_t)h->M + 1;
    h->adj_hbm = adj_factor_lds(h->M, 4) > 160 * 1024;
    if (h->adj_hbm) {
        size_t freeb = 0, totalb = 0;
        (void)hipMemGetInfo(&freeb, &totalb);
        const size_t need = h->hb.bytes_needed(h->M, (int)h->npx, h->O);
        if (need + (2ull << 30) > freeb)
            return set_err(h, BPLTV_E_NOMEM, "adjoint gradient: the band and its inverted diagonal blocks of %d images of %dx%d need %.1f GB of HBM (%.1f GB free); the nested-dissection factorisation (the default for this shape) needs a fraction of that and runs in image groups",
                           h->O, h->M, h->N, need / 1e9, freeb / 1e9);
        h->hb.opt_sync = h->opt.hb_sync; h->hb.opt_single_stream = h->opt.hb_single_stream; h->hb.opt_rw = h->opt.hb_rw;
        const int rc = h->hb.alloc(h->M, (int)h->npx, h->O, h->stream);
        if (rc) return set_err(h, rc, "adjoint gradient (HBM band): %s", h->hb.err.c_str());
    }
    int rc = BPLTV_OK;
    if (!h->adj_hbm) {
        const size_t nblk = (h->npx + SB - 1) / SB;
        rc = alloc_all(h, {{(void**)&h->d_L, tot * W * sizeof(double)},
                           {(void**)&h->d_invF, 2 * (size_t)h->O * nblk * SB * SB * sizeof(double)},
                           {(void**)&h->d_invB, 2 * (size_t)h->O * nblk * SB * SB * sizeof(double)}}, "adjoint gradient (LDS band)");
        if (rc) return rc;
    }
    {   // two-sided (twisted) factorisation: two workgroups per image meet in a dense middle block
        const AdjSplit sp = adj_split((int)h->npx, h->M);
        const size_t midb = sizeof(double) * ((size_t)sp.nm * (sp.nm + 1) + sp.nm);
        h->adj_twisted = !h->adj_hbm && sp.m >= ADJ_G && sp.nbot >= ADJ_G && sp.nm <= 140 && midb <= 160 * 1024 &&
                         sp.nm >= h->M && (size_t)sp.nm <= (size_t)h->M + 4;
        if (h->adj_twisted) {
            const size_t blk = (size_t)(h->M + ADJ_G) * (h->M + ADJ_G);
            rc = alloc_all(h, {{(void**)&h->d_L1, tot * W * sizeof(double)},
                               {(void**)&h->d_dump, 2 * (size_t)h->O * blk * sizeof(double)},
                               {(void**)&h->d_Lm, (size_t)h->O * blk * sizeof(double)},
                               {(void**)&h->d_spill, 2 * (size_t)h->O * RING * sizeof(double)}}, "adjoint gradient (LDS band, twisted)");
            if (rc) {
                for (double** q : {&h->d_L, &h->d_invF, &h->d_invB})
                    if (*q) { (void)hipFree(*q); *q = nullptr; }
                return rc;
            }
            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_mid_factor_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_mid_solve_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    }
    if (!h->adj_hbm) {
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_factor_kernel<8, 128>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_factor_kernel<8, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_factor_kernel<4, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    h->band_ready = true;
    return BPLTV_OK;
}

// ---- the three factorisations of the reduced adjoint system ----------------------
enum AdjMethod { ADJ_BAND_LDS = 1, ADJ_BCR = 2, ADJ_BAND_HBM = 3, ADJ_BAND_LU = 4, ADJ_ND = 5, ADJ_ND_LU = 6 };   // also bpltv_stats_t::adjoint_method

// Pick the factorisation for this handle (params.reserved[4]: 0 automatic, 1 banded Cholesky, 2 block cyclic
// reduction, 3 nested dissection), make sure its workspace exists and say in groups of how many images the gradient
// runs (*Oc = O unless the factor workspace of all images does not fit: adj_group).
// Automatic = nested dissection for every shape: it needs the fewest flop and the least memory at every size measured
// (10 x 128^2: factor + solve 0.66 + 0.20 ms against 1.0 + 0.16 ms per solve of block cyclic reduction; 8 x 1024^2:
// 31 + 6 ms against 350 + 70 ms of the HBM band).  The other three stay selectable as cross-checks.
int adj_choose(bpltv_t* h, const bpltv_params& p, AdjMethod* out, int* Oc) {
    const int want = p.reserved[4];
    *Oc = h->O;
    if (want == 2) {
        if (!bcr_applicable(h))
            return set_err(h, BPLTV_E_UNSUPPORTED, "block cyclic reduction needs M <= %d and N >= 2 (M = %d, N = %d)", BS_MP,
                           h->M, h->N);
        const int rc = bcr_alloc(h, Oc);
        if (rc == BPLTV_OK) *out = ADJ_BCR;
        return rc;
    }
    if (want == 0 || want == 3) {
        const int rc = nd_alloc(h, h->nd, nd_stencil_tv(), "nested dissection", Oc);
        if (rc == BPLTV_OK) *out = ADJ_ND;
        return rc;
    }
    const int rc = band_alloc(h);   // whole batch at once (the HBM band path keeps its pipeline state per handle)
    if (rc) return rc;
    *out = h->adj_hbm ? ADJ_BAND_HBM : ADJ_BAND_LDS;
    return BPLTV_OK;
}

// Banded Cholesky in HBM (M > 138): hb_band_solver.hpp / adjoint_hbm_kernels.hpp; the matrix is handed over as
// the four diagonals of adj_assemble_kernel.
int factor_band_hbm(bpltv_t* h) {
    BandDiags D;
    D.planes = h->d_band4; D.tot = h->tot; D.nd = 4;
    D.off[0] = 0; D.off[1] = 1; D.off[2] = h->M - 1; D.off[3] = h->M;
    const int rc = h->hb.factor(D, h->d_fail);
    if (rc) return set_err(h, rc, "adjoint gradient (HBM band): %s", h->hb.err.c_str());
    return BPLTV_OK;
}

// vec <- A^-1 vec, accv += solution; d_gpix is free during the solves and holds y.
void solve_band_hbm(bpltv_t* h, double* vec, double* accv) { h->hb.solve(vec, accv, h->d_gpix); }

// Banded Cholesky with the trailing window in LDS (M <= 138), twisted when the shape allows.
struct LdsBandPlan {
    int tw;
    unsigned nblk;
    AdjSplit sp;
    size_t mid_lds;
    double *invF1, *invB1;
    explicit LdsBandPlan(const bpltv_t* h) {
        tw = h->adj_twisted ? 1 : 0;
        nblk = (unsigned)((h->npx + SB - 1) / SB);
        sp = adj_split((int)h->npx, h->M);
        mid_lds = sizeof(double) * ((size_t)sp.nm * (sp.nm + 1) + sp.nm);
        invF1 = h->d_invF + (size_t)h->O * nblk * SB * SB;
        invB1 = h->d_invB + (size_t)h->O * nblk * SB * SB;
    }
};

void factor_band_lds(bpltv_t* h) {
    const int M = h->M, N = h->N, O = h->O;
    const LdsBandPlan pl(h);
    const dim3 fgrid(O, pl.tw ? 2 : 1);
    if (M == 128)  // compile-time instance: addressing folded
        hipLaunchKernelGGL((adj_factor_kernel<8, 128>), fgrid, dim3(ADJ_FT), adj_factor_lds(M, 8), h->stream,
                           h->d_band4, M, N, O, h->d_L, h->d_L1, pl.tw, h->d_dump, h->d_fail);
    else if (adj_factor_lds(M, 8) <= 160 * 1024)
        hipLaunchKernelGGL((adj_factor_kernel<8, 0>), fgrid, dim3(ADJ_FT), adj_factor_lds(M, 8), h->stream,
                           h->d_band4, M, N, O, h->d_L, h->d_L1, pl.tw, h->d_dump, h->d_fail);
    else
        hipLaunchKernelGGL((adj_factor_kernel<4, 0>), fgrid, dim3(ADJ_FT), adj_factor_lds(M, 4), h->stream,
                           h->d_band4, M, N, O, h->d_L, h->d_L1, pl.tw, h->d_dump, h->d_fail);
    if (pl.tw) {
        hipLaunchKernelGGL(adj_mid_factor_kernel, dim3(O), dim3(256), pl.mid_lds, h->stream, h->d_band4, h->d_dump, M, N, O,
                           h->d_Lm, h->d_fail);
        hipLaunchKernelGGL(adj_invdiag_kernel, dim3(pl.nblk, O), dim3(64), 0, h->stream, h->d_L, M, N, pl.sp.m, h->d_invF,
                           h->d_invB);
        hipLaunchKernelGGL(adj_invdiag_kernel, dim3(pl.nblk, O), dim3(64), 0, h->stream, h->d_L1, M, N, pl.sp.nbot, pl.invF1,
                           pl.invB1);
    } else {
        hipLaunchKernelGGL(adj_invdiag_kernel, dim3(pl.nblk, O), dim3(64), 0, h->stream, h->d_L, M, N, (int)h->npx,
                           h->d_invF, h->d_invB);
    }
}

void solve_band_lds(bpltv_t* h, double* vec, double* accv) {
    const int M = h->M, N = h->N, O = h->O;
    const LdsBandPlan pl(h);
    if (pl.tw) {
        hipLaunchKernelGGL(adj_solve_tw_kernel<0>, dim3(O, 2), dim3(256), 0, h->stream, h->d_L, h->d_L1, h->d_invF,
                           h->d_invB, M, N, vec, accv, h->d_spill);
        hipLaunchKernelGGL(adj_mid_solve_kernel, dim3(O), dim3(256), pl.mid_lds, h->stream, h->d_Lm, M, N, vec, accv,
                           h->d_spill);
        hipLaunchKernelGGL(adj_solve_tw_kernel<1>, dim3(O, 2), dim3(256), 0, h->stream, h->d_L, h->d_L1, h->d_invF,
                           h->d_invB, M, N, vec, accv, h->d_spill);
    } else {
        hipLaunchKernelGGL(adj_solve_kernel, dim3(O), dim3(256), 0, h->stream, h->d_L, h->d_invF, h->d_invB, M, N, vec, accv);
    }
}

// Adjoint gradient of the images (d_u, d_ubar) on the device; result (am*an doubles) -> d_out.
// The images are processed in groups of at most Oc (adj_choose): coefficients, assembly, factorisation, solve and
// refinement of a group use the factor workspace of the previous one; the per-pixel gradients of all images are
// summed at the end, per image and in image order, so the result does not depend on the grouping (bitwise).
// Reference: the per-image loop of /root/reference/src/TVLearningFunctionVec.jl:76-81,168-173 -- sequential, no limit.
int run_gradient_once(bpltv_t* h, const double* d_u, const double* d_ubar, int reg, const bpltv_params& p,
                      double* d_out, double kappa_scale) {
    int rc = adj_alloc(h);
    if (rc) return rc;
    AdjMethod method;
    int Oc = h->O;
    rc = adj_choose(h, p, &method, &Oc);
    if (rc) return rc;
    const int M = h->M, N = h->N, O = h->O, am = h->last_am, an = h->last_an;
    const size_t tot = h->tot, npx = h->npx;
    const int patch = !(am == 1 && an == 1);
    const double eps = 2.220446049250313e-16;
    double kcap = p.kappa_cap > 0.0 ? p.kappa_cap : 1e14;
    double kact = 1.0 / (patch ? std::sqrt(eps) : eps);  // TVLearningFunctionVec.jl:128 / :245
    if (kact > kcap) kact = kcap;
    kact *= kappa_scale;
    // Refinement sweeps: every sweep gains ~2 digits with the 1e14 active-set weight of the scalar gradient and
    // 4-5 digits with the 6.7e7 / 1e8 weights of the patch and regularised gradients, where the second sweep
    // already reaches rounding level (round-1 sweep of the refinement count).
    // The HBM band and nested-dissection paths solve with true triangular factors (only 128 x 128 diagonal blocks
    // are inverted): one sweep already reaches the level the block-cyclic-reduction path needs two for
    // (1024^2: pixel map 3.6e-9 / patch 6e-11 / regularised 1e-16 from the converged value after ONE sweep, scalar
    // 1.4e-9 after two).  The regularised systems (gamma = 1e8 instead of 1/eps) need none: 4e-10 / 1e-9 / 5e-11 from
    // the converged value without a sweep, scaled residual <= 4e-12.
    const bool direct = (method == ADJ_BAND_HBM || method == ADJ_ND);
    const int nref_default = direct ? (reg ? 0 : (patch ? 1 : 2)) : ((patch || reg) ? 2 : 3);
    const int nref = p.refine < 0 ? nref_default : p.refine;
    HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_fail, 0, sizeof(int) * O, h->stream));
    int chunks = 0;
    for (int c0 = 0; c0 < O; c0 += Oc, ++chunks) {
        const int nimg = std::min(Oc, O - c0);
        const size_t o0 = (size_t)c0 * npx, ctot = (size_t)nimg * npx;
        AdjCoef C;   // coefficient planes of the group: the whole-batch planes at the group's first image
        C.t1 = h->d_coef + o0; C.t2 = h->d_coef + tot + o0; C.c = h->d_coef + 2 * tot + o0; C.kap = h->d_coef + 3 * tot + o0;
        C.h1 = h->d_coef + 4 * tot + o0; C.h2 = h->d_coef + 5 * tot + o0; C.s = h->d_coef + 6 * tot + o0; C.rhs = h->d_coef + 7 * tot + o0;
        double* band4 = h->d_band4;   // the four diagonals of the group's matrices, planes of nimg * npx doubles
        double *dp = h->d_p + o0, *dr = h->d_r + o0, *dg = h->d_gpix + o0;
        int* dfail = h->d_fail + c0;
        const int gpx = (int)((ctot + 255) / 256);
        hipLaunchKernelGGL(adj_setup_kernel, dim3(gpx), dim3(256), 0, h->stream, d_u + o0, d_ubar + o0, h->d_alpha, am, an, M, N,
                           nimg, patch, reg, kact, C);
        hipLaunchKernelGGL(adj_assemble_kernel, dim3(gpx), dim3(256), 0, h->stream, C, M, N, nimg, band4);
        // factorisation
        const BcrArrays bcr = BcrArrays::carve(h->d_bcr, M, N, nimg, h->bcr_MP);
        if (method == ADJ_BCR) {
            bcr_factor_band4_launch(h->stream, bcr, band4, M, N, nimg, h->bcr_MP, dfail);
        } else if (method == ADJ_ND) {
            const int r2 = h->nd.factor(band4, ctot, nimg, dfail);
            if (r2) return set_err(h, r2, "adjoint gradient (nested dissection): %s", h->nd.err.c_str());
        } else if (method == ADJ_BAND_HBM) {
            rc = factor_band_hbm(h);
            if (rc) return rc;
        } else {
            factor_band_lds(h);
        }
        HIPCHK(h, hipGetLastError());
        auto solve = [&](double* vec, double* accv) -> int {
            if (method == ADJ_BCR) bcr_solve_launch(h->stream, bcr, M, N, nimg, h->bcr_MP, vec, accv, band4);
            else if (method == ADJ_ND) {
                const int r2 = h->nd.solve(vec, accv, nimg);
                if (r2) return set_err(h, r2, "adjoint gradient (nested dissection, substitution): %s", h->nd.err.c_str());
            }
            else if (method == ADJ_BAND_HBM) solve_band_hbm(h, vec, accv);
            else solve_band_lds(h, vec, accv);
            return BPLTV_OK;
        };
        // solve + iterative refinement against the matrix-free operator
        HIPCHK(h, hipMemcpyAsync(dp, C.rhs, ctot * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        rc = solve(dp, nullptr);
        if (rc) return rc;
        for (int it = 0; it < nref; ++it) {
            hipLaunchKernelGGL(adj_residual_kernel, dim3(gpx), dim3(256), 0, h->stream, C, dp, M, N, nimg, dr);
            rc = solve(dr, dp);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(adj_residual_kernel, dim3(gpx), dim3(256), 0, h->stream, C, dp, M, N, nimg, dr);
        double* resn_part = h->d_resn + 4 * (size_t)O + 4 * (size_t)c0 * RESN_BLK;
        hipLaunchKernelGGL(adj_resnorm_kernel, dim3(RESN_BLK, nimg), dim3(256), 0, h->stream, dr, C.rhs, band4, (int)npx, resn_part);
        hipLaunchKernelGGL(adj_resnorm_final_kernel, dim3((4 * nimg + 63) / 64), dim3(64), 0, h->stream, resn_part, nimg,
                           h->d_resn + 4 * (size_t)c0);
        // gradient per pixel
        hipLaunchKernelGGL(adj_gradpix_kernel, dim3(gpx), dim3(256), 0, h->stream, C, dp, M, N, nimg, patch, reg, dg);
        HIPCHK(h, hipGetLastError());
    }
    // ... then per parameter, over all images
    if (am == M && an == N && !(M == 1 && N == 1)) {  // pixelwise parameter map: plain sum over images
        hipLaunchKernelGGL(map_sum_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, h->stream, h->d_gpix, h->npx, O,
                           d_out);
    } else {
        rc = ensure(h, &h->d_red, &h->red_cap, (size_t)am * an * O);
        if (rc) return rc;
        hipLaunchKernelGGL(patch_sum_kernel, dim3(am * an, O), dim3(256), 0, h->stream, h->d_gpix, M, N, O, am, an,
                           h->d_red);
        hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, h->stream, h->d_red, O, am * an, 1.0, d_out,
                           (double*)nullptr);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    std::vector<int> fail(O);
    std::vector<double> resn(4 * (size_t)O);
    HIPCHK(h, hipMemcpyAsync(fail.data(), h->d_fail, sizeof(int) * O, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(resn.data(), h->d_resn, sizeof(double) * 4 * O, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
    h->st.adjoint_ms = ms;
    h->st.reg_gradient_used = reg;
    h->st.adjoint_method = (int)method;
    h->st.adjoint_chunks = chunks;
    h->st.hb_sync = (method == ADJ_BAND_HBM) ? (h->hb.value_sync ? 2 : 1) : 0;
    h->st.kappa_used = reg ? 0.0 : kact;
    double worst = 0.0, worst_raw = 0.0;
    for (int k = 0; k < O; ++k) {
        if (fail[k] != 0)
            return set_err(h, BPLTV_E_NUMERIC, "adjoint Cholesky: non-positive pivot at %s %d of image %d",
                           method == ADJ_BCR ? "block" : (method == ADJ_ND ? "front" : "column"), fail[k] - 1, k);
        const double* q = &resn[4 * (size_t)k];
        const double raw = std::sqrt(q[0]) / (q[1] > 0 ? std::sqrt(q[1]) : 1.0);
        const double scl = std::sqrt(q[2]) / (q[3] > 0 ? std::sqrt(q[3]) : 1.0);
        if (!(raw <= worst_raw)) worst_raw = raw;   // NaN-propagating max
        if (!(scl <= worst)) worst = scl;
    }
    h->st.adjoint_residual = worst;
    h->st.adjoint_residual_raw = worst_raw;
    if (!(worst <= BPLTV_RESIDUAL_GATE))
        return set_err(h, BPLTV_E_NUMERIC, "adjoint solve: scaled residual %.3e above the gate %.1e (weight %.3e, %d refinement sweeps)",
                       worst, (double)BPLTV_RESIDUAL_GATE, kact, nref);
    return BPLTV_OK;
}

// The reduced system is SPD, but its active-set weight (up to 1e14) sits 14 digits above the O(1)
// terms; should rounding ever produce a non-positive pivot, retry with a 100x smaller weight
// (1e12 still reproduces the hard-constraint limit to ~1e-5).
int run_gradient(bpltv_t* h, const double* d_u, const double* d_ubar, int reg, const bpltv_params& p,
                 double* d_out) {
    const bool patch = !(h->last_am == 1 && h->last_an == 1);
    if (reg && patch && !(h->alpha_min > 0.0))
        return set_err(h, BPLTV_E_ARG, "gradient_reg with a patch / pixel-map parameter symmetrises with sqrt(alpha): every entry must be > 0 (min = %g)", h->alpha_min);
    double scale = 1.0;
    int rc = BPLTV_OK;
    h->st.adjoint_attempts = 0;
    for (int attempt = 0; attempt < 3; ++attempt, scale *= 1e-2) {
        rc = run_gradient_once(h, d_u, d_ubar, reg, p, d_out, scale);
        h->st.adjoint_attempts = attempt + 1;
        if (rc != BPLTV_E_NUMERIC) break;
        if (reg) break;   // gradient_reg has no active-set weight to reduce
    }
    // stats.kappa_used / adjoint_attempts say which system produced the gradient; a retry that succeeded is
    // not an error, so the message of the failed attempt does not stay behind
    if (rc == BPLTV_OK) h->err.clear();
    return rc;
}

bpltv_params resolve(const bpltv_params* p) {
    bpltv_params q;
    if (p) q = *p; else bpltv_default_params(&q);
    return q;
}

// Parameter checks shared by every entry point that takes a bpltv_params.
int check_params(bpltv_t* h, const bpltv_params& p) {
#ifndef BPLTV_EXPERIMENTS
    if (p.reserved[3] != 0)
        return set_err(h, BPLTV_E_ARG, "params.reserved[3] must be 0 (the timing-experiment switches exist in tools/ builds only)");
#endif
    if (!(p.tau0 > 0.0) || !(p.sigma0 > 0.0) || !std::isfinite(p.tau0) || !std::isfinite(p.sigma0))
        return set_err(h, BPLTV_E_ARG, "tau0 and sigma0 must be positive and finite");
    if (!(p.rho >= 0.0) || !std::isfinite(p.rho)) return set_err(h, BPLTV_E_ARG, "rho must be >= 0 and finite");
    if (p.reserved[4] < 0 || p.reserved[4] > 3) return set_err(h, BPLTV_E_ARG, "unknown adjoint factorisation %d", p.reserved[4]);
    if ((p.init != 0 && p.init != 1) || (p.order != 0 && p.order != 1)) return set_err(h, BPLTV_E_ARG, "params.init and params.order must be 0 or 1");
    if (!(p.opnorm >= 0.0) || !std::isfinite(p.opnorm)) return set_err(h, BPLTV_E_ARG, "params.opnorm must be >= 0 and finite (0 = default)");
    return BPLTV_OK;
}

struct WallTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

int evaluate_common(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* pp,
                    double* u_out, double* d_partial_user, double* partial_host) {
    if (!h) return BPLTV_E_ARG;
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p = resolve(pp);
    if (int prc = check_params(h, p)) return prc;
    int rc = upload_alpha(h, alpha, am, an);
    if (rc) return rc;
    if (h->band_ready && h->adj_hbm && p.reserved[4] == 1) {   // HBM band path: zero the band while the PDHG solve runs
        const int prc = h->hb.prefill_async();
        if (prc) return set_err(h, prc, "adjoint gradient (HBM band): %s", h->hb.err.c_str());
    }
    rc = run_pdhg(h, p);
    if (rc) return rc;
    const double* d_u = h->d_state[h->result_buf][0];
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
    rc = compute_cost(h, d_u, h->d_ubar, h->d_partial);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev[5], h->stream));
    const int reg = !(delta > p.delta_t);  // TVLearningFunctionVec.jl:21-25
    rc = run_gradient(h, d_u, h->d_ubar, reg, p, h->d_partial + 1);
    if (rc) return rc;
    const size_t np = 1 + (size_t)am * an;
    if (d_partial_user)
        HIPCHK(h, hipMemcpyAsync(d_partial_user, h->d_partial, np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (partial_host)
        HIPCHK(h, hipMemcpyAsync(partial_host, h->d_partial, np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (u_out)
        HIPCHK(h, hipMemcpyAsync(u_out, d_u, h->tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[4], h->ev[5]));
    h->st.cost_ms = ms;
    h->st.total_ms = wt.ms();
    h->has_per_image = !(am == h->M && an == h->N && !(h->M == 1 && h->N == 1));
    return BPLTV_OK;
}


// ============================================================================================
// Sum-of-regularisers model (sumregs_kernels.hpp; /root/reference/src/SumRegsLearningFunction.jl)
// ============================================================================================
// PDHG kernels of the three-dual model: params.variant 1 = sr_tile_kernel<32,32> (one pixel per thread), 2 =
// sr_strip_kernel<3,48,16> (48 x 48 region, three pixels per thread); 0 = by image size.
struct SrVariant {
    int R, threads;
    size_t lds;
    void (*kernel)(SrArgs);
};
const SrVariant SR_VARIANTS[] = {
    {32, 32 * 32, sr_lds_bytes(32, 32), &sr_tile_kernel<32, 32>},
    {48, 48 * 16, sr_lds_bytes(48, 48), &sr_strip_kernel<3, 48, 16>},
};
constexpr int SR_NVARIANTS = 2;

int sr_upload_alpha(bpltv_t* h, const double* alpha, int am, int an) {
    if (!alpha || am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "alpha: null pointer or empty shape");
    if (am > h->M || an > h->N) return set_err(h, BPLTV_E_ARG, "alpha shape %dx%dx3 exceeds image %dx%d", am, an, h->M, h->N);
    const size_t need = 3 * (size_t)am * an;
    double amin = alpha[0];
    for (size_t e = 0; e < need; ++e) {
        if (!std::isfinite(alpha[e]) || alpha[e] < 0.0)
            return set_err(h, BPLTV_E_ARG, "alpha[%zu] = %g: parameters must be finite and >= 0", e, alpha[e]);
        if (alpha[e] < amin) amin = alpha[e];
    }
    h->alpha_min = amin;
    if (h->alpha_cap < need) {
        drop_graphs(h);
        for (auto& kv : h->sr_graphs)
            for (auto e : kv.second) (void)hipGraphExecDestroy(e);
        h->sr_graphs.clear();
        int rc = ensure(h, &h->d_alpha, &h->alpha_cap, need);
        if (rc) return rc;
    }
    if (h->partial_cap < need + 1) {
        int rc = ensure(h, &h->d_partial, &h->partial_cap, need + 1);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_alpha, alpha, need * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->last_am = am; h->last_an = an; h->last_slices = 3;
    return BPLTV_OK;
}

int sr_alloc(bpltv_t* h) {
    if (h->sr_ready) return BPLTV_OK;
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 7; ++c) HIPCHK(h, hipMalloc((void**)&h->d_sr[s][c], h->tot * sizeof(double)));
    for (const SrVariant& v : SR_VARIANTS)
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
    h->sr_ready = true;
    return BPLTV_OK;
}

// maxiter iterations of the three-dual PDHG, T fused per launch (halo 2T), replayed from a hipGraph.
int run_sr_pdhg(bpltv_t* h, const bpltv_params& p) {
    h->has_per_image = false;
    if (!h->has_data) return set_err(h, BPLTV_E_NODATA, "bpltv_set_data has not been called");
    if (p.maxiter < 0) return set_err(h, BPLTV_E_ARG, "maxiter < 0");
    if (p.rho != 0.0 && !(h->alpha_min > 0.0))
        return set_err(h, BPLTV_E_ARG, "rho != 0 divides by alpha: every parameter entry must be > 0 (min = %g)", h->alpha_min);
    int rc = sr_alloc(h);
    if (rc) return rc;
    double* d_tab = nullptr;
    if (p.init != 0 || p.order != 0)
        return set_err(h, BPLTV_E_UNSUPPORTED, "params.init / params.order are implemented for the TV model only");
    if (h->O > 65535) return set_err(h, BPLTV_E_UNSUPPORTED, "the sum-of-regularisers solve takes at most 65535 images per handle (images are a grid dimension)");
    rc = get_table(h, p, &d_tab, 18.0);   // ||G_f||^2 + ||G_b||^2 + ||G_c||^2 <= 8 + 8 + 2 (sumregs_oracle.c: SR_L)
    if (rc) return rc;
    const int M = h->M, N = h->N;
    if (p.reserved[0] < 0 || p.reserved[0] > SR_NVARIANTS) return set_err(h, BPLTV_E_ARG, "variant %d: the sum-of-regularisers model has 1..%d", p.reserved[0], SR_NVARIANTS);
    int vi = p.reserved[0] - 1;
    if (vi < 0) {
        // Both kernels are VALU-issue bound; the 48 x 48 region recomputes 2.25 x instead of 4 x but holds one
        // workgroup per CU.  Fitted on MI355X with the two launch chains in place (ms per 1000 iterations at T = 4,
        // gpurun_out/r3s, 10 ... 40 images of 128^2, 8 x 200^2, 3 ... 6 x 256^2, 4 x 512^2, 2 x 1024^2): the one-pixel kernel
        // 0.6 + 0.0056 per 32 x 32 tile; the strip kernel 5.0 while every CU holds at most one 48 x 48 workgroup, 9.1 for
        // a second round, then 0.0157 per tile.  The strip kernel takes over from about 40 images of 128^2 or 4 of 256^2.
        const int ncu = h->ncu > 0 ? h->ncu : 256;
        auto tiles = [&](int R) {
            const int Tt = std::max(1, std::min(4, std::min((M <= R) ? 4 : (R - 1) / 4, (N <= R) ? 4 : (R - 1) / 4)));
            return (double)tile_count(M, R, 2 * Tt) * tile_count(N, R, 2 * Tt) * h->O;
        };
        const double t32 = tiles(32), t48 = tiles(48);
        const double c32 = 0.6 + 0.0056 * t32;
        const double c48 = t48 <= ncu ? 5.0 : (t48 <= 2 * ncu ? 9.1 : std::max(9.1, 0.0157 * t48));
        vi = (M <= 32 && N <= 32) ? 0 : (c48 < c32 ? 1 : 0);
    }
    const SrVariant& V = SR_VARIANTS[vi];
    const int SR_R = V.R;
    int T = p.tile_iters > 0 ? p.tile_iters : 4;   // halo 8: core 16 of the 32 x 32, 32 of the 48 x 48 region
    auto maxT = [](int L, int R) { return (L <= R) ? (1 << 20) : (R - 1) / 4; };   // 2 * halo = 4T must leave a core
    T = std::min(T, std::min(maxT(M, SR_R), maxT(N, SR_R)));
    if (T < 1) return set_err(h, BPLTV_E_ARG, "tile_iters must be >= 1");
    const int nTi = tile_count(M, SR_R, 2 * T), nTj = tile_count(N, SR_R, 2 * T);
    if (nTi < 1 || nTj < 1) return set_err(h, BPLTV_E_ARG, "cannot tile %dx%d with T=%d", M, N, T);
    const int grid = nTi * nTj * h->O;
    h->st.tile_iters = T; h->st.tiles = grid; h->st.region_i = SR_R; h->st.region_j = SR_R; h->st.pdhg_variant = vi + 1;
    h->st.launches = 0; h->st.iterations = p.maxiter; h->st.graph_used = 0; h->st.last_gap = -1.0; h->st.launch_chains = 1;
    if (p.maxiter == 0) {
        HIPCHK(h, hipMemcpyAsync(h->d_sr[0][0], h->d_f, h->tot * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        for (int c = 1; c < 7; ++c) HIPCHK(h, hipMemsetAsync(h->d_sr[0][c], 0, h->tot * sizeof(double), h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->sr_result_buf = 0; h->sr_has_result = true; h->last_is_sr = true; h->st.pdhg_ms = 0.0;
        return BPLTV_OK;
    }
    // iterations [it0, it1) of the images [lo, hi) from the state set `cur`; returns the set holding the result.
    // stagger: the chain's first launch fuses T/2 iterations and writes set 1 (run_pdhg's launch chains).
    auto enqueue_range = [&](hipStream_t st, int it0, int it1, int cur, int lo, int hi, bool stagger) -> int {
        int step = stagger ? std::max(1, T / 2) : T;
        for (int it = it0; it < it1; it += step, step = T) {
            SrArgs a;
            const int nxt = (it == 0) ? (stagger ? 1 : 0) : 1 - cur;
            for (int c = 0; c < 7; ++c) { a.in[c] = h->d_sr[cur][c]; a.out[c] = h->d_sr[nxt][c]; }
            a.f = h->d_f; a.alpha = h->d_alpha; a.tab = d_tab; a.rho = p.rho;
            a.am = h->last_am; a.an = h->last_an;
            a.it0 = it; a.nit = std::min(step, it1 - it);
            a.M = M; a.N = N; a.O = h->O; a.nTi = nTi; a.nTj = nTj; a.halo = 2 * T;
            a.first = (it == 0) ? 1 : 0;
            a.img0 = lo;
            hipLaunchKernelGGL(V.kernel, dim3(nTi, nTj, hi - lo), dim3(V.threads), V.lds, st, a);
            cur = nxt;
        }
        return cur;
    };
    auto enqueue = [&](hipStream_t st) -> int { return enqueue_range(st, 0, p.maxiter, 0, 0, h->O, false); };
    if (p.check_every > 0) {   // duality-gap checks every check_every iterations, early stop at gap_tol (as the TV model)
        HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
        int it = 0, cur = 0, launches = 0;
        h->last_is_sr = true;
        while (it < p.maxiter) {
            const int it1 = std::min(p.maxiter, it + p.check_every);
            cur = enqueue_range(h->stream, it, it1, cur, 0, h->O, false);
            launches += (it1 - it + T - 1) / T;
            it = it1;
            HIPCHK(h, hipGetLastError());
            h->sr_result_buf = cur;
            h->sr_has_result = true;
            double gmax = 0.0;
            rc = compute_gap(h, nullptr, &gmax);
            if (rc) return rc;
            h->st.last_gap = gmax;
            if (p.gap_tol > 0.0 && gmax <= p.gap_tol) break;
        }
        HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms2 = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms2, h->ev[0], h->ev[1]));
        h->st.pdhg_ms = ms2;
        h->st.launches = launches;
        h->st.iterations = it;
        const bool amap2 = (h->last_am == M && h->last_an == N) && !(M == 1 && N == 1);
        h->st.bytes_per_px_iter = amap2 ? 144.0 : 120.0;
        h->st.algorithmic_bytes = h->st.bytes_per_px_iter * (double)h->tot * it;
        return BPLTV_OK;
    }
    const int nl = (p.maxiter + T - 1) / T;
    const int buf = (nl - 1) % 2 == 0 ? 0 : 1;   // launch 0 writes set 0, launch l writes set l % 2
    // two launch chains (image groups) as in run_pdhg: chain 0 on the handle's stream, chain 1 on a second one, half a
    // launch out of phase; reserved[1] = 1 keeps one chain
    const int h0 = std::max(1, T / 2);
    int nch = p.reserved[1] > 0 ? std::min(p.reserved[1], 2) : ((2 * grid > 3 * (h->ncu > 0 ? h->ncu : 256) && h->O >= 2) ? 2 : 1);
    if (nch > h->O) nch = h->O;
    const bool stag = T >= 2 && nl >= 8 && ((1 + (p.maxiter - h0 + T - 1) / T) - nl) % 2 == 1;
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    bool done = false;
    if (p.use_graph && nl <= 50000) {
        SrGraphKey key{p.maxiter, T, h->last_am, h->last_an, p.accel ? 1 : 0, vi + 16 * nch, p.rho, p.tau0, p.sigma0, (const void*)d_tab};
        auto it = h->sr_graphs.find(key);
        if (it == h->sr_graphs.end()) {
            if (h->sr_graphs.size() >= 8) {
                for (auto& kv : h->sr_graphs)
                    for (auto e : kv.second) (void)hipGraphExecDestroy(e);
                h->sr_graphs.clear();
            }
            std::vector<hipGraphExec_t> exs;
            for (int c = 0; c < nch; ++c) {
                const int lo = (int)(((long)h->O * c) / nch), hi = (int)(((long)h->O * (c + 1)) / nch);
                hipGraph_t g = nullptr;
                hipGraphExec_t ex = nullptr;
                if (!h->capture_stream && hipStreamCreateWithFlags(&h->capture_stream, hipStreamNonBlocking) != hipSuccess) { h->capture_stream = nullptr; break; }
                if (hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    (void)enqueue_range(h->capture_stream, 0, p.maxiter, 0, lo, hi, (c & 1) && stag);
                    if (hipStreamEndCapture(h->capture_stream, &g) == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess)
                        exs.push_back(ex);
                    if (g) (void)hipGraphDestroy(g);
                }
            }
            (void)hipGetLastError();
            if ((int)exs.size() == nch) {
                h->sr_graphs[key] = exs;
                it = h->sr_graphs.find(key);
            } else {
                for (auto e : exs) (void)hipGraphExecDestroy(e);
            }
        }
        if (it != h->sr_graphs.end()) {
            const std::vector<hipGraphExec_t>& exs = it->second;
            if (exs.size() == 1) {
                HIPCHK(h, hipGraphLaunch(exs[0], h->stream));
            } else {
                rc = launch_chains(h, exs, nl >= 128);   // short sequences: a helper thread costs more than it hides
                if (rc) return rc;
            }
            h->st.launch_chains = (int)exs.size();
            h->st.graph_used = 1;
            done = true;
        }
    }
    if (!done) (void)enqueue(h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    h->st.pdhg_ms = ms;
    h->st.launches = done ? nl * nch + ((nch > 1 && stag) ? 1 : 0) : nl;
    h->sr_result_buf = buf;
    h->sr_has_result = true;
    h->last_is_sr = true;
    const bool amap = (h->last_am == M && h->last_an == N) && !(M == 1 && N == 1);
    h->st.bytes_per_px_iter = amap ? 144.0 : 120.0;   // read x, 6 y, f (+ 3 alpha), write x, 6 y
    h->st.algorithmic_bytes = h->st.bytes_per_px_iter * (double)h->tot * p.maxiter;
    return BPLTV_OK;
}

int sr_adj_alloc(bpltv_t* h) {
    int rc = adj_alloc(h);   // d_p, d_r, d_gpix, d_resn, d_fail
    if (rc) return rc;
    if (h->sr_adj_ready) return BPLTV_OK;
    const size_t tot = h->tot;
    rc = alloc_all(h, {{(void**)&h->d_srcoef, 19 * tot * sizeof(double)},
                       {(void**)&h->d_srdiag, 7 * tot * sizeof(double)},
                       {(void**)&h->d_srw, 6 * tot * sizeof(double)},
                       {(void**)&h->d_srgpix, 3 * tot * sizeof(double)}}, "sum-of-regularisers adjoint workspace");
    if (rc) return rc;
    h->sr_adj_ready = true;
    return BPLTV_OK;
}

// HBM band solver of the 13-point system (bandwidth 2M), whole batch: the cross-check of the nested-dissection path
int sr_band_alloc(bpltv_t* h) {
    if (h->sr_band_ready) return BPLTV_OK;
    size_t freeb = 0, totalb = 0;
    (void)hipMemGetInfo(&freeb, &totalb);
    const int n = (int)h->npx, bw = std::min(2 * h->M, n - 1);
    const size_t need = h->hb_sr.bytes_needed(bw, n, h->O);
    if (need + (2ull << 30) > freeb)
        return set_err(h, BPLTV_E_NOMEM, "sum-of-regularisers adjoint (HBM band): %.1f GB of HBM needed, %.1f GB free", need / 1e9, freeb / 1e9);
    h->hb_sr.opt_sync = h->opt.hb_sync; h->hb_sr.opt_single_stream = h->opt.hb_single_stream; h->hb_sr.opt_rw = h->opt.hb_rw;
    const int rc = h->hb_sr.alloc(bw, n, h->O, h->stream);
    if (rc) return set_err(h, rc, "sum-of-regularisers adjoint (HBM band): %s", h->hb_sr.err.c_str());
    h->sr_band_ready = true;
    return BPLTV_OK;
}

// Gradient of the sum-of-regularisers model.  Factorisations of its 13-point system: nested dissection (default,
// separators two pixels wide, image groups when the workspace does not fit), the HBM band at bandwidth 2M
// (params.reserved[4] = 1), and -- sumregs_gradient_reg with a patch parameter, whose row-scaled system is not
// symmetric (SumRegsLearningFunction.jl:250) -- the LU variant of the nested dissection (banded LU with
// params.reserved[4] = 1).
int run_sr_gradient_once(bpltv_t* h, const double* d_u, const double* d_ubar, int reg, const bpltv_params& p, double* d_out,
                         double kappa_scale) {
    int rc = sr_adj_alloc(h);
    if (rc) return rc;
    const int M = h->M, N = h->N, O = h->O, am = h->last_am, an = h->last_an;
    const size_t tot = h->tot, P = (size_t)am * an, npx = h->npx;
    const int patch = !(am == 1 && an == 1);
    const bool rowsc = reg && patch;
    const bool lu = rowsc || h->opt.sr_force_lu == 1;   // option "sr_force_lu": test aid, the LU path on the symmetric systems too
    if (p.reserved[4] == 2) return set_err(h, BPLTV_E_UNSUPPORTED, "block cyclic reduction applies to the TV model only");
    const bool band = p.reserved[4] == 1;     // the band solvers (Cholesky / LU) instead of nested dissection
    int Oc = O;
    if (lu && rowsc && !(h->alpha_min > 0.0))
        return set_err(h, BPLTV_E_ARG, "sumregs_gradient_reg with a patch parameter needs every entry > 0 (min = %g)", h->alpha_min);
    if (lu && !h->d_srdiagU) {
        rc = alloc_all(h, {{(void**)&h->d_srdiagU, 7 * tot * sizeof(double)}}, "sum-of-regularisers adjoint (upper diagonals)");
        if (rc) return rc;
    }
    if (lu && !band) {
        rc = nd_alloc(h, h->nd_sr_lu, nd_stencil_sr(), "sum of regularisers, nested dissection (LU)", &Oc, true);
        if (rc) return rc;
    } else if (lu) {
        if (!h->lu_sr_ready) {
            const int n = (int)h->npx, bw = std::min(2 * M, n - 1);
            size_t freeb = 0, totalb = 0;
            (void)hipMemGetInfo(&freeb, &totalb);
            const size_t need = h->lu_sr.bytes_needed(bw, n, O);
            if (need + (2ull << 30) > freeb)
                return set_err(h, BPLTV_E_NOMEM, "sum-of-regularisers adjoint (banded LU): %.1f GB of HBM needed, %.1f GB free", need / 1e9, freeb / 1e9);
            const int rc2 = h->lu_sr.alloc(bw, n, O, h->stream);
            if (rc2) return set_err(h, rc2, "sum-of-regularisers adjoint (banded LU): %s", h->lu_sr.err.c_str());
            h->lu_sr_ready = true;
        }
    } else if (band) {
        rc = sr_band_alloc(h);
        if (rc) return rc;
    } else {
        rc = nd_alloc(h, h->nd_sr, nd_stencil_sr(), "sum of regularisers, nested dissection", &Oc);
        if (rc) return rc;
    }
    double kact = 1.0 / 2.220446049250313e-16;   // eps() in the vector AND the patch variant (:319, :389)
    const double kcap = p.kappa_cap > 0.0 ? p.kappa_cap : 1e14;
    if (kact > kcap) kact = kcap;
    kact *= kappa_scale;
    const int nref = p.refine < 0 ? (reg ? 1 : 2) : p.refine;
    HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_fail, 0, sizeof(int) * O, h->stream));
    const double* rowscale = rowsc ? h->d_alpha : nullptr;
    int chunks = 0;
    for (int c0 = 0; c0 < O; c0 += Oc, ++chunks) {
        const int nimg = std::min(Oc, O - c0);
        const size_t o0 = (size_t)c0 * npx, ctot = (size_t)nimg * npx;
        SrCoef C;   // planes keep the whole-batch stride C.tot; the group starts at its first image
        C.tot = tot;
        C.t1 = h->d_srcoef + o0; C.t2 = h->d_srcoef + 3 * tot + o0; C.c = h->d_srcoef + 6 * tot + o0; C.kap = h->d_srcoef + 9 * tot + o0;
        C.h1 = h->d_srcoef + 12 * tot + o0; C.h2 = h->d_srcoef + 15 * tot + o0; C.rhs = h->d_srcoef + 18 * tot + o0;
        double *diag = h->d_srdiag + o0, *diagU = lu ? h->d_srdiagU + o0 : nullptr, *w = h->d_srw + o0, *gp = h->d_srgpix + o0;
        double *dp = h->d_p + o0, *dr = h->d_r + o0;
        int* dfail = h->d_fail + c0;
        const int gpx = (int)((ctot + 255) / 256);
        hipLaunchKernelGGL(sr_adj_setup_kernel, dim3(gpx), dim3(256), 0, h->stream, d_u + o0, d_ubar + o0, h->d_alpha, am, an, M, N, nimg,
                           patch, reg, kact, C);
        if (lu) HIPCHK(h, hipMemsetAsync(h->d_srdiagU, 0, 7 * tot * sizeof(double), h->stream));
        hipLaunchKernelGGL(sr_adj_assemble_kernel, dim3(gpx), dim3(256), 0, h->stream, C, M, N, nimg, diag, rowscale, am, an, diagU);
        BandDiags D;
        D.planes = diag; D.tot = tot; D.nd = 7;
        D.off[0] = 0; D.off[1] = 1; D.off[2] = 2; D.off[3] = M - 1; D.off[4] = M; D.off[5] = M + 1; D.off[6] = 2 * M;
        if (lu && !band) {
            rc = h->nd_sr_lu.factor_lu(diag, diagU, tot, nimg, dfail);
            if (rc) return set_err(h, rc, "sum-of-regularisers adjoint (nested dissection, LU): %s", h->nd_sr_lu.err.c_str());
        } else if (lu) {
            BandDiags DU = D;
            DU.planes = diagU;
            rc = h->lu_sr.factor(D, DU, dfail);
            if (rc) return set_err(h, rc, "sum-of-regularisers adjoint (banded LU): %s", h->lu_sr.err.c_str());
        } else if (band) {
            rc = h->hb_sr.factor(D, dfail);
            if (rc) return set_err(h, rc, "sum-of-regularisers adjoint (HBM band): %s", h->hb_sr.err.c_str());
        } else {
            rc = h->nd_sr.factor(diag, tot, nimg, dfail);
            if (rc) return set_err(h, rc, "sum-of-regularisers adjoint (nested dissection): %s", h->nd_sr.err.c_str());
        }
        auto residual = [&](double* out) {
            hipLaunchKernelGGL(sr_adj_flux_kernel, dim3(gpx), dim3(256), 0, h->stream, C, dp, M, N, nimg, w);
            hipLaunchKernelGGL(sr_adj_residual_kernel, dim3(gpx), dim3(256), 0, h->stream, C, dp, w, M, N, nimg, out, rowscale, am, an);
        };
        auto solve = [&](double* v, double* acc) -> int {
            int r2 = 0;
            if (lu && !band) { if ((r2 = h->nd_sr_lu.solve(v, acc, nimg))) return set_err(h, r2, "sum-of-regularisers adjoint (nested dissection, LU, substitution): %s", h->nd_sr_lu.err.c_str()); }
            else if (lu) h->lu_sr.solve(v, acc, h->d_gpix);
            else if (band) h->hb_sr.solve(v, acc, h->d_gpix);
            else if ((r2 = h->nd_sr.solve(v, acc, nimg))) return set_err(h, r2, "sum-of-regularisers adjoint (nested dissection, substitution): %s", h->nd_sr.err.c_str());
            return BPLTV_OK;
        };
        HIPCHK(h, hipMemcpyAsync(dp, C.rhs, ctot * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        rc = solve(dp, nullptr);
        if (rc) return rc;
        for (int it = 0; it < nref; ++it) {
            residual(dr);
            rc = solve(dr, dp);
            if (rc) return rc;
        }
        residual(dr);
        double* resn_part = h->d_resn + 4 * (size_t)O + 4 * (size_t)c0 * RESN_BLK;
        hipLaunchKernelGGL(adj_resnorm_kernel, dim3(RESN_BLK, nimg), dim3(256), 0, h->stream, dr, C.rhs, diag, (int)npx, resn_part);
        hipLaunchKernelGGL(adj_resnorm_final_kernel, dim3((4 * nimg + 63) / 64), dim3(64), 0, h->stream, resn_part, nimg, h->d_resn + 4 * (size_t)c0);
        hipLaunchKernelGGL(sr_adj_gradpix_kernel, dim3(gpx), dim3(256), 0, h->stream, C, dp, M, N, nimg, patch, reg, gp);
        HIPCHK(h, hipGetLastError());
    }
    if (am == M && an == N && !(M == 1 && N == 1)) {   // three pixelwise maps: plain sums over the images
        for (int k = 0; k < 3; ++k)
            hipLaunchKernelGGL(map_sum_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, h->stream, h->d_srgpix + k * tot,
                               h->npx, O, d_out + (size_t)k * h->npx);
    } else {
        rc = ensure(h, &h->d_red, &h->red_cap, 3 * P * O);
        if (rc) return rc;
        for (int k = 0; k < 3; ++k)
            hipLaunchKernelGGL(patch_sum_kernel, dim3((unsigned)P, O), dim3(256), 0, h->stream, h->d_srgpix + k * tot, M, N, O, am, an,
                               h->d_red + (size_t)k * P * O);
        hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, h->stream, h->d_red, O, (int)(3 * P), 1.0, d_out, (double*)nullptr);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    std::vector<int> fail(O);
    std::vector<double> resn(4 * (size_t)O);
    HIPCHK(h, hipMemcpyAsync(fail.data(), h->d_fail, sizeof(int) * O, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(resn.data(), h->d_resn, sizeof(double) * 4 * O, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
    h->st.adjoint_ms = ms;
    h->st.reg_gradient_used = reg;
    h->st.adjoint_method = lu ? (band ? (int)ADJ_BAND_LU : (int)ADJ_ND_LU) : (band ? (int)ADJ_BAND_HBM : (int)ADJ_ND);
    h->st.adjoint_chunks = chunks;
    h->st.hb_sync = (band && !lu) ? (h->hb_sr.value_sync ? 2 : 1) : 0;
    h->st.kappa_used = reg ? 0.0 : kact;
    double worst = 0.0, worst_raw = 0.0;
    for (int k = 0; k < O; ++k) {
        if (fail[k] != 0)
            return set_err(h, BPLTV_E_NUMERIC, lu ? (band ? "sum-of-regularisers adjoint, banded LU without pivoting: zero, tiny or non-finite pivot at column %d of image %d"
                                                          : "sum-of-regularisers adjoint, LU without pivoting: zero, tiny or non-finite pivot at front %d of image %d")
                                                  : (band ? "sum-of-regularisers adjoint Cholesky: non-positive pivot at column %d of image %d"
                                                          : "sum-of-regularisers adjoint Cholesky: non-positive pivot at front %d of image %d"), fail[k] - 1, k);
        const double* q = &resn[4 * (size_t)k];
        const double raw = std::sqrt(q[0]) / (q[1] > 0 ? std::sqrt(q[1]) : 1.0);
        const double scl = std::sqrt(q[2]) / (q[3] > 0 ? std::sqrt(q[3]) : 1.0);
        if (!(raw <= worst_raw)) worst_raw = raw;
        if (!(scl <= worst)) worst = scl;
    }
    h->st.adjoint_residual = worst;
    h->st.adjoint_residual_raw = worst_raw;
    if (!(worst <= BPLTV_RESIDUAL_GATE))
        return set_err(h, BPLTV_E_NUMERIC, "sum-of-regularisers adjoint solve: scaled residual %.3e above the gate %.1e", worst,
                       (double)BPLTV_RESIDUAL_GATE);
    return BPLTV_OK;
}

int run_sr_gradient(bpltv_t* h, const double* d_u, const double* d_ubar, int reg, const bpltv_params& p, double* d_out) {
    double scale = 1.0;
    int rc = BPLTV_OK;
    h->st.adjoint_attempts = 0;
    for (int attempt = 0; attempt < 3; ++attempt, scale *= 1e-2) {
        rc = run_sr_gradient_once(h, d_u, d_ubar, reg, p, d_out, scale);
        h->st.adjoint_attempts = attempt + 1;
        if (rc != BPLTV_E_NUMERIC || reg) break;
    }
    if (rc == BPLTV_OK) h->err.clear();
    return rc;
}

// sumregs_learning_function(x, data, D): SumRegsLearningFunction.jl:8-36.  partial: [cost, grad (3*am*an)...]
int sr_evaluate_common(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* pp, double* u_out,
                       double* partial_host) {
    if (!h) return BPLTV_E_ARG;
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p = resolve(pp);
    if (int prc = check_params(h, p)) return prc;
    int rc = sr_upload_alpha(h, alpha, am, an);
    if (rc) return rc;
    rc = run_sr_pdhg(h, p);
    if (rc) return rc;
    const double* d_u = h->d_sr[h->sr_result_buf][0];
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
    rc = compute_cost(h, d_u, h->d_ubar, h->d_partial);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev[5], h->stream));
    const int reg = !(delta > p.delta_t);   // SumRegsLearningFunction.jl:14-18, 30-34
    rc = run_sr_gradient(h, d_u, h->d_ubar, reg, p, h->d_partial + 1);
    if (rc) return rc;
    const size_t np = 1 + 3 * (size_t)am * an;
    if (partial_host) HIPCHK(h, hipMemcpyAsync(partial_host, h->d_partial, np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (u_out) HIPCHK(h, hipMemcpyAsync(u_out, d_u, h->tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[4], h->ev[5]));
    h->st.cost_ms = ms;
    h->st.total_ms = wt.ms();
    h->has_per_image = !(am == h->M && an == h->N && !(h->M == 1 && h->N == 1));
    return BPLTV_OK;
}

// ============================================================================================
// Multi-device handle (multi_gpu.hpp): every entry point fans out to the shards' worker threads.
// ============================================================================================
#define NCCLCHK(h, call)                                                                            \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return set_err(h, BPLTV_E_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_),   \
                           __FILE__, __LINE__);                                                     \
    } while (0)

// Run f(k, shard k) on every shard's worker thread and wait for all of them.
template <class F>
int multi_run(bpltv_t* h, F f) {
    MultiState& ms = *h->multi;
    const int n = (int)ms.shard.size();
    for (int k = 0; k < n; ++k) ms.worker[k]->post([&f, &ms, k]() -> int { return f(k, ms.shard[k]); });
    int rc = BPLTV_OK, bad = -1;
    for (int k = 0; k < n; ++k) {
        const int r = ms.worker[k]->wait();
        if (r != BPLTV_OK && rc == BPLTV_OK) { rc = r; bad = k; }
    }
    if (rc != BPLTV_OK)
        set_err(h, rc, "shard %d (images %d..%d, device %d): %s", bad, ms.lo[bad], ms.hi[bad] - 1, ms.dev[bad],
                ms.shard[bad] ? ms.shard[bad]->err.c_str() : "creation failed");
    return rc;
}

void multi_free(bpltv_t* h) {
    MultiState* ms = h->multi;
    if (!ms) return;
    const int n = (int)ms->shard.size();
    if (!ms->worker.empty()) {
        (void)multi_run(h, [ms](int k, bpltv_t* c) -> int {
            if (k < (int)ms->d_rows.size() && ms->d_rows[k]) (void)hipFree(ms->d_rows[k]);
            if (k < (int)ms->d_all.size() && ms->d_all[k]) (void)hipFree(ms->d_all[k]);
            if (c) (void)bpltv_destroy(c);
            return BPLTV_OK;
        });
    }
    for (size_t r = 0; r < ms->rep.size(); ++r) {   // replicas (not the borrowed shard 0), each on its device's thread
        if (!ms->rep[r] || (r == 0 && ms->rep_borrowed0())) continue;
        bpltv_t* c = ms->rep[r];
        ms->rep_worker[r]->post([c]() -> int { (void)bpltv_destroy(c); return 0; });
        (void)ms->rep_worker[r]->wait();
    }
    ms->rep.clear();
    ms->rep_owned.clear();
    for (ncclComm_t c : ms->comm) (void)ncclCommDestroy(c);
    ms->worker.clear();   // joins the threads
    (void)n;
    delete ms;
    h->multi = nullptr;
}

int multi_create(bpltv_t** out, int M, int N, int O, const int* devices, int nshards, int dtype) {
    if (!out) return BPLTV_E_ARG;
    *out = nullptr;
    if (M < 1 || N < 1 || O < 1 || (dtype != 64 && dtype != 32) || !devices || nshards < 1) return BPLTV_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BPLTV_E_HIP;
    for (int k = 0; k < nshards; ++k)
        if (devices[k] < 0 || devices[k] >= ndev) return BPLTV_E_ARG;
    bpltv_t* h = new (std::nothrow) bpltv_handle();
    if (!h) return BPLTV_E_NOMEM;
    *out = h;   // returned even on failure so that bpltv_last_error works; caller destroys
    h->M = M; h->N = N; h->O = O; h->device = devices[0];
    h->npx = (size_t)M * N;
    h->tot = h->npx * O;
    std::memset(&h->st, 0, sizeof(h->st));
    h->st.M = M; h->st.N = N; h->st.O = O; h->st.device = devices[0];
    h->st.last_gap = -1.0;
    MultiState* ms = new (std::nothrow) MultiState();
    if (!ms) return set_err(h, BPLTV_E_NOMEM, "out of host memory");
    h->multi = ms;
    ms->dtype = dtype;
    ms->req_dev.assign(devices, devices + nshards);
    const int n = std::min(nshards, O);   // no empty image shards; parameter sweeps use the other devices too (replicas)
    std::set<int> distinct;
    for (int k = 0; k < n; ++k) {
        int lo, hi;
        shard_range(O, n, k, &lo, &hi);
        ms->lo.push_back(lo); ms->hi.push_back(hi); ms->dev.push_back(devices[k]);
        ms->maxloc = std::max(ms->maxloc, hi - lo);
        distinct.insert(devices[k]);
        ms->shard.push_back(nullptr);
        ms->worker.emplace_back(new ShardWorker(devices[k]));
    }
    ms->d_rows.assign(n, nullptr);
    ms->d_all.assign(n, nullptr);
    ms->rows_cap.assign(n, 0);
    h->st.ngpus = (int)distinct.size();
    h->st.shards = n;
    int rc = multi_run(h, [ms, M, N, dtype](int k, bpltv_t*) -> int {
        return bpltv_create(&ms->shard[k], M, N, ms->hi[k] - ms->lo[k], ms->dev[k], dtype);
    });
    if (rc) return rc;
    if ((int)distinct.size() == n) {   // one rank per device: RCCL communicator of this process
        ms->comm.assign(n, nullptr);
        ncclResult_t r = ncclCommInitAll(ms->comm.data(), n, ms->dev.data());
        if (r != ncclSuccess) {
            ms->comm.clear();
            return set_err(h, BPLTV_E_HIP, "ncclCommInitAll over %d devices failed: %s", n, ncclGetErrorString(r));
        }
        int cnt = 0;   // what RCCL itself reports for the communicator (bench.py / the tests print it)
        if (ncclCommCount(ms->comm[0], &cnt) == ncclSuccess) h->st.nccl_ranks = cnt;
    }
    return BPLTV_OK;
}

int multi_stats(bpltv_t* h) {   // aggregate the shards' statistics into h->st
    MultiState& ms = *h->multi;
    bpltv_stats_t a = ms.shard[0]->st;
    a.O = h->O; a.device = ms.dev[0];
    for (size_t k = 1; k < ms.shard.size(); ++k) {
        const bpltv_stats_t& b = ms.shard[k]->st;
        a.tiles += b.tiles;
        a.launches = std::max(a.launches, b.launches);
        a.pdhg_ms = std::max(a.pdhg_ms, b.pdhg_ms);
        a.cost_ms = std::max(a.cost_ms, b.cost_ms);
        a.adjoint_ms = std::max(a.adjoint_ms, b.adjoint_ms);
        a.algorithmic_bytes += b.algorithmic_bytes;
        a.last_gap = std::max(a.last_gap, b.last_gap);
        a.adjoint_residual = std::max(a.adjoint_residual, b.adjoint_residual);
        a.adjoint_residual_raw = std::max(a.adjoint_residual_raw, b.adjoint_residual_raw);
        a.kappa_used = std::min(a.kappa_used, b.kappa_used);
        a.adjoint_attempts = std::max(a.adjoint_attempts, b.adjoint_attempts);
        a.iterations = std::max(a.iterations, b.iterations);
    }
    a.ngpus = h->st.ngpus; a.shards = h->st.shards; a.nccl_ranks = h->st.nccl_ranks;
    a.collective = h->st.collective; a.collective_ms = h->st.collective_ms;
    a.total_ms = h->st.total_ms;
    h->st = a;
    return BPLTV_OK;
}

// Run f(r, replica r) on the worker thread of replica r's device, r < cnt, and wait for all of them.
template <class F>
int rep_run(bpltv_t* h, int cnt, F f) {
    MultiState& ms = *h->multi;
    for (int r = 0; r < cnt; ++r) ms.rep_worker[r]->post([&f, &ms, r]() -> int { return f(r, ms.rep[r]); });
    int rc = BPLTV_OK, bad = -1;
    for (int r = 0; r < cnt; ++r) {
        const int q = ms.rep_worker[r]->wait();
        if (q != BPLTV_OK && rc == BPLTV_OK) { rc = q; bad = r; }
    }
    if (rc != BPLTV_OK)
        set_err(h, rc, "replica %d (all %d images, device %d): %s", bad, h->O, ms.req_dev[bad], ms.rep[bad] ? ms.rep[bad]->err.c_str() : "creation failed");
    return rc;
}

// Make sure `want` replicas exist and hold the dataset.  A replica is an ordinary single-device handle over all O
// images; the dataset comes from the shards' resident copies through the host (once per bpltv_set_data -- the library
// keeps no host pointer of the caller's).
int multi_replicas_ready(bpltv_t* h, int want) {
    MultiState& ms = *h->multi;
    const int n = (int)ms.shard.size();
    const int have = (int)ms.rep.size();
    if (want > (int)ms.req_dev.size()) return set_err(h, BPLTV_E_ARG, "replicas: %d wanted, %zu devices requested", want, ms.req_dev.size());
    if (want > have) {
        for (int r = have; r < want; ++r) {
            ms.rep.push_back(nullptr);
            if (r < n) ms.rep_worker.push_back(ms.worker[r].get());
            else {
                ms.rep_owned.emplace_back(new ShardWorker(ms.req_dev[r]));
                ms.rep_worker.push_back(ms.rep_owned.back().get());
            }
        }
        const int M = h->M, N = h->N, O = h->O;
        int rc = rep_run(h, want, [&ms, M, N, O, n, have](int r, bpltv_t*) -> int {
            if (r < have) return BPLTV_OK;
            if (r == 0 && n == 1) { ms.rep[0] = ms.shard[0]; return BPLTV_OK; }   // one shard holds every image already
            const int rc2 = bpltv_create(&ms.rep[r], M, N, O, ms.req_dev[r], ms.dtype);
            if (rc2 == BPLTV_OK) ms.rep[r]->opt = ms.shard[0]->opt;
            return rc2;
        });
        if (rc) {   // leave no half-made replica behind: a later call starts from `have` again
            for (int r = want - 1; r >= have; --r) {
                bpltv_t* c = ms.rep[r];
                if (c && !(r == 0 && n == 1)) {
                    ms.rep_worker[r]->post([c]() -> int { (void)bpltv_destroy(c); return 0; });
                    (void)ms.rep_worker[r]->wait();
                }
                ms.rep.pop_back();
                ms.rep_worker.pop_back();
                if (r >= n) ms.rep_owned.pop_back();
            }
            return rc;
        }
        ms.rep_data = false;
        std::set<int> distinct(ms.dev.begin(), ms.dev.end());
        for (int r = 0; r < want; ++r) distinct.insert(ms.req_dev[r]);
        h->st.ngpus = (int)distinct.size();
    }
    if (!ms.rep_data) {
        if (!h->has_data) return set_err(h, BPLTV_E_NODATA, "bpltv_set_data has not been called");
        std::vector<double> ub(h->tot), f(h->tot);
        const size_t npx = h->npx;
        int rc = multi_run(h, [&](int k, bpltv_t* c) -> int {
            HIPCHK(c, hipMemcpy(ub.data() + ms.lo[k] * npx, c->d_ubar, c->tot * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(f.data() + ms.lo[k] * npx, c->d_f, c->tot * sizeof(double), hipMemcpyDeviceToHost));
            return BPLTV_OK;
        });
        if (rc) return rc;
        rc = rep_run(h, (int)ms.rep.size(), [&](int r, bpltv_t* c) { return (r == 0 && ms.rep_borrowed0()) ? BPLTV_OK : bpltv_set_data(c, ub.data(), f.data()); });
        if (rc) return rc;
        ms.rep_data = true;
    }
    return BPLTV_OK;
}

int multi_set_data(bpltv_t* h, const double* ubar, const double* f) {
    if (!ubar || !f) return set_err(h, BPLTV_E_ARG, "set_data: null pointer");
    MultiState& ms = *h->multi;
    const size_t npx = h->npx;
    int rc = multi_run(h, [&](int k, bpltv_t* c) { return bpltv_set_data(c, ubar + ms.lo[k] * npx, f + ms.lo[k] * npx); });
    if (rc) return rc;
    h->has_data = true;
    ms.rep_data = false;
    if (!ms.rep.empty()) {   // replicas exist already: they take the whole dataset from the caller's arrays
        rc = rep_run(h, (int)ms.rep.size(), [&](int r, bpltv_t* c) { return (r == 0 && ms.rep_borrowed0()) ? BPLTV_OK : bpltv_set_data(c, ubar, f); });
        if (rc) return rc;
        ms.rep_data = true;
    }
    return BPLTV_OK;
}

int multi_denoise(bpltv_t* h, const double* alpha, int am, int an, const bpltv_params* pp, double* u_out, int slices = 1) {
    WallTimer wt;
    MultiState& ms = *h->multi;
    const size_t npx = h->npx;
    int rc = multi_run(h, [&](int k, bpltv_t* c) {
        double* uo = u_out ? u_out + ms.lo[k] * npx : nullptr;
        return slices == 3 ? bpltv_sumregs_denoise(c, alpha, am, an, pp, uo) : bpltv_denoise(c, alpha, am, an, pp, uo);
    });
    if (rc) return rc;
    h->has_result = true;
    h->st.collective = 0; h->st.collective_ms = 0.0;
    h->st.total_ms = wt.ms();
    return multi_stats(h);
}

// tv_op_learning_function over the shards: every device evaluates its images, then ONE collective on the
// [cost, grad...] vector.  out: host, 1 + am*an doubles.
int multi_evaluate(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* pp,
                   double* u_out, double* out, int slices = 1) {
    WallTimer wt;
    MultiState& ms = *h->multi;
    const int n = (int)ms.shard.size();
    const size_t npx = h->npx, P = (size_t)am * an * slices, np = 1 + P;
    const bpltv_params p = resolve(pp);
    int rc = multi_run(h, [&](int k, bpltv_t* c) {
        double* uo = u_out ? u_out + ms.lo[k] * npx : nullptr;
        return slices == 3 ? sr_evaluate_common(c, alpha, am, an, delta, pp, uo, nullptr)
                           : evaluate_common(c, alpha, am, an, delta, pp, uo, nullptr, nullptr);
    });
    if (rc) return rc;
    const bool amap = (am == h->M && an == h->N) && !(h->M == 1 && h->N == 1);
    const bool ordered = p.deterministic != 0 && !amap;
    const bool rccl = !ms.comm.empty();
    WallTimer ct;
    if (ordered) {
        // per-image rows [cost_k, grad_k...] of every shard, added in global image order (plain left-to-right
        // sums == what sum_final_kernel does on one device): totals bitwise independent of the sharding
        std::vector<double> rows((size_t)n * ms.maxloc * np, 0.0);
        if (rccl) {
            const size_t need = (size_t)ms.maxloc * np;
            rc = multi_run(h, [&](int k, bpltv_t* c) -> int {
                if (ms.rows_cap[k] < need) {
                    if (ms.d_rows[k]) (void)hipFree(ms.d_rows[k]);
                    if (ms.d_all[k]) (void)hipFree(ms.d_all[k]);
                    ms.d_rows[k] = ms.d_all[k] = nullptr;
                    ms.rows_cap[k] = 0;
                    const int arc = alloc_all(c, {{(void**)&ms.d_rows[k], need * sizeof(double)},
                                                  {(void**)&ms.d_all[k], need * n * sizeof(double)}}, "all-gather buffers");
                    if (arc) return arc;
                    ms.rows_cap[k] = need;
                }
                hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, c->stream, c->d_perimg,
                                   c->d_red, c->O, (int)P, ms.maxloc, ms.d_rows[k]);
                HIPCHK(c, hipGetLastError());
                return BPLTV_OK;
            });
            if (rc) return rc;
            NCCLCHK(h, ncclGroupStart());
            for (int k = 0; k < n; ++k)
                NCCLCHK(h, ncclAllGather(ms.d_rows[k], ms.d_all[k], need, ncclDouble, ms.comm[k], ms.shard[k]->stream));
            NCCLCHK(h, ncclGroupEnd());
            rc = multi_run(h, [&](int k, bpltv_t* c) -> int {
                if (k == 0)
                    HIPCHK(c, hipMemcpyAsync(rows.data(), ms.d_all[0], rows.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                return BPLTV_OK;
            });
            if (rc) return rc;
            h->st.collective = 2;
        } else {
            rc = multi_run(h, [&](int k, bpltv_t* c) { return bpltv_per_image(c, rows.data() + (size_t)k * ms.maxloc * np); });
            if (rc) return rc;
            h->st.collective = 3;
        }
        // cost: sum_final_kernel adds per-image values in image order; gradient entry q likewise
        for (size_t e = 0; e < np; ++e) out[e] = 0.0;
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < ms.hi[k] - ms.lo[k]; ++i) {
                const double* r = rows.data() + ((size_t)k * ms.maxloc + i) * np;
                for (size_t e = 0; e < np; ++e) out[e] += r[e];
            }
    } else if (rccl) {
        // ONE ncclAllReduce(sum, f64) over xGMI, in place on the shards' partial vectors (every rank of this
        // single-process communicator is driven from the caller's thread, grouped)
        NCCLCHK(h, ncclGroupStart());
        for (int k = 0; k < n; ++k)
            NCCLCHK(h, ncclAllReduce(ms.shard[k]->d_partial, ms.shard[k]->d_partial, np, ncclDouble, ncclSum, ms.comm[k],
                                     ms.shard[k]->stream));
        NCCLCHK(h, ncclGroupEnd());
        rc = multi_run(h, [&](int k, bpltv_t* c) -> int {
            if (k == 0)
                HIPCHK(c, hipMemcpyAsync(out, c->d_partial, np * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return BPLTV_OK;
        });
        if (rc) return rc;
        h->st.collective = 1;
    } else {
        // repeated devices (rehearsal): RCCL cannot hold two ranks on one device; same sum on the host
        std::vector<double> part((size_t)n * np);
        rc = multi_run(h, [&](int k, bpltv_t* c) -> int {
            HIPCHK(c, hipMemcpy(part.data() + (size_t)k * np, c->d_partial, np * sizeof(double), hipMemcpyDeviceToHost));
            return BPLTV_OK;
        });
        if (rc) return rc;
        for (size_t e = 0; e < np; ++e) {
            double acc = part[e];
            for (int k = 1; k < n; ++k) acc += part[(size_t)k * np + e];
            out[e] = acc;
        }
        h->st.collective = n > 1 ? 3 : 0;
    }
    h->st.collective_ms = ct.ms();
    h->has_result = true;
    h->has_per_image = !amap;
    h->last_am = am; h->last_an = an; h->last_slices = slices;
    h->st.total_ms = wt.ms();
    return multi_stats(h);
}

int multi_gradient(bpltv_t* h, const double* u, const double* ubar, const double* alpha, int am, int an, int reg,
                   const bpltv_params* pp, double* grad_out) {
    if (!u || !ubar || !grad_out || !alpha || am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "gradient: null pointer or empty shape");
    WallTimer wt;
    MultiState& ms = *h->multi;
    const int n = (int)ms.shard.size();
    const size_t npx = h->npx, P = (size_t)am * an;
    std::vector<double> g((size_t)n * P);
    int rc = multi_run(h, [&](int k, bpltv_t* c) {
        return bpltv_gradient(c, u + ms.lo[k] * npx, ubar + ms.lo[k] * npx, alpha, am, an, reg, pp, g.data() + (size_t)k * P);
    });
    if (rc) return rc;
    for (size_t e = 0; e < P; ++e) {   // gradient wrappers sum per-image terms (TVLearningFunctionVec.jl:76-82,168-173)
        double acc = g[e];
        for (int k = 1; k < n; ++k) acc += g[(size_t)k * P + e];
        grad_out[e] = acc;
    }
    h->st.total_ms = wt.ms();
    return multi_stats(h);
}

int multi_sweep(bpltv_t* h, const double* alphas, int K, int am, int an, const bpltv_params* pp, double* cost_out,
                double* u_out) {
    if (!alphas || !cost_out || K < 1) return set_err(h, BPLTV_E_ARG, "sweep: null pointer or K < 1");
    if (am < 1 || an < 1 || am > h->M || an > h->N) return set_err(h, BPLTV_E_ARG, "sweep: bad parameter shape %dx%d", am, an);
    WallTimer wt;
    MultiState& ms = *h->multi;
    const int n = (int)ms.shard.size();
    const size_t npx = h->npx;
    h->st.sweep_shards = 0;
    {
        // Which axis to split: the images (what the shards already hold: device k solves K x O_k problems) or the K
        // parameter blocks over replicas (device r solves K_r x O problems).  Automatic = the smaller largest share;
        // ties keep the image split (no second copy of the dataset).  The reference's sweeps
        // (/root/reference/src/BPLDenoising.jl:92-111,136-158,381-415) run on num_samples = 1 by default (:313), where
        // only the parameter axis can use more than one device.
        const int nrep = (int)std::min<size_t>(ms.req_dev.size(), (size_t)K);
        const long share_img = (long)K * ms.maxloc, share_par = (long)((K + nrep - 1) / nrep) * h->O;
        const bool by_par = ms.sweep_split == 2 || (ms.sweep_split == 0 && share_par < share_img);
        if (by_par && nrep >= 1 && !(n == 1 && nrep == 1)) {
            int rc = multi_replicas_ready(h, nrep);
            if (rc) return rc;
            const size_t npar = (size_t)am * an;
            rc = rep_run(h, nrep, [&](int r, bpltv_t* c) -> int {
                int lo, hi;
                shard_range(K, nrep, r, &lo, &hi);
                // parameter-major outputs: replica r's blocks [lo, hi) are contiguous in cost_out and u_out
                return bpltv_sweep(c, alphas + (size_t)lo * npar, hi - lo, am, an, pp, cost_out + lo,
                                   u_out ? u_out + (size_t)lo * h->O * npx : nullptr);
            });
            if (rc) return rc;
            bpltv_stats_t a = ms.rep[0]->st;
            for (int r = 1; r < nrep; ++r) {
                const bpltv_stats_t& b = ms.rep[r]->st;
                a.tiles += b.tiles;
                a.launches = std::max(a.launches, b.launches);
                a.pdhg_ms = std::max(a.pdhg_ms, b.pdhg_ms);
                a.algorithmic_bytes += b.algorithmic_bytes;
                a.iterations = std::max(a.iterations, b.iterations);
            }
            a.O = h->O; a.ngpus = h->st.ngpus; a.shards = h->st.shards; a.nccl_ranks = h->st.nccl_ranks;
            a.collective = 0; a.collective_ms = 0.0;
            a.sweep_shards = nrep;
            a.total_ms = wt.ms();
            h->st = a;
            return BPLTV_OK;
        }
    }
    std::vector<double> cost((size_t)n * K);
    std::vector<std::vector<double>> ubuf(n);
    int rc = multi_run(h, [&](int k, bpltv_t* c) -> int {
        const size_t Ok = (size_t)(ms.hi[k] - ms.lo[k]);
        if (u_out) ubuf[k].resize((size_t)K * Ok * npx);
        const int r = bpltv_sweep(c, alphas, K, am, an, pp, cost.data() + (size_t)k * K, u_out ? ubuf[k].data() : nullptr);
        if (r == BPLTV_OK && u_out)   // parameter-major [K][O][N][M]: this shard's images of every parameter block
            for (int q = 0; q < K; ++q)
                std::memcpy(u_out + ((size_t)q * h->O + ms.lo[k]) * npx, ubuf[k].data() + (size_t)q * Ok * npx, Ok * npx * sizeof(double));
        return r;
    });
    if (rc) return rc;
    for (int q = 0; q < K; ++q) {
        double acc = cost[q];
        for (int k = 1; k < n; ++k) acc += cost[(size_t)k * K + q];
        cost_out[q] = acc;
    }
    h->st.total_ms = wt.ms();
    return multi_stats(h);
}

int multi_unsupported(bpltv_t* h, const char* what) {
    if (h->multi->shard.size() == 1) return -1;   // one shard: forward to it
    return set_err(h, BPLTV_E_UNSUPPORTED, "%s takes a device pointer, which is ambiguous on a handle over %zu shards; use the host-array entry points",
                   what, h->multi->shard.size());
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
#pragma GCC visibility push(default)
extern "C" {

int bpltv_version(void) { return BPLTV_VERSION; }

int bpltv_default_params(bpltv_params* p) {
    if (!p) return BPLTV_E_ARG;
    std::memset(p, 0, sizeof(*p));
    p->rho = 0.0;            // /root/reference/src/TVLearningFunctionVec.jl:34
    p->tau0 = 5.0;           // :36
    p->sigma0 = 0.99 / 5;    // :37
    p->accel = 1;            // :38
    p->maxiter = 5000;       // :40
    p->delta_t = 1e-6;       // :14
    p->check_every = 0;
    p->gap_tol = 0.0;
    p->tile_iters = 0;
    p->use_graph = 1;
    p->kappa_cap = 0.0;
    p->refine = -1;
    return BPLTV_OK;
}

int bpltv_create(bpltv_t** out, int M, int N, int O, int device, int dtype) {
    if (!out) return BPLTV_E_ARG;
    *out = nullptr;
    if (M < 1 || N < 1 || O < 1 || (dtype != 64 && dtype != 32)) return BPLTV_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BPLTV_E_HIP;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return BPLTV_E_HIP;
    }
    if (device >= ndev) return BPLTV_E_ARG;
    bpltv_t* h = new (std::nothrow) bpltv_handle();
    if (!h) return BPLTV_E_NOMEM;
    h->M = M; h->N = N; h->O = O; h->device = device; h->dtype = dtype;
    h->npx = (size_t)M * N;
    h->tot = h->npx * O;
    std::memset(&h->st, 0, sizeof(h->st));
    h->st.M = M; h->st.N = N; h->st.O = O; h->st.device = device;
    h->st.last_gap = -1.0;
    h->st.ngpus = 1;
    *out = h;  // returned even on failure below so that bpltv_last_error works; caller destroys
    HIPCHK(h, hipSetDevice(device));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->ncu = prop.multiProcessorCount;
        h->st.ncu = h->ncu;
    }
    HIPCHK(h, device_streams_acquire(device, &h->stream));
    for (auto& e : h->ev) HIPCHK(h, hipEventCreate(&e));
    HIPCHK(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
    HIPCHK(h, hipMalloc((void**)&h->d_phase, 64));
    HIPCHK(h, hipMemset(h->d_phase, 0, 64));
    HIPCHK(h, hipMalloc((void**)&h->d_ubar, h->tot * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->d_f, h->tot * sizeof(double)));
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 3; ++c) HIPCHK(h, hipMalloc((void**)&h->d_state[s][c], h->tot * sizeof(double)));
    h->cur_state = h->d_state;
    h->cur_nimg = O;
    h->cur_astride = 0;
    HIPCHK(h, hipMalloc((void**)&h->d_perimg, (size_t)O * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->d_scalar, 4 * sizeof(double)));
    // LDS above 64 KB needs the opt-in attribute
    for (const Variant& V : kVariants) {
        if (V.lds > 64 * 1024)
            HIPCHK(h, hipFuncSetAttribute(V.func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)V.lds));
        if (dtype == 32 && V.lds32 > 64 * 1024)
            HIPCHK(h, hipFuncSetAttribute(V.func32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)V.lds32));
    }
    return BPLTV_OK;
}

int bpltv_create_sharded(bpltv_t** out, int M, int N, int O, const int* devices, int nshards, int dtype) {
    return multi_create(out, M, N, O, devices, nshards, dtype);
}

int bpltv_create_multi(bpltv_t** out, int M, int N, int O, int ngpus, int dtype) {
    if (!out) return BPLTV_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BPLTV_E_HIP;
    if (ngpus == 0) ngpus = ndev;   // all visible devices
    if (ngpus < 0 || ngpus > ndev) return BPLTV_E_ARG;
    std::vector<int> devs(ngpus);
    for (int k = 0; k < ngpus; ++k) devs[k] = k;
    return multi_create(out, M, N, O, devs.data(), ngpus, dtype);
}

int bpltv_destroy(bpltv_t* h) {
    if (!h) return BPLTV_OK;
    if (h->multi) {
        multi_free(h);
        delete h;
        return BPLTV_OK;
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->launcher.reset();   // joins the launcher thread (idle: every call waits for its job)
    drop_graphs(h);
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    for (auto ce : h->chain_events) (void)hipEventDestroy(ce);   // (the chain streams belong to the device: device_streams_release below)
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    if (h->d_phase) (void)hipFree(h->d_phase);
    for (auto& kv : h->tabs) (void)hipFree(kv.second);
    for (auto& kv : h->tabs32) (void)hipFree(kv.second);
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 3; ++c)
            if (h->f32_state[s][c]) (void)hipFree(h->f32_state[s][c]);
    if (h->f32_f) (void)hipFree(h->f32_f);
    if (h->f32_alpha) (void)hipFree(h->f32_alpha);
    void* ptrs[] = {h->d_ubar, h->d_f, h->d_alpha, h->d_partial, h->d_red, h->d_perimg, h->d_scalar, h->d_coef,
                    h->d_band4, h->d_bcr, h->d_L, h->d_invF, h->d_invB, h->d_L1, h->d_dump, h->d_Lm, h->d_spill, h->d_p, h->d_r, h->d_gpix, h->d_resn, h->d_fail, h->d_u2, h->d_ubar2};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 3; ++c) {
            if (h->d_state[s][c]) (void)hipFree(h->d_state[s][c]);
            if (h->d_sweep[s][c]) (void)hipFree(h->d_sweep[s][c]);
        }
    if (h->d_sweep_cost) (void)hipFree(h->d_sweep_cost);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    h->hb.release();
    h->nd.release();
    h->nd_sr.release();
    h->nd_sr_lu.release();
    h->hb_sr.release();
    h->lu_sr.release();
    if (h->d_srdiagU) (void)hipFree(h->d_srdiagU);
    for (auto& kv : h->sr_graphs)
        for (auto e : kv.second) (void)hipGraphExecDestroy(e);
    for (void* q : {(void*)h->d_srcoef, (void*)h->d_srdiag, (void*)h->d_srw, (void*)h->d_srgpix})
        if (q) (void)hipFree(q);
    for (int s2 = 0; s2 < 2; ++s2)
        for (int c = 0; c < 7; ++c)
            if (h->d_sr[s2][c]) (void)hipFree(h->d_sr[s2][c]);
    if (h->stream) device_streams_release(h->device);
    delete h;
    return BPLTV_OK;
}

static int set_data_impl(bpltv_t* h, const double* ubar, const double* f, hipMemcpyKind kind) {
    if (!h) return BPLTV_E_ARG;
    if (!ubar || !f) return set_err(h, BPLTV_E_ARG, "set_data: null pointer");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_ubar, ubar, h->tot * sizeof(double), kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_f, f, h->tot * sizeof(double), kind, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->has_data = true;
    h->has_result = false;
    h->f32_f_valid = false;
    return BPLTV_OK;
}

int bpltv_set_data(bpltv_t* h, const double* ubar, const double* f) {
    if (h && h->multi) return multi_set_data(h, ubar, f);
    return set_data_impl(h, ubar, f, hipMemcpyHostToDevice);
}

int bpltv_set_data_device(bpltv_t* h, const double* d_ubar, const double* d_f) {
    if (h && h->multi) {
        const int rc = multi_unsupported(h, "bpltv_set_data_device");
        if (rc >= 0) return rc;
        const int r = bpltv_set_data_device(h->multi->shard[0], d_ubar, d_f);
        if (r) h->err = h->multi->shard[0]->err; else h->has_data = true;
        return r;
    }
    return set_data_impl(h, d_ubar, d_f, hipMemcpyDeviceToDevice);
}

int bpltv_denoise(bpltv_t* h, const double* alpha, int am, int an, const bpltv_params* pp, double* u_out) {
    if (!h) return BPLTV_E_ARG;
    if (h->multi) return multi_denoise(h, alpha, am, an, pp, u_out);
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p = resolve(pp);
    if (int prc = check_params(h, p)) return prc;
    int rc = upload_alpha(h, alpha, am, an);
    if (rc) return rc;
    rc = run_pdhg(h, p);
    if (rc) return rc;
    if (u_out) {
        HIPCHK(h, hipMemcpyAsync(u_out, h->d_state[h->result_buf][0], h->tot * sizeof(double), hipMemcpyDeviceToHost,
                                 h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->st.total_ms = wt.ms();
    return BPLTV_OK;
}

int bpltv_denoise_device(bpltv_t* h, const double* d_alpha, int am, int an, const bpltv_params* pp) {
    if (!h) return BPLTV_E_ARG;
    if (h->multi) {
        const int rc = multi_unsupported(h, "bpltv_denoise_device");
        if (rc >= 0) return rc;
        const int r = bpltv_denoise_device(h->multi->shard[0], d_alpha, am, an, pp);
        if (r) h->err = h->multi->shard[0]->err; else { h->has_result = true; multi_stats(h); }
        return r;
    }
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p = resolve(pp);
    if (int prc = check_params(h, p)) return prc;
    int rc = upload_alpha_device(h, d_alpha, am, an);
    if (rc) return rc;
    rc = run_pdhg(h, p);
    if (rc) return rc;
    h->st.total_ms = wt.ms();
    return BPLTV_OK;
}

int bpltv_evaluate(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* p,
                   double* u_out, double* cost_out, double* grad_out) {
    if (!h) return BPLTV_E_ARG;
    if (!cost_out || !grad_out) return set_err(h, BPLTV_E_ARG, "evaluate: null output pointer");
    if (am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "alpha: empty shape");
    std::vector<double> part(1 + (size_t)am * an);
    int rc = h->multi ? multi_evaluate(h, alpha, am, an, delta, p, u_out, part.data())
                      : evaluate_common(h, alpha, am, an, delta, p, u_out, nullptr, part.data());
    if (rc) return rc;
    *cost_out = part[0];
    std::memcpy(grad_out, part.data() + 1, sizeof(double) * (size_t)am * an);
    return BPLTV_OK;
}

int bpltv_sumregs_default_params(bpltv_params* p) {
    const int rc = bpltv_default_params(p);   // same solver block (SumRegsLearningFunction.jl:39-55 == TVLearningFunctionVec.jl:33-43)
    if (rc) return rc;
    p->delta_t = 1e-3;                        // SumRegsLearningFunction.jl:8,22
    return BPLTV_OK;
}

int bpltv_sumregs_denoise(bpltv_t* h, const double* alpha, int am, int an, const bpltv_params* pp, double* u_out) {
    if (!h) return BPLTV_E_ARG;
    if (h->multi) return multi_denoise(h, alpha, am, an, pp, u_out, 3);
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p;
    if (pp) p = *pp; else bpltv_sumregs_default_params(&p);
    if (int prc = check_params(h, p)) return prc;
    int rc = sr_upload_alpha(h, alpha, am, an);
    if (rc) return rc;
    rc = run_sr_pdhg(h, p);
    if (rc) return rc;
    if (u_out) {
        HIPCHK(h, hipMemcpyAsync(u_out, h->d_sr[h->sr_result_buf][0], h->tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->st.total_ms = wt.ms();
    return BPLTV_OK;
}

int bpltv_sumregs_evaluate(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* pp, double* u_out,
                           double* cost_out, double* grad_out) {
    if (!h) return BPLTV_E_ARG;
    if (!cost_out || !grad_out) return set_err(h, BPLTV_E_ARG, "evaluate: null output pointer");
    if (!alpha || am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "alpha: null pointer or empty shape");
    bpltv_params p;
    if (pp) p = *pp; else bpltv_sumregs_default_params(&p);
    std::vector<double> part(1 + 3 * (size_t)am * an);
    int rc = h->multi ? multi_evaluate(h, alpha, am, an, delta, &p, u_out, part.data(), 3)
                      : sr_evaluate_common(h, alpha, am, an, delta, &p, u_out, part.data());
    if (rc) return rc;
    *cost_out = part[0];
    std::memcpy(grad_out, part.data() + 1, sizeof(double) * 3 * (size_t)am * an);
    return BPLTV_OK;
}

int bpltv_evaluate_partial(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* p,
                           double* u_out, double* partial_out) {
    if (!h) return BPLTV_E_ARG;
    if (!partial_out) return set_err(h, BPLTV_E_ARG, "evaluate_partial: null output pointer");
    if (h->multi) {   // the "partial" of a multi-device handle is the total over its shards
        if (!alpha || am < 1 || an < 1) return set_err(h, BPLTV_E_ARG, "alpha: null pointer or empty shape");
        return multi_evaluate(h, alpha, am, an, delta, p, u_out, partial_out);
    }
    return evaluate_common(h, alpha, am, an, delta, p, u_out, nullptr, partial_out);
}

int bpltv_evaluate_device(bpltv_t* h, const double* alpha, int am, int an, double delta, const bpltv_params* p,
                          double* d_partial) {
    if (!h) return BPLTV_E_ARG;
    if (!d_partial) return set_err(h, BPLTV_E_ARG, "evaluate_device: null output pointer");
    if (h->multi) {
        const int rc = multi_unsupported(h, "bpltv_evaluate_device");
        if (rc >= 0) return rc;
        const int r = bpltv_evaluate_device(h->multi->shard[0], alpha, am, an, delta, p, d_partial);
        if (r) h->err = h->multi->shard[0]->err; else { h->has_result = true; multi_stats(h); }
        return r;
    }
    return evaluate_common(h, alpha, am, an, delta, p, nullptr, d_partial, nullptr);
}

int bpltv_u_device(bpltv_t* h, const double** d_u) {
    if (!h || !d_u) return BPLTV_E_ARG;
    if (h->multi) {
        const int rc = multi_unsupported(h, "bpltv_u_device");
        if (rc >= 0) return rc;
        const int r = bpltv_u_device(h->multi->shard[0], d_u);
        if (r) h->err = h->multi->shard[0]->err;
        return r;
    }
    if (h->last_is_sr && h->sr_has_result) { *d_u = h->d_sr[h->sr_result_buf][0]; return BPLTV_OK; }
    if (!h->has_result) return set_err(h, BPLTV_E_NODATA, "no solve has been run yet");
    *d_u = h->d_state[h->result_buf][0];
    return BPLTV_OK;
}

int bpltv_copy_u_device(bpltv_t* h, double* d_dst) {
    if (!h || !d_dst) return BPLTV_E_ARG;
    if (h->multi) {
        const int rc = multi_unsupported(h, "bpltv_copy_u_device");
        if (rc >= 0) return rc;
        const int r = bpltv_copy_u_device(h->multi->shard[0], d_dst);
        if (r) h->err = h->multi->shard[0]->err;
        return r;
    }
    const bool sr = h->last_is_sr && h->sr_has_result;
    if (!sr && !h->has_result) return set_err(h, BPLTV_E_NODATA, "no solve has been run yet");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(d_dst, sr ? h->d_sr[h->sr_result_buf][0] : h->d_state[h->result_buf][0], h->tot * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BPLTV_OK;
}

int bpltv_duality_gap(bpltv_t* h, double* gap_out) {
    if (!h || !gap_out) return BPLTV_E_ARG;
    if (h->multi) {
        MultiState& ms = *h->multi;
        const int rc = multi_run(h, [&](int k, bpltv_t* c) { return bpltv_duality_gap(c, gap_out + ms.lo[k]); });
        return rc ? rc : multi_stats(h);
    }
    if (!(h->last_is_sr ? h->sr_has_result : h->has_result)) return set_err(h, BPLTV_E_NODATA, "no solve has been run yet");
    HIPCHK(h, hipSetDevice(h->device));
    double gmax = 0.0;
    int rc = compute_gap(h, gap_out, &gmax);
    if (rc) return rc;
    h->st.last_gap = gmax;
    return BPLTV_OK;
}

int bpltv_grad_fwd(bpltv_t* h, const double* x, double* d1, double* d2) {
    if (!h || !x || !d1 || !d2) return BPLTV_E_ARG;
    if (h->multi)   // one image: the first shard's device
        return multi_run(h, [&](int k, bpltv_t* c) { return k == 0 ? bpltv_grad_fwd(c, x, d1, d2) : BPLTV_OK; });
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = h->npx;
    int rc = ensure(h, &h->d_red, &h->red_cap, 3 * n);
    if (rc) return rc;
    double* b = h->d_red;
    HIPCHK(h, hipMemcpyAsync(b, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(grad_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, b, h->M, h->N,
                       b + n, b + 2 * n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(d1, b + n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(d2, b + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BPLTV_OK;
}

int bpltv_grad_fwd_adjoint(bpltv_t* h, const double* y1, const double* y2, double* out) {
    if (!h || !y1 || !y2 || !out) return BPLTV_E_ARG;
    if (h->multi)
        return multi_run(h, [&](int k, bpltv_t* c) { return k == 0 ? bpltv_grad_fwd_adjoint(c, y1, y2, out) : BPLTV_OK; });
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = h->npx;
    int rc = ensure(h, &h->d_red, &h->red_cap, 3 * n);
    if (rc) return rc;
    double* b = h->d_red;
    HIPCHK(h, hipMemcpyAsync(b, y1, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(b + n, y2, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(grad_fwd_T_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, b, b + n, h->M,
                       h->N, b + 2 * n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, b + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BPLTV_OK;
}

int bpltv_gradient(bpltv_t* h, const double* u, const double* ubar, const double* alpha, int am, int an, int reg,
                   const bpltv_params* pp, double* grad_out) {
    if (h) h->has_per_image = false;
    if (!h) return BPLTV_E_ARG;
    if (h->multi) return multi_gradient(h, u, ubar, alpha, am, an, reg, pp, grad_out);
    if (!u || !ubar || !grad_out) return set_err(h, BPLTV_E_ARG, "gradient: null pointer");
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p = resolve(pp);
    if (int prc = check_params(h, p)) return prc;
    int rc = upload_alpha(h, alpha, am, an);
    if (rc) return rc;
    if (!h->d_u2) {
        HIPCHK(h, hipMalloc((void**)&h->d_u2, h->tot * sizeof(double)));
        HIPCHK(h, hipMalloc((void**)&h->d_ubar2, h->tot * sizeof(double)));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_u2, u, h->tot * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_ubar2, ubar, h->tot * sizeof(double), hipMemcpyHostToDevice, h->stream));
    rc = run_gradient(h, h->d_u2, h->d_ubar2, reg ? 1 : 0, p, h->d_partial + 1);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(grad_out, h->d_partial + 1, sizeof(double) * (size_t)am * an, hipMemcpyDeviceToHost,
                             h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->st.total_ms = wt.ms();
    return BPLTV_OK;
}

int bpltv_sweep(bpltv_t* h, const double* alphas, int K, int am, int an, const bpltv_params* pp, double* cost_out,
                double* u_out) {
    if (!h) return BPLTV_E_ARG;
    if (h->multi) return multi_sweep(h, alphas, K, am, an, pp, cost_out, u_out);
    if (!alphas || !cost_out || K < 1) return set_err(h, BPLTV_E_ARG, "sweep: null pointer or K < 1");
    if (am < 1 || an < 1 || am > h->M || an > h->N) return set_err(h, BPLTV_E_ARG, "sweep: bad parameter shape %dx%d", am, an);
    if (!h->has_data) return set_err(h, BPLTV_E_NODATA, "bpltv_set_data has not been called");
    WallTimer wt;
    HIPCHK(h, hipSetDevice(h->device));
    bpltv_params p = resolve(pp);
    if (int prc = check_params(h, p)) return prc;
    p.check_every = 0;  // the gap kernels address the dataset context only
    const size_t nimg = (size_t)K * h->O, npar = (size_t)am * an;
    if (h->sweep_cap < nimg) {
        drop_graphs(h);
        for (int s = 0; s < 2; ++s)
            for (int c = 0; c < 3; ++c) {
                if (h->d_sweep[s][c]) HIPCHK(h, hipFree(h->d_sweep[s][c]));
                h->d_sweep[s][c] = nullptr;
                HIPCHK(h, hipMalloc((void**)&h->d_sweep[s][c], nimg * h->npx * sizeof(double)));
            }
        if (h->d_sweep_cost) HIPCHK(h, hipFree(h->d_sweep_cost));
        HIPCHK(h, hipMalloc((void**)&h->d_sweep_cost, nimg * sizeof(double)));
        h->sweep_cap = nimg;
    }
    double amin = alphas[0];
    for (size_t e = 0; e < (size_t)K * npar; ++e) {
        if (!std::isfinite(alphas[e]) || alphas[e] < 0.0)
            return set_err(h, BPLTV_E_ARG, "sweep: alphas[%zu] = %g: parameters must be finite and >= 0", e, alphas[e]);
        if (alphas[e] < amin) amin = alphas[e];
    }
    h->alpha_min = amin;
    // all K parameter blocks live in the alpha buffer; problem k*O + i uses block k and image i
    if (h->alpha_cap < K * npar) {
        drop_graphs(h);
        int rc = ensure(h, &h->d_alpha, &h->alpha_cap, K * npar);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_alpha, alphas, K * npar * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->last_am = am;
    h->last_an = an;
    h->cur_state = h->d_sweep;
    h->cur_nimg = (int)nimg;
    h->cur_astride = (int)npar;
    int rc = run_pdhg(h, p);
    const int rb = h->result_buf;
    h->cur_state = h->d_state;
    h->cur_nimg = h->O;
    h->cur_astride = 0;
    h->has_result = false;  // the default context holds no result of this call
    if (rc) return rc;
    // loss of every problem against ubar[img % O], then summed per parameter on the host
    const int nblk = 16;
    rc = ensure(h, &h->d_red, &h->red_cap, nimg * nblk * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(cost_partial_mod_kernel, dim3(nblk, (unsigned)nimg), dim3(256), 0, h->stream, h->d_sweep[rb][0],
                       h->d_ubar, (int)h->npx, h->O, h->d_red);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, h->stream, h->d_red, nblk, (int)nimg, 0.5,
                       h->d_sweep_cost, (double*)nullptr);
    HIPCHK(h, hipGetLastError());
    std::vector<double> per(nimg);
    HIPCHK(h, hipMemcpyAsync(per.data(), h->d_sweep_cost, nimg * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (u_out)
        HIPCHK(h, hipMemcpyAsync(u_out, h->d_sweep[rb][0], nimg * h->npx * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < K; ++k) {
        double sacc = 0.0;
        for (int i = 0; i < h->O; ++i) sacc += per[(size_t)k * h->O + i];
        cost_out[k] = sacc;
    }
    h->st.total_ms = wt.ms();
    return BPLTV_OK;
}

int bpltv_per_image(bpltv_t* h, double* out) {
    if (!h || !out) return BPLTV_E_ARG;
    if (h->multi) {
        if (!h->has_per_image)
            return set_err(h, BPLTV_E_UNSUPPORTED, "per-image pieces exist after evaluate with a scalar or patch parameter only");
        MultiState& ms = *h->multi;
        const size_t W = 1 + (size_t)h->last_am * h->last_an * h->last_slices;
        return multi_run(h, [&](int k, bpltv_t* c) { return bpltv_per_image(c, out + ms.lo[k] * W); });
    }
    if (!h->has_per_image)
        return set_err(h, BPLTV_E_UNSUPPORTED, "per-image pieces exist after evaluate with a scalar or patch parameter only");
    HIPCHK(h, hipSetDevice(h->device));
    const int O = h->O, P = h->last_am * h->last_an * h->last_slices;
    std::vector<double> cost(O), g((size_t)P * O);
    HIPCHK(h, hipMemcpy(cost.data(), h->d_perimg, sizeof(double) * O, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(g.data(), h->d_red, sizeof(double) * P * O, hipMemcpyDeviceToHost));   // [patch][image]
    for (int k = 0; k < O; ++k) {
        out[(size_t)k * (1 + P)] = cost[k];
        for (int q = 0; q < P; ++q) out[(size_t)k * (1 + P) + 1 + q] = g[(size_t)q * O + k];
    }
    return BPLTV_OK;
}

int bpltv_set_option(bpltv_t* h, const char* name, double value) {
    if (!h) return BPLTV_E_ARG;
    if (!name || !std::isfinite(value)) return set_err(h, BPLTV_E_ARG, "set_option: null name or non-finite value");
    if (h->multi) {
        const std::string nm(name);
        if (nm == "sweep_split") {
            const int v = (int)value;
            if (v < 0 || v > 2) return set_err(h, BPLTV_E_ARG, "set_option(sweep_split): 0 automatic, 1 images, 2 parameters");
            h->multi->sweep_split = v;
            return BPLTV_OK;
        }
        int rc = multi_run(h, [&](int, bpltv_t* c) { return bpltv_set_option(c, nm.c_str(), value); });
        if (rc == BPLTV_OK && !h->multi->rep.empty())
            rc = rep_run(h, (int)h->multi->rep.size(), [&](int r, bpltv_t* c) { return (r == 0 && h->multi->rep_borrowed0()) ? BPLTV_OK : bpltv_set_option(c, nm.c_str(), value); });
        return rc;
    }
    if (std::string(name) == "sweep_split") return BPLTV_OK;   // single-device handle: nothing to split
    HIPCHK(h, hipSetDevice(h->device));
    const std::string nm(name);
    const int iv = (int)value;
    if (nm == "adjoint_budget_mb") {
        if (value < 0.0) return set_err(h, BPLTV_E_ARG, "set_option(adjoint_budget_mb): must be >= 0");
        h->opt.adjoint_budget_mb = value;
    } else if (nm == "sr_force_lu") {
        h->opt.sr_force_lu = iv ? 1 : 0;
    } else if (nm == "nd_leaf") {
        if (iv < 0 || iv > 4096) return set_err(h, BPLTV_E_ARG, "set_option(nd_leaf): 0 (default) or 1..4096 pixels");
        if (iv != h->opt.nd_leaf) {   // the trees are built for a leaf size: drop them
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->nd.release(); h->nd_sr.release(); h->nd_sr_lu.release();
        }
        h->opt.nd_leaf = iv;
    } else if (nm == "nd_wave") {
        h->opt.nd_wave = iv ? 1 : 0;
    } else if (nm == "nd_skinny") {
        h->opt.nd_skinny = iv ? 1 : 0;
    } else if (nm == "nd_skinny_min") {
        h->opt.nd_skinny_min = iv;
    } else if (nm == "nd_skinny2_min") {
        h->opt.nd_skinny2_min = iv;
    } else if (nm == "nd_staged") {
        h->opt.nd_staged = iv ? 1 : 0;
    } else if (nm == "hb_sync" || nm == "hb_single_stream" || nm == "hb_rw") {
        if (nm == "hb_sync" && (iv < 0 || iv > 2)) return set_err(h, BPLTV_E_ARG, "set_option(hb_sync): 0 automatic, 1 events, 2 stream memory operations");
        if (nm == "hb_rw" && iv != 0 && iv != 32 && iv != 128) return set_err(h, BPLTV_E_ARG, "set_option(hb_rw): 0, 32 or 128");
        int& f = nm == "hb_sync" ? h->opt.hb_sync : (nm == "hb_single_stream" ? h->opt.hb_single_stream : h->opt.hb_rw);
        const int nv = nm == "hb_single_stream" ? (iv ? 1 : 0) : iv;
        if (nv != f) {   // the band solvers read these when their workspace is made: drop the workspaces
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (h->band_ready && h->adj_hbm) { h->hb.release(); h->band_ready = false; }
            if (h->sr_band_ready) { h->hb_sr.release(); h->sr_band_ready = false; }
        }
        f = nv;
    } else {
        return set_err(h, BPLTV_E_ARG, "set_option: unknown option '%s'", name);
    }
    return BPLTV_OK;
}

int bpltv_stats(bpltv_t* h, bpltv_stats_t* out) {
    if (!h || !out) return BPLTV_E_ARG;
    *out = h->st;
    return BPLTV_OK;
}

const char* bpltv_last_error(bpltv_t* h) {
    if (!h) return "null handle";
    return h->err.c_str();
}

}  // extern "C"
#pragma GCC visibility pop

#ifdef BPLTV_EXPERIMENTS
// tools/chain_phase.py: the launch-start stamps of the last solve run with params.reserved[3] & 1024 (2 x 4096 ticks of 10 ns)
extern "C" __attribute__((visibility("default"))) int bpltv_debug_tlog(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bpltv::pdhg_tlog), sizeof(long long) * 2 * 4096) == hipSuccess ? 0 : 2;
}
#endif
