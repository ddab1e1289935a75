#!/usr/bin/env python3
"""bench.py -- PDHG iterations/sec of the batched TV-denoising hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1: launched by
torch.distributed.run, one rank per GPU).  One *step* = one pass of the hot path over one batch:
`denoise` of the resident batch with the reference's fixed iteration count (5000 PDHG iterations,
/root/reference/src/TVLearningFunctionVec.jl:40), inputs already in HBM, result left in HBM.

Workload (BASELINE.json configs[1] / north_star): ONE batch of 10 x 128 x 128 Float64 images
(faces_train_128_10), scalar alpha.  Scaling: "strong" (default, what north_star asks to be reported at
1/2/4/8 GPUs) -- the images of that one batch are block-sharded over the ranks (2,2,1,1,1,1,1,1 for
N = 8; images are independent ROF problems, no data-path collective; `--evaluate` adds the one
all-reduce of [cost, grad] per evaluation); value = iterations/s of THAT batch.  The replica
number (every rank solving its own 10-image batch, summed) is reported beside it as `weak_value`,
never as `value`; `--scaling weak` makes it the timed workload.  `--multi-handle` times the
single-process form instead: ONE bpltv_create_multi handle over N devices (worker thread per device
and the RCCL collective inside the library) -- the form INTEGRATION.md gives the Julia caller.

Rank 0 prints one JSON line with `roofline` (dominant kernel pdhg_tile_kernel: the binding bound --
f64 VALU issue or measured HBM traffic, whichever fraction is larger, <= 1 -- with the contractual 56 B per
pixel-iteration figure aside as `contractual_hbm_frac`) and `cpu_baseline` (the oracle's C restatement timed on
the host cores: kind "port" -- the reference is Julia and cannot run here).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def synth_batch(O, N, M, seed=20211004):
    """Synthetic stand-in for cameraman/faces: piecewise-smooth truth in [0,1],
    f = round(255*clip(truth + N(0, 0.1^2), 0, 1))/255."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
    ub = np.zeros((O, N, M))
    for k in range(O):
        img = 0.3 + 0.4 * (ii / M) * rng.random() + 0.2 * (jj / N) * rng.random()
        for _ in range(6):
            ci, cj, r = rng.random() * M, rng.random() * N, (0.05 + 0.2 * rng.random()) * min(M, N)
            img = np.where((ii - ci) ** 2 + (jj - cj) ** 2 < r * r, rng.random(), img)
        ub[k] = np.clip(img, 0, 1)
    f = np.round(255 * np.clip(ub + 0.1 * rng.standard_normal(ub.shape), 0, 1)) / 255
    return ub, f


def load_batch(name, O, N, M, seed):
    """(ubar, f, label): the reference's own images where BASELINE.json names them
    (/root/reference/src/BPLDenoising.jl:331-332 slices `[:,:,1:num_samples]` of testdataset(name),
    /root/reference/datasets/<name>/filelist.txt), read from the committed pixel fixture
    tests/golden/datasets.npz; a set with fewer pairs than O is repeated cyclically (cameraman_128_10
    holds one pair: SURVEY 8d "replicate it x10")."""
    import numpy as np
    if name == "auto":
        name = "faces_train_128_10" if (M == 128 and N == 128) else "synthetic"
    if name == "synthetic":
        ub, f = synth_batch(O, N, M, seed)
        return ub, f, "synthetic"
    from bpldenoising_amd.datasets import testdataset
    t, d = testdataset(name, npz=os.path.join(ROOT, "tests", "golden", "datasets.npz"))
    if t.shape[1:] != (N, M):
        raise SystemExit("bench.py: dataset %s is %dx%d, --size asks for %dx%d" % (name, t.shape[2], t.shape[1], M, N))
    idx = np.arange(O) % t.shape[0]
    label = name if O <= t.shape[0] else "%s (%d pairs repeated to %d images)" % (name, t.shape[0], O)
    return np.ascontiguousarray(t[idx]), np.ascontiguousarray(d[idx]), label


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, f_full, ub_full, alpha, N, M):
    """The restated CPU path (kind "port": the reference is Julia and cannot run here) as BASELINE.md
    section 2 defines it: oracle/bpltv_oracle.c compiled ON THIS HOST with -O3 -march=native
    (oracle/Makefile: libbpltv_oracle_native.so -- a second object; the -O2 -ffp-contract=off build stays
    the bit-exact checker), timed (i) on 1 thread -- what stock Julia does with the reference -- and
    (ii) on the host's cores: OpenMP over images, and over images x column blocks (two barriers per
    iteration) for several thread counts; the fastest is reported with the threads it used.  Bounded
    sample: a few seconds per leg."""
    from oracle import c_oracle as co
    big = M * N * args.images > 200000
    it1 = args.cpu_iters or (20 if big else min(args.iters, 3000))
    flags = "gcc -O3 -march=native -fopenmp"
    try:
        co.native_lib()
        run1 = lambda it, nt: co.pdhg_native(f_full, alpha, maxiter=it, nthreads=nt)
        runr = lambda it, nt, cb: co.pdhg_rows(f_full, alpha, maxiter=it, nthreads=nt, colblock=cb, native=True)
    except Exception as e:   # no compiler on this host: the checker build (gcc -O2 -mavx2 -ffp-contract=off)
        flags = "gcc -O2 -mavx2 -mfma -ffp-contract=off (native build unavailable: %s)" % type(e).__name__
        run1 = lambda it, nt: co.pdhg(f_full, alpha, maxiter=it, nthreads=nt)
        runr = lambda it, nt, cb: co.pdhg_rows(f_full, alpha, maxiter=it, nthreads=nt, colblock=cb)
    run1(2, 1)  # page in
    t1 = time.perf_counter(); run1(it1, 1); c1 = time.perf_counter() - t1
    v1, how1 = it1 / c1, flags
    if "native" in flags and "unavailable" not in flags:
        # the checker build (-O2 -mavx2 -mfma, no contraction) is sometimes the faster scalar code: time it too and
        # report the faster of the two as the 1-thread baseline, naming it
        co.pdhg(f_full, alpha, maxiter=2, nthreads=1)
        t1 = time.perf_counter(); co.pdhg(f_full, alpha, maxiter=it1, nthreads=1); c2 = time.perf_counter() - t1
        if it1 / c2 > v1:
            v1, how1 = it1 / c2, "gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp (faster here than -O3 -march=native: %.0f it/s)" % (it1 / c1)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    itn = it1 if big else min(args.iters, 2000)
    cands = []

    def timed(fn, label, nt):
        """Bounded: a short calibration run first; the full sample only if it fits ~6 s."""
        cal = max(2, itn // 40)
        t0 = time.perf_counter(); fn(cal); dt = time.perf_counter() - t0
        if dt * itn / cal <= 6.0:
            t0 = time.perf_counter(); fn(itn); dt = time.perf_counter() - t0; n_it = itn
        else:
            n_it = cal
        cands.append((n_it / dt, nt, label, n_it))

    nimg = max(1, min(ncpu, args.images))
    timed(lambda it: run1(it, nimg), "OpenMP over images", nimg)
    for nt in sorted({min(ncpu, x) for x in (16, 32, 64, 128)}):
        if nt <= nimg:
            continue
        cb = max(1, (N * args.images) // (nt * (4 if big else 1)))   # ~1 (4) column blocks per thread and pass
        cb = min(cb, N)
        timed(lambda it, nt=nt, cb=cb: runr(it, nt, cb), "OpenMP over images x blocks of %d columns" % cb, nt)
    best = max(cands)
    cpu_adj = None
    if M <= 138:
        # CPU share of one evaluation: the oracle's banded adjoint solve, one image (checker build)
        u1 = co.pdhg(f_full[:1], alpha, maxiter=min(it1, 500), nthreads=1)
        t1 = time.perf_counter(); co.gradient(alpha, u1, ub_full[:1]); cpu_adj = time.perf_counter() - t1
    return {
        "value": v1, "unit": "PDHG iterations/s of the same %dx%dx%d batch" % (args.images, N, M),
        "cores": 1, "kind": "port",
        "sample": "%d iterations of the full batch, oracle/bpltv_oracle.c (%s), 1 thread (stock Julia runs the reference serially)" % (it1, how1),
        "all_cores": {"value": best[0], "cores": best[1], "how": best[2], "iterations": best[3],
                      "tried": [{"it_per_s": round(v, 1), "threads": n, "how": h, "iterations": k} for v, n, h, k in cands]},
        "host_cpus_visible": ncpu, "host_cpus_total": os.cpu_count(), "cpu_model": cpu_model(),
        "compiler_flags": how1,                  # the build that produced `value` (1 thread)
        "compiler_flags_all_cores": flags,       # the build of the `all_cores` legs
        "adjoint_s_per_image": cpu_adj,   # the oracle's banded Cholesky + 3 refinement sweeps, one image, 1 thread
    }


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This runs before
    torch is imported or any HIP call is made (a process that has touched the GPU must never be
    replaced or forked into ranks); the child is an ordinary subprocess whose stdout (rank 0's JSON
    line) and exit code are passed through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def make_alpha(args, N, M):
    import numpy as np
    if args.alpha_map:
        jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
        return 0.11 + 0.09 * np.sin(2 * np.pi * ii / M) * np.cos(2 * np.pi * jj / N)
    return args.alpha


def tile_count_py(L, R, T):
    """tile_count of csrc/pdhg_kernels.hpp: regions of R pixels with halo T covering L pixels (image borders need no halo)."""
    if L <= R:
        return 1
    S = R - 2 * T
    a = 0
    while True:
        cs = 0 if a == 0 else (R - T) + (a - 1) * S
        oo = 0 if a == 0 else cs - T
        if oo + R >= L:
            return a + 1
        a += 1


def kernel_name(st):
    """The PDHG kernel a solve ran (stats.pdhg_variant indexes the variant table of csrc/bpltv.hip)."""
    v = st.get("pdhg_variant", 0)
    return "pdhg_rows_kernel" if v >= 19 else ("pdhg_wave_kernel" if v >= 16 else "pdhg_tile_kernel")


CLOCK_GHZ = 2.4        # /opt/skills/guides/MI355X_MICROARCH.md: max clock 2400 MHz
F64_ISSUE_CYCLES = 4   # a wave64 f64 VALU instruction occupies its SIMD for 4 cycles (half-rate; v_fma_f64 sustains 59 of the
                       # nominal 78.6 TFLOP/s in tools/mfma_bench.hip)


def roofline_of(args, st, M, N, O_local, iters, launch_us, kernel_us, f32):
    """The `roofline` object of the PDHG kernel (pdhg_tile_kernel; pdhg_rows_kernel on large images) from one solve's
    statistics.

    `bound` / `achieved` / `peak` / `frac` are those of the BINDING bound: the candidate (f64 VALU issue, HBM traffic as
    the counters see it) with the largest fraction -- <= 1 by construction.  The kernel is temporally blocked (state
    touches HBM once per tile_iters iterations), so the contractual figure of SURVEY 8d -- algorithmic 56 / 64 B per
    pixel-iteration over the launch time -- is a figure of merit that can exceed the HBM peak; it is reported aside as
    `contractual_hbm_frac`, with `kernels_in_flight` beside it, never as `frac`.
    Time base of every fraction: avg_launch_us = HIP-event time of the launch sequence / dispatches (whole-chip time
    per dispatch; with two launch chains two kernels are in flight and a dispatch holds half the workgroups).
    Counter figures are NOT measured by this run: they are the committed summaries of separate rocprofv3 --pmc passes of
    the same command (profiles/traffic.json, written by tools/refresh_profiles.py), scaled to the workgroups of a
    dispatch and labelled with their source.  VALU floor = SQ_INSTS_VALU x 4 cycles / (CUs x 4 SIMDs) / 2.4 GHz."""
    bytes_px = st["bytes_per_px_iter"]
    bytes_per_launch = bytes_px * M * N * O_local * (iters / max(st["launches"], 1))
    contractual = bytes_per_launch / (launch_us * 1e-6) / 1e9
    wl_key = "%dx%dx%d %s" % (O_local, N, M, "map" if bytes_px in (64.0, 32.0) else "scalar")
    traffic, traffic_src, valu_instr = None, None, None
    nl = -(-iters // max(st["tile_iters"], 1))
    chains = st.get("launch_chains") or max(1, round(st["launches"] / max(nl, 1)))
    tiles_disp = st["tiles"] / chains
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf) and not f32:   # the committed counters are those of the Float64 kernel
        try:
            tj = json.load(open(tf)).get("workloads", {}).get(wl_key)
            if tj and tj.get("tile_iters") == st["tile_iters"] and tj.get("tiles") in (st["tiles"], round(tiles_disp)):
                scale = tiles_disp / tj["tiles"]
                traffic = tj.get("hbm_bytes_per_launch") * scale
                valu_instr = tj.get("valu_wave_instructions_per_launch") * scale if tj.get("valu_wave_instructions_per_launch") else None
                traffic_src = "profiles/traffic.json[%s] (%s%s)" % (wl_key, tj.get("round", "?"),
                                                                    "" if scale == 1.0 else ", scaled by %.3g: %d of its %d workgroups per dispatch" % (scale, round(tiles_disp), tj["tiles"]))
        except Exception:
            traffic = None
    nit_avg = iters * chains / max(st["launches"], 1)   # iterations per dispatch; a dispatch holds tiles / chains workgroups
    computed_px_it = tiles_disp * st["region_i"] * st["region_j"] * nit_avg
    if kernel_name(st) == "pdhg_rows_kernel":
        # halo pixel rows stop once nobody reads them (pdhg_rows_kernel): row k next to a near region edge that is not
        # the image border runs k of the T iterations, row k next to such a far edge k + 1 -- T(T+1)/2 resp. T(T-1)/2
        # row-iterations saved per region (of 64 lanes) and edge
        T_, RJ = st["tile_iters"], st["region_j"]
        nTj = tile_count_py(N, RJ, T_)
        saved_rows = (nTj - 1) * (T_ * (T_ + 1) / 2 + T_ * (T_ - 1) / 2) / nTj     # per region, in row-iterations at depth T
        computed_px_it -= tiles_disp * st["region_i"] * saved_rows * (nit_avg / T_)
    useful_px_it = M * N * O_local / chains * nit_avg
    redundancy = computed_px_it / useful_px_it
    ncu = st.get("ncu") or 256
    valu_peak = ncu * 4 * CLOCK_GHZ * 1e9 / F64_ISSUE_CYCLES / 1e9       # G wave-instructions/s at the f64 rate
    cands, valu = {}, None
    if valu_instr:
        rate = valu_instr / (launch_us * 1e-6) / 1e9
        floor_us = valu_instr * F64_ISSUE_CYCLES / (ncu * 4) / (CLOCK_GHZ * 1e3)
        cands["valu_f64_issue"] = (rate, valu_peak, "G VALU wave-instructions/s (f64 rate: %d cycles per wave64 instruction)" % F64_ISSUE_CYCLES)
        valu = {"wave_instructions_per_dispatch": valu_instr, "instr_per_computed_px_iter": valu_instr * 64.0 / computed_px_it,
                "floor_us": floor_us, "frac": floor_us / launch_us, "cus": ncu, "source": traffic_src,
                # v_fma_f64 sustains 59 of the nominal 78.6 TFLOP/s on this part (tools/mfma_bench.hip, 8 independent
                # chains per lane, 1-4 waves per SIMD): against that measured rate the kernel sits this much higher
                "frac_of_measured_issue_rate": floor_us / launch_us * (78.6 / 59.0),
                "note": "upper estimate of the issue floor: every VALU instruction priced at the nominal f64 rate (4 cycles at 2.4 GHz)"}
    if traffic:
        cands["hbm"] = (traffic / (launch_us * 1e-6) / 1e9, HBM_PEAK_GBS, "GB/s")
    if cands:
        bound = max(cands, key=lambda k: cands[k][0] / cands[k][1])
        achieved, peak, unit = cands[bound]
        src = traffic_src
    else:
        # no counter entry for this plan: the compulsory state traffic (every state word read and written once per
        # launch, no halo) is what certainly crosses the HBM side -- a lower bound, <= 1 by construction
        bound, unit, peak = "hbm", "GB/s", HBM_PEAK_GBS
        achieved = bytes_px * M * N * O_local / chains / (launch_us * 1e-6) / 1e9
        src = "no counter entry in profiles/traffic.json for this plan: compulsory state traffic per launch (lower bound)"
    frac = achieved / peak
    return {"bound": bound, "kernel": kernel_name(st), "achieved": achieved, "peak": peak, "unit": unit, "frac": frac,
            "frac_useful": frac / redundancy if bound == "valu_f64_issue" else None,   # issue share spent on pixel-iterations the recurrence needs
            "bound_note": ("the largest of the candidate fractions; what is left of the dispatch time is launch / prologue / barrier "
                           "latency that neither issue nor bandwidth explains (DESIGN.md 4.1)"),
            "bound_source": src,
            "candidates": {k: v[0] / v[1] for k, v in cands.items()},
            "traffic": traffic, "traffic_source": traffic_src,
            "measured_hbm_frac": (traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "redundancy": redundancy,   # computed / useful pixel-iterations (halo recompute)
            "region": [st["region_i"], st["region_j"]], "valu_f64": valu,
            # the contractual figure of SURVEY 8d, kept aside: NOT a fraction of a bound for a temporally blocked kernel
            "contractual_hbm_frac": contractual / HBM_PEAK_GBS, "contractual_hbm_gbs": contractual,
            "contractual_note": "algorithmic bytes (%g B per pixel-iteration) / dispatch time / 8 TB/s; the fused launch moves those bytes once per %d iterations" % (bytes_px, st["tile_iters"]),
            "kernels_in_flight": max(1, round(kernel_us / launch_us)) if kernel_us else chains,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": launch_us,
            "avg_kernel_us_serialized": kernel_us,
            "launch_chains": chains,   # concurrent chains of image groups: avg_launch_us = whole time / all dispatches
            "overlap_note": (None if chains == 1 else
                             "%d launch chains run concurrently on two hardware queues: a rocprofv3 kernel trace shows ~%d kernels in flight, "
                             "each about avg_kernel_us_serialized long (a little longer while overlapped), so dispatches x kernel duration / %d ~ the step time"
                             % (chains, chains, chains)),
            "bytes_per_px_iter": bytes_px}


def extra_workload(TVSolver, torch, name, O, size, iters, steps, alpha_map, evaluate_once, data, eval_iters=None):
    """One more workload measured outside the timed region of the default run (N = 1), so that the driver's record
    holds it: the reference's default num_samples = 1 (/root/reference/src/BPLDenoising.jl:313) and the per-GPU
    share of BASELINE config 5.  Inputs resident in HBM; `value` = PDHG iterations/s of that batch."""
    import numpy as np

    class A:   # the few fields make_alpha / roofline_of read
        pass
    a = A(); a.alpha_map = alpha_map; a.alpha = 0.1
    ub, f, label = load_batch(data, O, size, size, 20211004)
    alpha = make_alpha(a, size, size)
    s = TVSolver(size, size, O, device=torch.cuda.current_device())
    t_ub, t_f = torch.from_numpy(ub).cuda(), torch.from_numpy(f).cuda()
    torch.cuda.synchronize()
    s.set_data_device(t_ub.data_ptr(), t_f.data_ptr())
    if alpha_map:   # the parameter map resident in HBM too (8 MiB for 1024^2): no host array in the timed region
        t_alpha = torch.from_numpy(np.ascontiguousarray(alpha)).cuda()
        torch.cuda.synchronize()
        step = lambda: s.denoise_device(t_alpha.data_ptr(), size, size, maxiter=iters)
    else:
        step = lambda: s.denoise(alpha, fetch=False, maxiter=iters)
    step()
    torch.cuda.synchronize()
    ev_ms, ev_l = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        st = s.stats(); ev_ms += st["pdhg_ms"]; ev_l += st["launches"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launch_us = 1e3 * ev_ms / max(ev_l, 1)
    out = {"workload": "%dx%dx%d f64, %s alpha%s, %d PDHG iterations per step, data %s" % (
               O, size, size, "per-pixel" if alpha_map else "scalar", " (resident in HBM)" if alpha_map else "", iters, label),
           "value": steps * iters / dt, "unit": "PDHG iterations/s of that batch", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "value_from_events": steps * iters / (1e-3 * ev_ms) if ev_ms else None,   # the same from the HIP events of the launch sequence
           "tile_iters": st["tile_iters"], "tiles_per_launch": st["tiles"],
           "roofline": roofline_of(a, st, size, size, O, iters, launch_us, None, False)}
    if evaluate_once:
        part = torch.zeros(1 + (size * size if alpha_map else 1), dtype=torch.float64, device="cuda")
        best = None
        for _ in range(2):   # the first call allocates the adjoint workspace
            t1 = time.perf_counter()
            s.evaluate_device(alpha, 0.1, part.data_ptr(), maxiter=eval_iters or iters)
            torch.cuda.synchronize()
            e_ms = 1e3 * (time.perf_counter() - t1)
            s3 = s.stats()
            if best is None or e_ms < best["evaluate_ms"]:
                best = {"evaluate_ms": e_ms, "pdhg_iterations": eval_iters or iters, "pdhg_ms": s3["pdhg_ms"], "adjoint_ms": s3["adjoint_ms"],
                        "adjoint_method": s3["adjoint_method"], "adjoint_residual": s3["adjoint_residual"],
                        "adjoint_chunks": s3["adjoint_chunks"]}
        out["learning_function"] = best
    s.close()
    del t_ub, t_f
    torch.cuda.empty_cache()
    return out


def outer_loop_extra():
    """The unit the reference's user waits for: one whole run of the outer trust-region loop (bilevel_learn,
    /root/reference/src/TRBox.jl:192-273, driven as scalar_bilevel_tv_learn does, /root/reference/src/BPLDenoising.jl:316-336)
    on faces_train_128_10 through the reference-named entry point -- host arrays in, (u, cost, grad) out per evaluation,
    i.e. the PCIe-inclusive drop-in path: every evaluation is 5000 PDHG iterations + loss + adjoint gradient of 10 images."""
    import bpldenoising_amd as B
    ub, f, label = load_batch("faces_train_128_10", 10, 128, 128, 20211004)
    B.learning_function.clear_cache()
    t0 = time.perf_counter()
    x, u, hist = B.trbox.bilevel_learn((ub, f), B.tv_op_learning_function, 0.1, 0.1, maxiter=20)
    dt_first = time.perf_counter() - t0          # with the one-time costs: handle, dataset upload, adjoint workspace, elimination tree, graphs
    t0 = time.perf_counter()
    x, u, hist = B.trbox.bilevel_learn((ub, f), B.tv_op_learning_function, 0.1, 0.1, maxiter=20)
    dt = time.perf_counter() - t0                # the same run again on the cached handle
    B.learning_function.clear_cache()
    return {"workload": "bilevel_learn (TRBox, scalar alpha, alpha0 = 0.1, Delta0 = 0.1, maxiter 20, tol 1e-5) on %s, 10 images; "
                        "host arrays through tv_op_learning_function (PCIe inclusive)" % label,
            "wall_s": dt, "first_run_wall_s": dt_first, "evaluations": len(hist) + 1, "ms_per_evaluation": 1e3 * dt / (len(hist) + 1),
            "learned_alpha": float(x), "final_cost": float(hist[-1]["function_value"]) if hist else None}


def sweep_bench(args, TVSolver, shard_range, torch):
    """`--sweep K`: generate_cost of /root/reference/src/BPLDenoising.jl:92-111 -- K parameters x the images of a dataset,
    TVDenoise each -- as ONE bpltv_sweep through a handle over --gpus devices.  The reference's default is ONE image
    (num_samples = 1, :313; cameraman_128_10 holds one pair), so the K parameter blocks are what the devices share:
    shard_range(K, n) blocks per device on replicas of the dataset.  One JSON line: solves/s (a solve = one ROF problem
    run for --iters iterations), the per-device parameter ranges, what the library reports (stats.sweep_shards)."""
    import numpy as np
    K, n = args.sweep, args.gpus
    M = N = args.size
    images = args.images if args.images != 10 or args.data != "auto" else 1     # default: the one-image case
    data = "cameraman_128_10" if (args.data == "auto" and M == 128) else args.data
    ub, f, label = load_batch(data, images, N, M, 20211004)
    iters = args.iters if args.iters != 5000 else 10000                          # TVDenoise: maxiter = 10000 (:47)
    alphas = np.linspace(0.005, 0.5, K)
    if n == 1 and not args.one_device:
        s = TVSolver(M, N, images, device=0)
    elif args.one_device:
        s = TVSolver(M, N, images, devices=[0] * n)
    else:
        s = TVSolver(M, N, images, ngpus=n)
    s.set_data(ub, f)
    for _ in range(max(1, args.warmup)):
        costs = s.sweep(alphas, maxiter=iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        costs = s.sweep(alphas, maxiter=iters)
    T = time.perf_counter() - t0
    st = s.stats()
    nsh = st["sweep_shards"]
    ranges = [list(shard_range(K, nsh, r)) for r in range(nsh)] if nsh else None
    out = {"metric": "parameter-sweep ROF solves/s (extra mode; BASELINE's metric is the default run)",
           "value": args.steps * K * images / T, "unit": "solves/s (one solve = %d PDHG iterations of one %dx%d image)" % (iters, N, M),
           "n_gpus": n, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": 1e3 * T / args.steps,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": label,
           "config": {"workload": "bpltv_sweep: %d scalar parameters x %d image(s) %dx%d, %d iterations each" % (K, images, N, M, iters)},
           "sweep": {"K": K, "images": images, "split": "parameters over replicas" if nsh else ("images over shards" if st["shards"] > 1 else "single device"),
                     "sweep_shards": nsh, "parameter_ranges_per_device": ranges, "image_shards": st["shards"],
                     "devices_distinct": st["ngpus"], "pdhg_event_ms_max_over_devices": st["pdhg_ms"],
                     "pdhg_iterations_per_s_of_the_batch": iters / (1e-3 * st["pdhg_ms"]) if st["pdhg_ms"] else None,
                     "cost_min_at": float(alphas[int(np.argmin(costs))])}}
    print(json.dumps(out), flush=True)
    s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5000, help="PDHG iterations per step (reference: 5000)")
    ap.add_argument("--images", type=int, default=10)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--alpha-map", action="store_true", help="spatially varying alpha (64 B/px/iter)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): ONE --images batch sharded over the ranks (north_star); weak: every rank its own batch")
    ap.add_argument("--multi-handle", action="store_true",
                    help="one process, one bpltv_create_multi handle over --gpus devices (the Julia drop-in form); "
                         "run as `python bench.py --gpus N --multi-handle`, no launcher")
    ap.add_argument("--sweep", type=int, default=0, metavar="K",
                    help="forward-only parameter sweep instead of the headline workload: K scalar parameters x the dataset's images x "
                         "--iters iterations through ONE handle over --gpus devices (bpltv_sweep; one-image sets split the parameters "
                         "over replicas); reports solves/s -- an extra mode, not BASELINE's metric; no launcher")
    ap.add_argument("--evaluate", action="store_true", help="time full evaluate (loss + adjoint gradient + all-reduce)")
    ap.add_argument("--tile-iters", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--f32", action="store_true",
                    help="opt-in single-precision PDHG (bpltv_create dtype = 32): narrower than the reference's Float64, "
                         "never the headline; the line then says dtype f32 and counts 28/32 B per pixel-iteration")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads / weak_value measured outside the timed region")
    ap.add_argument("--chains", type=int, default=0, help="independent launch chains (0 = library default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a 1-GPU rehearsal)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank / shard uses device 0")
    ap.add_argument("--cpu-iters", type=int, default=0, help="iterations of the CPU sample (0 = auto)")
    ap.add_argument("--data", default="auto",
                    help="auto (faces_train_128_10 at --size 128, else synthetic) | faces_train_128_10 | "
                         "faces_val_128_10 | cameraman_128_10 | circle_128_10 | synthetic")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.multi_handle and not args.sweep:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not args.multi_handle and not args.sweep:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with that many ranks" % args.gpus, file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    # the native library normally travels prebuilt; build it (rank 0) if this is a bare checkout
    lib_so = os.path.join(ROOT, "bpldenoising_amd", "libbpltv.so")
    if not os.path.exists(lib_so):
        if local_rank == 0:
            import __graft_entry__ as ge
            ge.build()
        else:
            t_wait = time.time()
            while not os.path.exists(lib_so) and time.time() - t_wait < 600:
                time.sleep(1.0)
            time.sleep(2.0)
    from bpldenoising_amd import TVSolver, shard_range

    M = N = args.size
    gloo = world > 1 and args.backend != "nccl"
    kw = dict(maxiter=args.iters, tile_iters=args.tile_iters, use_graph=0 if args.no_graph else 1)
    if args.variant:
        kw["variant"] = args.variant
    if args.chains:
        kw["chains"] = args.chains
    alpha = make_alpha(args, N, M)
    npar = 1 + (M * N if args.alpha_map else 1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup, stats_of=None):
        """W untimed steps, then exactly `steps` steps between two fences; MAX over ranks."""
        for _ in range(warmup):
            step()
        fence()
        ev_ms, ev_l, ev_steps = 0.0, 0, []
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if stats_of is not None:
                st = stats_of()
                ev_ms += st["pdhg_ms"]; ev_steps.append(st["pdhg_ms"]); ev_l += st["launches"]   # HIP events, library stream
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else "cuda")
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), ev_ms, ev_l, ev_steps

    def make_solver(ub, f, O_local):
        s = TVSolver(M, N, O_local, device=local_rank, dtype=32 if args.f32 else 64)
        t_ub, t_f = torch.from_numpy(ub).cuda(), torch.from_numpy(f).cuda()
        torch.cuda.synchronize()
        s.set_data_device(t_ub.data_ptr(), t_f.data_ptr())  # inputs resident in HBM (copied device to device)
        return s

    def make_step(s, part):
        def step():
            if args.evaluate:
                part.zero_()
                if s is not None:
                    s.evaluate_device(alpha, 0.1, part.data_ptr(), **kw)
                if world > 1:
                    if gloo:
                        pc = part.cpu(); dist.all_reduce(pc); part.copy_(pc)
                    else:
                        dist.all_reduce(part)
            elif s is not None:
                s.denoise(alpha, fetch=False, **kw)
        return step

    if args.sweep:
        sweep_bench(args, TVSolver, shard_range, torch)
        return

    multi_info = None
    if args.multi_handle:
        # ---- ONE process, ONE handle over args.gpus devices (bpltv_create_multi): the drop-in form ----------------
        if rank != 0:   # launched under torchrun anyway: the other ranks only wait for rank 0 at the final barrier
            dist.barrier(); dist.destroy_process_group()
            return
        ub_full, f_full, data_label = load_batch(args.data, args.images, N, M, 20211004)
        ndev = torch.cuda.device_count()
        if args.one_device:
            solver = TVSolver(M, N, args.images, devices=[0] * args.gpus, dtype=32 if args.f32 else 64)
        else:
            solver = TVSolver(M, N, args.images, ngpus=args.gpus, dtype=32 if args.f32 else 64)
        solver.set_data(ub_full, f_full)   # each device copies its slice; resident afterwards
        O_local = args.images

        def step():
            if args.evaluate:
                solver.evaluate(alpha, 0.1, fetch_u=False, **kw)
            else:
                solver.denoise(alpha, fetch=False, **kw)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        ev_ms, ev_launches, ev_steps = 0.0, 0, []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()   # the ABI call returns when every device is quiescent (SURVEY 8b threading)
            st = solver.stats()
            ev_ms += st["pdhg_ms"]; ev_steps.append(st["pdhg_ms"]); ev_launches += st["launches"]
        T = time.perf_counter() - t0
        st = solver.stats()
        n_eff = st["shards"]
        multi_info = {"mode": "bpltv_create_multi (single process, worker thread per device, RCCL inside the library)",
                      "devices_visible": ndev, "ngpus": st["ngpus"], "shards": st["shards"],
                      "nccl_ranks_reported_by_rccl": st["nccl_ranks"],   # ncclCommCount of the handle's communicator
                      "collective": st["collective"], "collective_ms": st["collective_ms"],
                      "shard_ranges": [list(shard_range(args.images, n_eff, k)) for k in range(n_eff)]}
        scaling, world_out = "strong", args.gpus
        weak = None
        rank_info = None
    else:
        # ---- one process per GPU (torch.distributed; the driver's launch) ------------------------------------------
        if args.scaling == "weak":
            lo, O_local, seed = 0, args.images, 20211004 + rank
        else:
            lo, hi = shard_range(args.images, world, rank)
            O_local, seed = hi - lo, 20211004
        ub_full, f_full, data_label = load_batch(args.data, args.images, N, M, seed)
        ub, f = (ub_full, f_full) if args.scaling == "weak" else (ub_full[lo:lo + O_local], f_full[lo:lo + O_local])
        solver = make_solver(ub, f, O_local) if O_local > 0 else None
        part = torch.zeros(npar, dtype=torch.float64, device="cuda")
        T, ev_ms, ev_launches, ev_steps = timed(make_step(solver, part), args.steps, args.warmup,
                                                (lambda: solver.stats()) if solver is not None else None)
        scaling, world_out = args.scaling, world
        # what every rank did, and what the collective library itself says about the job
        rank_info = None
        if world > 1:
            mine = torch.tensor([float(torch.cuda.current_device()), float(lo), float(lo + O_local),
                                 ev_ms / max(args.steps, 1)], dtype=torch.float64, device="cpu" if gloo else "cuda")
            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            rank_info = [{"rank": r, "device": int(t[0].item()), "images": [int(t[1].item()), int(t[2].item())],
                          "pdhg_event_ms_per_step": float(t[3].item())} for r, t in enumerate(allr)]
        # the replica figure (every rank its own full batch, summed): reported beside `value`, never as it
        weak = None
        if world > 1 and args.scaling == "strong" and not args.no_extras:
            ubw, fw, _ = load_batch(args.data, args.images, N, M, 20211004 + rank)
            sw = make_solver(ubw, fw, args.images)
            wsteps = max(1, min(args.steps, 5))
            Tw, _, _, _ = timed(make_step(sw, torch.zeros(npar, dtype=torch.float64, device="cuda")), wsteps, 1)
            weak = {"weak_value": world * wsteps * args.iters / Tw, "weak_ms_per_step": 1e3 * Tw / wsteps, "weak_steps": wsteps}
            sw.close()

    if rank == 0:
        if solver is None:
            raise SystemExit("bench.py: rank 0 holds no image (images < 1?)")
        st = solver.stats()
        batches = world_out if scaling == "weak" else 1
        value = batches * args.steps * args.iters / T
        launch_us = 1e3 * ev_ms / max(ev_launches, 1)
        # isolated duration of one launch (chains replayed one after the other), the number a
        # rocprofv3 --kernel-trace of this command reports per kernel
        ser_ms, ser_l = 0.0, 0
        if not args.evaluate and not args.multi_handle:
            for _ in range(3):
                solver.denoise(alpha, fetch=False, serialize_chains=1, **kw)
                s2 = solver.stats(); ser_ms += s2["pdhg_ms"]; ser_l += s2["launches"]
        kernel_us = 1e3 * ser_ms / max(ser_l, 1) if ser_l else None
        # the roofline is that of rank 0's kernel: its launch processes rank 0's images (the largest shard)
        O_roof = (shard_range(args.images, st["shards"], 0)[1] if args.multi_handle else O_local)
        if args.multi_handle:
            st_roof = dict(st); st_roof["tiles"] = st["tiles"] * O_roof // max(args.images, 1)
        else:
            st_roof = st
        out = {
            "metric": "PDHG iters/sec (batched 128x128 images)",
            "value": value,
            "unit": "PDHG iterations/s of %s %dx%dx%d %s batch%s" % (
                "ONE" if scaling == "strong" else "a", args.images, N, M,
                "f32 (opt-in, narrower than the reference)" if args.f32 else "f64",
                " sharded over the GPUs" if scaling == "strong" and world_out > 1 else
                (" per GPU, summed over GPUs" if scaling == "weak" and world_out > 1 else "")),
            "n_gpus": world_out, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * T / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if args.f32 else "f64", "data": data_label,
            "config": {"workload": "%dx%dx%d %s batch, %s alpha, %d PDHG iterations per step (%s)" % (
                           args.images, N, M, "f32" if args.f32 else "f64", "per-pixel" if args.alpha_map else "scalar", args.iters,
                           "evaluate: loss + adjoint gradient + all-reduce" if args.evaluate else "denoise"),
                       "images_per_gpu": O_roof, "tile_iters": st["tile_iters"], "tiles_per_launch": st_roof["tiles"],
                       "launches_per_step": st["launches"], "hipgraph": bool(st["graph_used"]),
                       "parallelism": ("images of one batch sharded, dp%d" if scaling == "strong" else "one batch per GPU, dp%d") % world_out},
            "roofline": roofline_of(args, st_roof, M, N, O_roof, args.iters, launch_us, kernel_us, args.f32),
            "pdhg_event_ms_per_step": ev_ms / args.steps,
            "pdhg_event_ms_median": float(np.median(ev_steps)) if ev_steps else None,   # median of the repeats
        }
        if world_out > 1:
            comm = {"form": "torch.distributed, one process per GPU" if not args.multi_handle else "in-library",
                    "world_size": dist.get_world_size() if world > 1 else 1,
                    "backend": (dist.get_backend() if world > 1 else None)}
            if world > 1 and args.backend == "nccl":
                try:
                    comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
                except Exception:
                    pass
            comm["collective_per_step"] = ("all_reduce of [cost, grad] (%d doubles)" % npar) if args.evaluate else "none (denoise: images are independent)"
            out["comm"] = comm
            out["ranks"] = rank_info
        if multi_info:
            out["multi_handle"] = multi_info
        if weak:
            out.update(weak)
            out["weak_note"] = "replicas: every rank solves its own %d-image batch; sum over ranks; NOT `value`" % args.images
        default_wl = (world_out == 1 and not args.evaluate and not args.multi_handle and not args.f32 and not args.alpha_map
                      and M == 128 and args.images == 10)
        if world_out == 1 and not args.evaluate and not args.multi_handle and M * N * O_local <= 16 * 128 * 128:
            # outside the timed region: one full learning-function evaluation on the same resident batch
            # (PDHG + loss + adjoint gradient), the unit of work of the outer trust-region loop
            part = torch.zeros(npar, dtype=torch.float64, device="cuda")
            tt, pm, am = [], [], []
            for _ in range(3):
                t1 = time.perf_counter()
                solver.evaluate_device(alpha, 0.1, part.data_ptr(), **kw)
                torch.cuda.synchronize()
                tt.append(1e3 * (time.perf_counter() - t1))
                s3 = solver.stats(); pm.append(s3["pdhg_ms"]); am.append(s3["adjoint_ms"])
            out["learning_function"] = {"evaluate_ms": min(tt), "pdhg_ms": min(pm), "adjoint_ms": min(am),
                                        "adjoint_method": s3["adjoint_method"], "adjoint_residual": s3["adjoint_residual"],
                                        "note": "tv_op_learning_function on the same batch, best of 3, not part of `value`"}
        if default_wl and not args.no_extras:
            # extra workloads of the default run, outside the timed region: the reference's
            # default num_samples = 1, and one GPU's share of BASELINE config 5 (8 x 1024^2, per-pixel alpha)
            try:
                out["extra_workloads"] = {
                    "single_image": extra_workload(TVSolver, torch, "single_image", 1, 128, args.iters, 5, False, True, "faces_train_128_10"),
                    "config5_share": extra_workload(TVSolver, torch, "config5_share", 8, 1024, 2000, 3, True, True, "synthetic", eval_iters=400),
                    "outer_loop": outer_loop_extra(),
                }
            except Exception as e:   # never lose the headline line to an extra
                out["extra_workloads"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu_baseline and world_out == 1:  # the CPU baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, f_full, ub_full, alpha, N, M)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
