/*
 * bpltv.h -- C ABI of libbpltv: the MI355X (gfx950) inner TV-denoising solver that drops in behind
 * BPLDenoising's evaluate/solve surface.
 *
 * What it replaces in the reference (dvillacis/BPLDenoising, all paths relative to its root):
 *   src/TVLearningFunctionVec.jl:14-27   tv_op_learning_function(x, data, D) -> (u, cost, grad)
 *   src/TVLearningFunctionVec.jl:45-70   denoise(data, x::Real|AbstractArray, op)
 *   src/BPLDenoising.jl:41-82            TVDenoise(data, parameter)          (maxiter = 10000)
 *   src/TVLearningFunctionVec.jl:72-254  gradient / gradient_reg (adjoint state, per image)
 * and, inside those, the external VariationalImaging.op_denoise_pdps loop they call
 * (src/TVLearningFunctionVec.jl:52,67).  The caller -- bilevel_learn, src/TRBox.jl:36,46,227 --
 * is untouched: it only sees a function (x, data, D) -> (u, cost, grad).
 *
 * Conventions
 *   - Plain C: pointers and sizes only.  Every function returns 0 on success or a BPLTV_E_* code;
 *     bpltv_last_error() gives the message.  Nothing throws, prints, or exits.
 *   - Images are Julia `Array{Float64,3}` of size (M, N, O), column major: element (i, j, k) at
 *     i + M*j + M*N*k.  data[1] = ubar (ground truth), data[2] = f (noisy), src/TVLearningFunctionVec.jl:15-16.
 *   - The parameter x is passed as (alpha, am, an), column major am x an:
 *        1 x 1  scalar alpha;  m x n patch parameter (upsampled piecewise-constant, PatchOp);
 *        M x N  per-pixel map.   grad has the same shape (src/TRBox.jl:37-39,167,237).
 *   - Host pointers are read/written during the call only; the library keeps no host pointer.
 *     One call in flight per handle; calls block until the device work is complete.  Any number of handles may be
 *     alive; the handles of one device share its streams (a second handle runs at the speed of the first), so handles
 *     of the SAME device driven from different host threads at once are serialised on the device, not concurrent.
 *   - bpltv_create drives one GPU.  bpltv_create_multi drives several from ONE host thread (the single
 *     Julia task of src/TRBox.jl:192-273): images block-sharded over the devices, one worker thread and
 *     stream per device inside the library, one RCCL collective over xGMI per evaluation on the
 *     [cost, grad...] vector; every other entry point takes either kind of handle.  A host that prefers
 *     one process per GPU shards the images itself and all-reduces bpltv_evaluate_partial /
 *     bpltv_evaluate_device (bpltv_per_image for sharding-independent sums); see INTEGRATION.md.
 */
#ifndef BPLTV_H
#define BPLTV_H

#ifdef __cplusplus
extern "C" {
#endif

#define BPLTV_VERSION 4

enum {
    BPLTV_OK = 0,
    BPLTV_E_ARG = 1,      /* bad argument (null pointer, size mismatch, unsupported shape)   */
    BPLTV_E_HIP = 2,      /* HIP runtime error (message holds hipGetErrorString)             */
    BPLTV_E_NODATA = 3,   /* evaluate/denoise before set_data                                */
    BPLTV_E_NUMERIC = 4,  /* adjoint factorisation broke down (non-positive pivot)           */
    BPLTV_E_NOMEM = 5,
    BPLTV_E_UNSUPPORTED = 6
};
/* Sizes: PDHG, loss, sweep and the adjoint gradient accept any M x N x O whose images fit in HBM.  The gradient
 * factors its linear system by a nested-dissection (multifrontal) Cholesky (about 0.9 KB of workspace per pixel:
 * 0.94 GB for a 1024 x 1024 image); when the workspace of all O images does not fit it runs in groups of as many
 * images as fit, with bitwise the same result (stats.adjoint_chunks; bpltv_set_option "adjoint_budget_mb" forces a budget).
 * params.reserved[4] selects the cross-check solvers: banded Cholesky (LDS window for M <= 138, HBM-resident band
 * of M*N*(M+1) doubles per image beyond; whole batch at once) or block cyclic reduction (M <= 128).  All workspaces
 * are allocated on first use and released again if an allocation fails. */

typedef struct bpltv_handle bpltv_t;

/* The solver NamedTuple of src/TVLearningFunctionVec.jl:33-43 (+ the learning function's Dt). */
typedef struct bpltv_params {
    double rho;          /* Huber smoothing of the TV term; reference 0                           */
    double tau0;         /* reference 5                                                           */
    double sigma0;       /* reference 0.99/5                                                      */
    int accel;           /* reference true                                                        */
    int maxiter;         /* reference 5000 (TVDenoise: 10000); fixed count, no early stop         */
    double delta_t;      /* reference 1e-6: D > delta_t -> gradient, else gradient_reg            */
    int check_every;     /* > 0: evaluate the duality gap every check_every iterations            */
    double gap_tol;      /* > 0 with check_every > 0: stop once max-image gap <= gap_tol.
                            0 = reference behaviour (always maxiter iterations)                   */
    int tile_iters;      /* PDHG iterations fused per kernel launch (temporal blocking depth);
                            0 = library default for the image size                                */
    int use_graph;       /* 1 (default): replay the launch sequence from a hipGraph               */
    double kappa_cap;    /* cap on the active-set weight 1/eps() of the adjoint system; 0 = 1e14  */
    int refine;          /* iterative-refinement sweeps of the adjoint solve; < 0 = default: 2 / 1 / 0 for the
                            scalar gradient / patch and pixel-map parameters / gradient_reg with nested
                            dissection and the HBM band, 3 / 2 / 2 with block cyclic reduction and the
                            LDS band                                                              */
    int deterministic;   /* multi-GPU handles, scalar / patch parameters: 1 = all-gather the per-image rows
                            [cost_k, grad_k...] and add them in global image order, so that cost and grad are
                            bitwise the same for every number of GPUs (and equal to a single handle's);
                            0 (default) = one all-reduce(sum) of the per-device partial vectors           */
    int reserved[5];     /* tuning / measurement knobs, 0 = default:
                            [0] PDHG kernel variant (1-based index into the variant table of bpltv.hip; sum of
                                regularisers: 1 = 32x32 region / 1 px per thread, 2 = 48x48 / 3 px per thread)
                            [1] number of independent launch chains (image groups replayed concurrently)
                            [2] 1 = replay those chains one after the other (isolated kernel timing)
                            [3] must be 0 (BPLTV_E_ARG otherwise); timing-experiment switches exist only in
                                tools/ builds compiled with -DBPLTV_EXPERIMENTS
                            [4] adjoint factorisation: 0 automatic, 1 banded Cholesky, 2 block cyclic reduction
                                (M <= 128, N >= 2; BPLTV_E_UNSUPPORTED otherwise), 3 nested-dissection
                                (multifrontal) Cholesky                                             */
    /* The three choices of the PDHG recurrence that the reference does not pin (its loop, op_denoise_pdps, lives
     * in the absent package VariationalImaging: src/TVLearningFunctionVec.jl:33-43,52).  0
     * everywhere = the restatement every parity claim refers to.  A user who has VariationalImaging on disk and
     * finds it differs aligns the library with these fields instead of rebuilding it.  TV model, dtype 64. */
    int init;            /* 0: x0 = f (default); 1: x0 = 0                                         */
    int order;           /* 0: primal step first (default); 1: dual step first (y from xbar of the previous
                            iteration, then x, then the over-relaxation)                           */
    double opnorm;       /* operator-norm estimate L dividing tau0 and sigma0; 0 = sqrt(8) (sum of
                            regularisers: sqrt(18)), e.g. 2*sqrt(2)*(1 - 1/n) for a tighter bound   */
} bpltv_params;

typedef struct bpltv_stats {
    int M, N, O, device;
    int iterations;            /* PDHG iterations executed by the last denoise/evaluate          */
    int launches;              /* PDHG kernel launches of that call                               */
    int tile_iters;            /* fused iterations per launch actually used                       */
    int tiles;                 /* workgroups per PDHG launch                                      */
    int region_i, region_j;    /* pixels one workgroup computes per fused iteration (core + halo): with tiles
                                  and tile_iters this gives the redundancy of the temporal blocking         */
    int graph_used;
    double pdhg_ms;            /* HIP-event time of the PDHG launch sequence (device)            */
    double cost_ms;
    double adjoint_ms;         /* HIP-event time of the adjoint gradient (all images)            */
    double total_ms;           /* host wall time of the last call                                 */
    double bytes_per_px_iter;  /* algorithmic bytes: 56 (scalar/patch alpha) or 64 (alpha map)    */
    double algorithmic_bytes;  /* bytes_per_px_iter * M*N*O * iterations                          */
    double last_gap;           /* max over images of the duality gap if it was computed, else -1  */
    double adjoint_residual;   /* max over images of ||D^-1/2 (rhs - A p)|| / ||D^-1/2 rhs||, D = diag(A), after
                                  refinement: the quality gate of the adjoint solve (<= 1e-8 on a correct
                                  solve; above BPLTV_RESIDUAL_GATE the call fails with BPLTV_E_NUMERIC)   */
    double adjoint_residual_raw; /* the same without the diagonal scaling: dominated by the rounding of
                                  the 1e14-weighted active rows, informational only                   */
    double kappa_used;         /* active-set weight of the adjoint system that produced the returned
                                  gradient: 1/eps() capped by kappa_cap (scalar), 1/sqrt(eps()) (patch),
                                  times 1e-2 per retry; 0 for gradient_reg                             */
    int adjoint_attempts;      /* factorisations tried by the last gradient: 1 = no breakdown, 2..3 = the
                                  weight was reduced by 1e-2 per retry after a non-positive pivot      */
    int adjoint_method;        /* 1 banded Cholesky (LDS window), 2 block cyclic reduction,
                                  3 banded Cholesky (HBM band), 4 banded LU (sum of regularisers, row-scaled
                                  gradient_reg system, with reserved[4] = 1), 5 nested-dissection (multifrontal)
                                  Cholesky, 6 nested-dissection LU (that row-scaled system, the default)  */
    int reg_gradient_used;     /* 1 if the last evaluate took the gradient_reg branch             */
    int ngpus;                 /* distinct devices behind this handle (1 for bpltv_create)        */
    int shards;                /* image shards (= worker threads) behind this handle              */
    int collective;            /* last evaluate of a multi handle: 0 none (one shard), 1 ncclAllReduce,
                                  2 ncclAllGather + ordered sum, 3 host sum (repeated devices)     */
    double collective_ms;      /* host wall time of that collective (launch + completion)        */
    int nccl_ranks;            /* ranks of the RCCL communicator behind a multi handle as RCCL itself reports them
                                  (ncclCommCount); 0 = no communicator (single-device handle, repeated devices)  */
    int hb_sync;               /* cross-stream dependencies of the HBM band pipeline used by the last gradient:
                                  0 not used, 1 HIP events, 2 stream memory operations (hipStreamWaitValue32)    */
    int adjoint_chunks;        /* image groups the last adjoint gradient was processed in (1 = whole batch at once;
                                  more when the factor workspace of all images does not fit, option "adjoint_budget_mb") */
    int pdhg_variant;          /* 1-based index of the PDHG kernel the last solve ran (the variant table of
                                  csrc/bpltv.hip: 1..15 pdhg_tile_kernel, 16..18 pdhg_wave_kernel, 19.. pdhg_rows_kernel and its re-cuts;
                                  sum of regularisers: 1 sr_tile_kernel, 2 sr_strip_kernel)                      */
    int ncu;                   /* compute units of the device (hipDeviceProp_t.multiProcessorCount): what bench.py prices
                                  the VALU issue floor against                                                   */
    int launch_chains;         /* independent launch chains (image groups replayed concurrently) of the last solve */
    int sweep_shards;          /* last bpltv_sweep of a multi handle: devices the K parameter blocks were split over
                                  (replica mode), 0 = the images were split / single device                        */
    int reserved_i;
    double launch_host_ms[2];  /* host time the last solve's hipGraphLaunch calls took: chain 0 (calling thread) and
                                  chain 1 (launcher thread); 0 when the solve had one chain or ran without graphs    */
} bpltv_stats_t;

#define BPLTV_RESIDUAL_GATE 1e-6

/* Fill *p with the reference defaults (src/TVLearningFunctionVec.jl:33-43, delta_t 1e-6). */
int bpltv_default_params(bpltv_params *p);

/* Create a solver for O images of size M x N on HIP device `device` (-1 = current device).
 * dtype: 64 = Float64, the reference's arithmetic (src/TVLearningFunctionVec.jl:8-9) and what every parity claim
 * refers to.  32 = opt-in: the PDHG iteration of the TV model (bpltv_denoise, bpltv_evaluate, bpltv_sweep) runs in
 * single precision -- f, the parameter and the step table rounded to float, state in float, half the bytes per
 * pixel-iteration -- and its result is widened to double; loss, duality gap, adjoint gradient, the sum-of-regularisers
 * model and every array crossing this boundary stay Float64.  Narrower than the reference: u differs from the
 * Float64 result by up to ~2e-5 absolute after 5000 iterations, and the gradient by ~1 %: the reference's active set
 * |grad u| < 1e-12 (src/TVLearningFunctionVec.jl:110) does not survive float noise in u (tests/test_gpu_f32.py).  Other values: BPLTV_E_ARG. */
int bpltv_create(bpltv_t **h, int M, int N, int O, int device, int dtype);
/* The same over `ngpus` devices (0 = all visible; devices 0..ngpus-1) driven from one host thread -- the
 * form SURVEY section 8(b)/(e) specifies for the single Julia task of src/TRBox.jl:192-273.  Images
 * [lo_k, hi_k) = block distribution of O over min(ngpus, O) shards (the first O % shards get one more);
 * communicator from ncclCommInitAll; per evaluation ONE RCCL collective on [cost, grad...] (1 + am*an
 * doubles): ncclAllReduce(sum, f64), or ncclAllGather of the per-image rows when params.deterministic.
 * set_data / denoise / evaluate / gradient / sweep / per_image / duality_gap take and return whole-batch
 * host arrays exactly as with bpltv_create (each device copies its slice); the device-pointer entry points
 * (set_data_device, evaluate_device, u_device, copy_u_device) return BPLTV_E_UNSUPPORTED on more than one shard.
 * Status: verified with ngpus = 1 (a one-rank communicator) and with several shards on one device (host sum); the
 * collectives over ngpus > 1 have not yet run on hardware -- tests/test_gpu_multi.py holds the checks that switch on
 * when two or more devices are visible.  stats.nccl_ranks reports what ncclCommCount says.
 * Devices beyond min(ngpus, O) hold no image shard (one image cannot be split), but bpltv_sweep uses them: see there. */
int bpltv_create_multi(bpltv_t **h, int M, int N, int O, int ngpus, int dtype);
/* Explicit placement: shard k of `nshards` runs on HIP device devices[k].  A device may appear more than
 * once (rehearsal of the sharded path on one GPU); RCCL cannot put two ranks on one device, so the collective
 * is then replaced by the same sum / ordered sum on the host. */
int bpltv_create_sharded(bpltv_t **h, int M, int N, int O, const int *devices, int nshards, int dtype);
int bpltv_destroy(bpltv_t *h);

/* Upload the dataset (ubar, f) once; it is identical for every evaluation of a run
 * (src/TRBox.jl:210,227 pass the same `ds`).  Host pointers. */
int bpltv_set_data(bpltv_t *h, const double *ubar, const double *f);
/* Same, from buffers already resident in HBM (device pointers, copied device-to-device). */
int bpltv_set_data_device(bpltv_t *h, const double *d_ubar, const double *d_f);

/* denoise(data, x, op; kwargs...): src/TVLearningFunctionVec.jl:45-70, src/BPLDenoising.jl:41-82.
 * u_out: host, M*N*O doubles, or NULL to leave the result on the device (bpltv_u_device). */
int bpltv_denoise(bpltv_t *h, const double *alpha, int am, int an, const bpltv_params *p,
                  double *u_out);

/* The same solve with the parameter already resident in HBM (d_alpha: device pointer, am*an doubles, column major) and
 * the result left there (bpltv_u_device): no host array crosses the boundary, which is how bench.py times a pixel-map
 * parameter (8 MiB for 1024 x 1024) without a PCIe copy in the timed region.  The entries are checked on the device
 * (finite, >= 0) exactly as bpltv_denoise checks a host array.  Single-device handles (multi: BPLTV_E_UNSUPPORTED beyond
 * one shard). */
int bpltv_denoise_device(bpltv_t *h, const double *d_alpha, int am, int an, const bpltv_params *p);

/* tv_op_learning_function(x, data, D): src/TVLearningFunctionVec.jl:14-27.
 * cost_out: 1 double; grad_out: am*an doubles; u_out: host M*N*O doubles or NULL. */
int bpltv_evaluate(bpltv_t *h, const double *alpha, int am, int an, double delta,
                   const bpltv_params *p, double *u_out, double *cost_out, double *grad_out);

/* Sum-of-regularisers model: min_u 0.5||u - f||^2 + a1 ||G_fwd u|| + a2 ||G_bwd u|| + a3 ||G_ctr u|| (isotropic
 * 2,1 norms; forward, backward and centred differences), src/SumRegsLearningFunction.jl.
 *   bpltv_sumregs_evaluate = sumregs_learning_function(x, data, D; Dt = 1e-3) -> (u, cost, grad)   (:8-36)
 *   bpltv_sumregs_denoise  = sumregs_denoise(data, x, op1, op2, op3[, pOp])                       (:38-85)
 * alpha: 3 * am * an doubles, the three parameter slices x[:, :, k] (column major am x an) one after the other;
 * am = an = 1 is the Vector x = [a1; a2; a3] (:8), m x n x 3 the patch parameter (:22).  grad_out has the same
 * layout.  p = NULL: bpltv_sumregs_default_params (delta_t = 1e-3).  D > delta_t: sumregs_gradient (:264-407),
 * else sumregs_gradient_reg (:112-262; with a patch parameter its row-scaled system is not symmetric and is
 * factored by a banded LU).  The adjoint system is a 13-point stencil, factored by nested dissection (separators two
 * pixels wide; params.reserved[4] = 1: the HBM band solver at bandwidth 2M).
 * Both take single- and multi-device handles; set_data, per_image, u_device, duality_gap, stats are shared with the TV
 * model. */
int bpltv_sumregs_default_params(bpltv_params *p);
int bpltv_sumregs_denoise(bpltv_t *h, const double *alpha, int am, int an, const bpltv_params *p, double *u_out);
int bpltv_sumregs_evaluate(bpltv_t *h, const double *alpha, int am, int an, double delta, const bpltv_params *p,
                           double *u_out, double *cost_out, double *grad_out);

/* Sharded form: this handle's images only.  partial_out (host, 1 + am*an doubles) receives
 * [cost, grad...] summed over the handle's O images; the caller all-reduces it across shards
 * (cost and grad are plain sums over images: src/TVLearningFunctionVec.jl:20,80,172). */
int bpltv_evaluate_partial(bpltv_t *h, const double *alpha, int am, int an, double delta,
                           const bpltv_params *p, double *u_out, double *partial_out);
/* Same with the partial vector written to device memory (d_partial: 1 + am*an doubles in HBM),
 * ready for an RCCL all-reduce without a host round trip. */
int bpltv_evaluate_device(bpltv_t *h, const double *alpha, int am, int an, double delta,
                          const bpltv_params *p, double *d_partial);

/* Per-image pieces of the last evaluate (scalar or patch parameter; BPLTV_E_UNSUPPORTED for a pixel map):
 * out (host, O*(1 + am*an) doubles), image k at out + k*(1 + am*an): [cost_k, grad_k...].  The totals of
 * evaluate are these rows added in image order, so a host that gathers the rows of all shards and adds
 * them in global image order gets results that do not depend on how the images were sharded. */
int bpltv_per_image(bpltv_t *h, double *out);

/* Device pointer of the last primal result u (M*N*O doubles, valid until the next call). */
int bpltv_u_device(bpltv_t *h, const double **d_u);
/* Copy the last primal result to a device buffer owned by the caller. */
int bpltv_copy_u_device(bpltv_t *h, double *d_dst);

/* Duality gap of the last solve per image (host, O doubles): gap_k >= 0.5*||u_k - u*_k||^2.  TV model and
 * sum-of-regularisers model (the gap of whichever was solved last). */
int bpltv_duality_gap(bpltv_t *h, double *gap_out);

/* FwdGradientOp and its adjoint on the device (src/TVLearningFunctionVec.jl:17; matrix form
 * :106).  Host pointers, one M x N image; d1/d2 are the two stacked components. */
int bpltv_grad_fwd(bpltv_t *h, const double *x, double *d1, double *d2);
int bpltv_grad_fwd_adjoint(bpltv_t *h, const double *y1, const double *y2, double *out);

/* Adjoint gradient alone for given (u, ubar) held by the caller (host, M*N*O each):
 * gradient (reg = 0, src/TVLearningFunctionVec.jl:98-135,219-254) or gradient_reg (reg = 1,
 * :137-161,192-215), summed over the O images as the batch wrappers do (:72-96,163-190). */
int bpltv_gradient(bpltv_t *h, const double *u, const double *ubar, const double *alpha, int am,
                   int an, int reg, const bpltv_params *p, double *grad_out);

/* Forward-only parameter sweep: generate_cost / generate_2d_cost (src/BPLDenoising.jl:92-111,
 * :136-158) evaluate cost(alpha_k) = 0.5*||TVDenoise(f, alpha_k) - ubar||^2 for a range of parameters,
 * one solve after the other.  Here the K parameter blocks (each am x an, column major, K*am*an
 * doubles) times the O resident images form ONE batch of K*O independent ROF problems -- the second
 * data-parallel axis that fills a GPU even with a single image.  cost_out: K doubles; u_out: NULL
 * or K*M*N*O doubles (parameter-major).  Use maxiter = 10000 for the TVDenoise setting.
 * Multi-device handles split whichever axis leaves the smaller largest share per device: the images (device k solves
 * K x O_k problems on the shard it already holds) or the K parameter blocks (device r solves K_r x O problems on a
 * REPLICA -- a second, whole copy of the dataset made on every requested device at the first such sweep, filled from
 * the shards' resident data).  With the reference's default num_samples = 1 (src/BPLDenoising.jl:313) or the one-pair
 * sets (datasets/cameraman_128_10/filelist.txt) only the parameter axis can use more than one GPU: 100 parameters x 1
 * image on 8 devices = 13,13,13,13,12,12,12,12 problems per device.  Costs are concatenated (every replica sums over
 * all O images in image order), u_out slices are written in place: the results are bitwise those of one single-device
 * handle either way.  bpltv_set_option "sweep_split" (0 automatic, 1 images, 2 parameters) forces an axis;
 * stats.sweep_shards = devices the parameter blocks were split over (0: image split / single device). */
int bpltv_sweep(bpltv_t *h, const double *alphas, int K, int am, int an, const bpltv_params *p,
                double *cost_out, double *u_out);

/* Handle options: aids for tests and measurements, none of them changes a result or is needed for the reference's
 * behaviour (the reference has no counterpart; its sparse `\` at src/TVLearningFunctionVec.jl:131,248 has no knobs).
 * They replace the environment variables earlier versions read on the product path.  Unknown names: BPLTV_E_ARG.
 *   "adjoint_budget_mb"  > 0: HBM (MB) the adjoint's factor workspace may take -- forces the gradient to run in image
 *                        groups (stats.adjoint_chunks; bitwise the same result); 0 = what is free minus a 2 GB reserve
 *   "sr_force_lu"        1: sum of regularisers -- factor the symmetric systems by the LU variant of the nested
 *                        dissection as well (cross-check of that variant)
 *   "nd_leaf"            leaf size in pixels of the nested-dissection tree (0 = 32)
 *   "nd_wave"            0: fronts of <= 64 rows are factored by the workgroup-per-front kernel like the larger small fronts
 *                        (cross-check of the wave-per-front kernel, which is the default: 1)
 *   "nd_skinny"          0: fronts of <= 32 pivots that keep only their pivot block columns in LDS go through the older
 *                        kernels instead (cross-check; default 1)
 *   "nd_skinny_min", "nd_skinny2_min"   (front, image) pairs a level needs before those two kernels take it (defaults 0 / 256:
 *                        below that the five short launches of the large regime finish a level sooner than one long one)
 *   "nd_staged"          0: substitutions of the small levels by the column-loop kernels (cross-check: the same bits; default 1)
 *   "hb_sync"            HBM band cross-check solver (params.reserved[4] = 1): 0 automatic, 1 HIP events (what a
 *                        rocprofv3 run needs), 2 stream memory operations (BPLTV_E_HIP when the device has none)
 *   "hb_single_stream"   1: that solver's three streams folded into one (rocprofv3 --pmc)
 *   "hb_rw"              32 | 128: rows per workgroup of its substitutions (0 = by size)
 *   "sweep_split"        multi-device handles, bpltv_sweep: 0 automatic, 1 split the images, 2 split the parameter blocks
 * Multi-device handles pass the other options to every shard (and sweep replica). */
int bpltv_set_option(bpltv_t *h, const char *name, double value);

int bpltv_stats(bpltv_t *h, bpltv_stats_t *out);
const char *bpltv_last_error(bpltv_t *h);
int bpltv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BPLTV_H */
