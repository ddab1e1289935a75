"""Experiment drivers and result artefacts -- host code around the
learning function, mirroring the reference's entry points:

  scalar_bilevel_tv_learn(dataset_name=..., num_samples=...)   /root/reference/src/BPLDenoising.jl:325-343
  patch_bilevel_tv_learn(dataset_name=..., num_samples=...)    :360-377
  save_results(params, b, b_data, x, opt_img, log)             :182-297
  scalar_bilevel_sumregs_learn / patch_bilevel_sumregs_learn   :422-489   (sum-of-regularisers model)
  validate_sumregs_parameter                                   :506-539

Written files (same names and columns as the reference):
  <prefix>.txt           performance log: iter, time, function_value, gradient_value, radius_value,
                         stopping_criteria (BilevelLogEntry, src/BilevelVisualise.jl:39-46), preceded by
                         the comment line `# params = ..., x = ...` (src/BPLDenoising.jl:194)
  <prefix>_quality.txt   img_num, orig_ssim, orig_psnr, out_ssim, out_psnr per image + the mean line (:195-214)
  <prefix>_{true,data,reco}_<i>.png, <prefix>_par.png (array parameters, :221-256)

PARITY UNPINNED: `write_log` (AlgTools) and `assess_ssim` / `assess_psnr` (ImageQualityIndexes.jl) are
external and absent; the log layout is tab separated with a header row, PSNR is 10 log10(1/MSE) for
images in [0, 1] and SSIM is the standard Wang et al. index (11x11 Gaussian window, sigma 1.5, K = (0.01,
0.03), mean over the map).  The reference's quirk `mean_psnr += mean_psnr` (:282, the save_results method for
m x n x 3 parameters: the mean PSNR it writes is always 0) is reproduced for that method only.
"""
import os
import time

import numpy as np

from . import trbox
from .datasets import testdataset

# /root/reference/src/BPLDenoising.jl:306-323, :350-357
default_params = dict(verbose_iter=1, maxiter=20, save_results=True, dataset_name="cameraman_128_5",
                      save_iterations=False, tol=1e-5, num_samples=1)
bilevel_params = dict(eta1=0.25, eta2=0.75, beta1=0.25, beta2=1.9, delta0=0.1, alpha0=0.1)
patch_bilevel_params = dict(eta1=0.25, eta2=0.75, beta1=0.25, beta2=1.9, delta0=1e-4, alpha0=1e-4 * np.ones((2, 2)))
sumregs_bilevel_params = dict(eta1=0.25, eta2=0.75, beta1=0.25, beta2=1.9, delta0=0.01,
                              alpha0=np.array([0.001, 0.001, 0.001]))                      # :422-430
patch_sumregs_bilevel_params = dict(eta1=0.25, eta2=0.75, beta1=0.25, beta2=1.5, delta0=0.1,
                                    alpha0=0.001 * np.ones((3, 2, 2)))                     # :455-462, Julia ones(2,2,3)
default_save_prefix = "results"

_ALIASES = {"η₁": "eta1", "η₂": "eta2", "β₁": "beta1", "β₂": "beta2", "Δ₀": "delta0", "α₀": "alpha0"}


def linear_stretch(a):
    """adjust_histogram!(a, LinearStretching()): affine map of [min, max] onto [0, 1]."""
    a = np.asarray(a, dtype=np.float64)
    lo, hi = float(a.min()), float(a.max())
    return (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)


def assess_psnr(ref, img, peak=1.0):
    mse = float(np.mean((np.asarray(ref, dtype=np.float64) - np.asarray(img, dtype=np.float64)) ** 2))
    return float("inf") if mse == 0 else float(10.0 * np.log10(peak * peak / mse))


def assess_ssim(ref, img, sigma=1.5, K=(0.01, 0.03), peak=1.0):
    from scipy.ndimage import gaussian_filter
    x = np.asarray(ref, dtype=np.float64); y = np.asarray(img, dtype=np.float64)
    g = lambda a: gaussian_filter(a, sigma, truncate=5.0 / sigma * 1.0, mode="nearest")   # 11 x 11 support
    mx, my = g(x), g(y)
    sxx, syy, sxy = g(x * x) - mx * mx, g(y * y) - my * my, g(x * y) - mx * my
    c1, c2 = (K[0] * peak) ** 2, (K[1] * peak) ** 2
    s = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
    return float(np.mean(s))


def write_log(path, log, comment=""):
    cols = ["iter", "time", "function_value", "gradient_value", "radius_value", "stopping_criteria"]
    with open(path, "w") as fh:
        if comment:
            fh.write(comment if comment.endswith("\n") else comment + "\n")
        fh.write("\t".join(cols) + "\n")
        for e in log:
            fh.write("\t".join(repr(float(e[c])) if c != "iter" else str(e[c]) for c in cols) + "\n")


def _save_png(path, a):
    from PIL import Image
    img = np.round(255.0 * np.clip(np.asarray(a, dtype=np.float64), 0.0, 1.0)).astype(np.uint8)
    Image.fromarray(img.T).save(path)          # (N, M) batch layout -> PIL rows = Julia rows


def patch_upsample(x, M, N):
    """PatchOp: piecewise-constant upsampling of the m x n parameter to the image size (as the kernels do)."""
    x = np.asarray(x, dtype=np.float64)
    an, am = x.shape                            # python (n, m) == Julia (m, n)
    jj = (np.arange(N) * an) // N
    ii = (np.arange(M) * am) // M
    return x[np.ix_(jj, ii)]


def save_results(params, b, b_data, x, opt_img, log, out_root=None):
    """b, b_data, opt_img: (O, N, M) batches.  Returns the dict of written paths."""
    if not params.get("save_results", True):
        return {}
    out_path = os.path.join(out_root or default_save_prefix, params["dataset_name"])
    os.makedirs(out_path, exist_ok=True)
    prefix = os.path.join(out_path, params["save_prefix"])
    written = {"perf": prefix + ".txt", "quality": prefix + "_quality.txt", "png": []}
    shown = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in params.items()}
    write_log(written["perf"], log, "# params = %s, x = %s" % (shown, np.asarray(x).tolist()))
    O = b.shape[0]
    with open(written["quality"], "w") as io:
        io.write("img_num \t orig_ssim \t orig_psnr \t out_ssim \t out_psnr\n")
        mean_ssim = mean_psnr = 0.0
        for i in range(O):
            ns, npsnr = assess_ssim(b[i], b_data[i]), assess_psnr(b[i], b_data[i])
            os_, opsnr = assess_ssim(b[i], opt_img[i]), assess_psnr(b[i], opt_img[i])
            io.write("%d\t %r \t %r \t %r \t %r\n" % (i + 1, float(ns), float(npsnr), float(os_), float(opsnr)))
            mean_ssim += os_
            mean_psnr += mean_psnr if np.ndim(x) == 3 else opsnr     # :282 (`mean_psnr += mean_psnr`) vs :203, :243
            for tag, arr in (("true", b[i]), ("data", b_data[i]), ("reco", opt_img[i])):
                p = "%s_%s_%d.png" % (prefix, tag, i + 1)
                _save_png(p, arr); written["png"].append(p)
        io.write("\t\t\t\t\t %r\t %r\n" % (float(mean_ssim / O), float(mean_psnr / O)))
    if np.ndim(x) == 2:
        p = prefix + "_par.png"
        _save_png(p, linear_stretch(patch_upsample(x, b.shape[2], b.shape[1])))
        written["png"].append(p)
    elif np.ndim(x) == 3:                          # :290-296: one histogram stretch over the three upsampled slices
        xb = linear_stretch(np.stack([patch_upsample(x[k], b.shape[2], b.shape[1]) for k in range(3)]))
        for k in range(3):
            p = "%s_par_%d.png" % (prefix, k + 1)
            _save_png(p, xb[k]); written["png"].append(p)
    return written


def _resolve(defaults, kwargs):
    p = dict(default_params); p.update(defaults)
    for k, v in kwargs.items():
        p[_ALIASES.get(k, k)] = v
    return p


def _run(params, learning_function, datasets_root, npz, out_root, lf_kwargs):
    b, b_noisy = testdataset(params["dataset_name"], root=datasets_root, npz=npz)
    n = params["num_samples"]
    b, b_noisy = np.ascontiguousarray(b[:n]), np.ascontiguousarray(b_noisy[:n])
    log, t0 = [], time.perf_counter()

    def iterate(e):
        e = dict(e); e["time"] = time.perf_counter() - t0
        log.append(e)
        if params["verbose_iter"] and e["iter"] % params["verbose_iter"] == 0:
            print("%d/%d J=%g |g|=%g radius=%g res=%g" % (e["iter"], params["maxiter"], e["function_value"],
                                                        e["gradient_value"], e["radius_value"], e["stopping_criteria"]))
    inner = dict(lf_kwargs or {})    # solver kwargs (e.g. the inner maxiter) must not meet the outer loop's own

    def lf(x, ds, delta):
        return learning_function(x, ds, delta, **inner)
    x, u, _ = trbox.bilevel_learn((b, b_noisy), lf, params["alpha0"], params["delta0"],
                                  maxiter=params["maxiter"], tol=params["tol"], eta1=params["eta1"], eta2=params["eta2"],
                                  beta1=params["beta1"], beta2=params["beta2"], log=iterate)
    return b, b_noisy, x, np.asarray(u), log


def scalar_bilevel_tv_learn(learning_function=None, datasets_root=None, npz=None, out_root=None, lf_kwargs=None, **kwargs):
    """x, u, log, written = scalar_bilevel_tv_learn(dataset_name="cameraman_128_5", num_samples=1, ...)"""
    if learning_function is None:
        from .learning_function import tv_op_learning_function as learning_function
    params = _resolve(bilevel_params, kwargs)
    params["save_prefix"] = "tv_optimal_parameter_scalar_" + params["dataset_name"]
    b, b_noisy, x, u, log = _run(params, learning_function, datasets_root, npz, out_root, lf_kwargs)
    u, b, b_noisy = linear_stretch(u), linear_stretch(b), linear_stretch(b_noisy)      # :337-339
    return x, u, log, save_results(params, b, b_noisy, x, u, log, out_root)


def patch_bilevel_tv_learn(learning_function=None, datasets_root=None, npz=None, out_root=None, lf_kwargs=None, **kwargs):
    if learning_function is None:
        from .learning_function import tv_op_learning_function as learning_function
    params = _resolve(patch_bilevel_params, kwargs)
    params["save_prefix"] = "tv_optimal_parameter_%s_%s" % (tuple(np.shape(params["alpha0"])[::-1]), params["dataset_name"])
    b, b_noisy, x, u, log = _run(params, learning_function, datasets_root, npz, out_root, lf_kwargs)
    u = linear_stretch(u)                                                               # :371
    return x, u, log, save_results(params, b, b_noisy, x, u, log, out_root)


def scalar_bilevel_sumregs_learn(learning_function=None, datasets_root=None, npz=None, out_root=None, lf_kwargs=None, **kwargs):
    """/root/reference/src/BPLDenoising.jl:432-449: x in R^3 (forward, backward, centred TV weights)."""
    if learning_function is None:
        from .learning_function import sumregs_learning_function as learning_function
    params = _resolve(sumregs_bilevel_params, kwargs)
    params["save_prefix"] = "sumregs_optimal_parameter_scalar_" + params["dataset_name"]
    b, b_noisy, x, u, log = _run(params, learning_function, datasets_root, npz, out_root, lf_kwargs)
    u = linear_stretch(u)                                                               # :444
    return x, u, log, save_results(params, b, b_noisy, x, u, log, out_root)


def patch_bilevel_sumregs_learn(learning_function=None, datasets_root=None, npz=None, out_root=None, lf_kwargs=None, **kwargs):
    """/root/reference/src/BPLDenoising.jl:464-481: x of shape m x n x 3 (numpy (3, n, m))."""
    if learning_function is None:
        from .learning_function import sumregs_learning_function as learning_function
    params = _resolve(patch_sumregs_bilevel_params, kwargs)
    shp = np.shape(params["alpha0"])
    params["save_prefix"] = "sumregs_optimal_parameter_patch_%s%s" % ((shp[2], shp[1], shp[0]), params["dataset_name"])
    b, b_noisy, x, u, log = _run(params, learning_function, datasets_root, npz, out_root, lf_kwargs)
    u = linear_stretch(u)                                                               # :476
    return x, u, log, save_results(params, b, b_noisy, x, u, log, out_root)


def validate_sumregs_parameter(parameter, dataset_name="cameraman_128_5", datasets_root=None, npz=None, out_root=None,
                               learning_function=None, **solver_kwargs):
    """/root/reference/src/BPLDenoising.jl:506-539 as written there calls `sumregs_learning_function(parameter,
    noisy, 0.1)` with the noisy batch in the place of the (true, noisy) tuple -- which cannot index `data[2]` as
    intended; the evident intent (denoise the validation set with the learned parameter, report cost and quality)
    is what is implemented.  Returns (u, cost, written)."""
    if learning_function is None:
        from .learning_function import sumregs_learning_function as learning_function
    full = _full(dataset_name)
    img, noisy = _dataset(dataset_name, None, datasets_root, npz)
    u, cost, _ = learning_function(np.asarray(parameter, dtype=np.float64), (img, noisy), 0.1, **solver_kwargs)
    u = np.asarray(u)
    params = dict(save_results=True, dataset_name=full,
                  save_prefix="val_sumregs_optimal_parameter_scalar_%s_%s" % (tuple(np.shape(parameter)[::-1]), full))
    written = save_results(params, img, noisy, 0.0, u, [], out_root)
    os.remove(written.pop("perf"))
    return u, cost, written


# ---- forward-only sweeps and validation (src/BPLDenoising.jl:92-178, 381-415) ----------
def _dataset(dataset_name, num_samples, datasets_root, npz):
    true_, data = testdataset(dataset_name, root=datasets_root, npz=npz)
    if num_samples is not None:
        true_, data = true_[:num_samples], data[:num_samples]
    return np.ascontiguousarray(true_), np.ascontiguousarray(data)


def _use(ngpus, devices):
    """ngpus / devices of the sweep and validation drivers: the in-library multi-device handle behind the reference-named
    entry points (learning_function.use_devices).  With num_samples = 1 -- the reference's default, src/BPLDenoising.jl:313
    -- the K parameters of a sweep are what is split over the devices (bpltv_sweep, replica mode)."""
    if ngpus is not None or devices is not None:
        from .learning_function import use_devices
        use_devices(ngpus=ngpus, devices=devices)


def generate_scalar_tv_cost(dataset_name, parameter_range, num_samples=1, datasets_root=None, npz=None, out_root=None,
                            ngpus=None, devices=None, **solver_kwargs):
    """costs[i] = L2CostFunction(TVDenoise(data, parameter_range[i]), true): all parameters as ONE batch of
    K*O problems on the GPU (bpltv_sweep; over `ngpus` devices the parameters or the images are split, whichever
    leaves the smaller share per device).  Saved as <dataset>_cost.npz (the reference writes JLD2)."""
    from .learning_function import generate_cost
    _use(ngpus, devices)
    full = _full(dataset_name)
    true_, data = _dataset(dataset_name, num_samples, datasets_root, npz)
    parameter_range = np.asarray(parameter_range, dtype=np.float64)
    costs = np.asarray(generate_cost((true_, data), parameter_range, **solver_kwargs))
    out = os.path.join(out_root or default_save_prefix, full)
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, full + "_cost.npz"), parameter_range=parameter_range, costs=costs)
    return costs


def generate_2d_tv_cost(dataset_name, parameter_range_1, parameter_range_2, num_samples=1, datasets_root=None, npz=None,
                        out_root=None, ngpus=None, devices=None, **solver_kwargs):
    """costs[i, j] for the 2 x 1 patch parameter [p1[i]; p2[j]] (src/BPLDenoising.jl:136-158)."""
    from .learning_function import generate_cost
    _use(ngpus, devices)
    full = _full(dataset_name)
    true_, data = _dataset(dataset_name, num_samples, datasets_root, npz)
    p1 = np.asarray(parameter_range_1, dtype=np.float64); p2 = np.asarray(parameter_range_2, dtype=np.float64)
    # Julia [a; b] .* ones(2,1) is a 2 x 1 matrix (two patches along dim 1) == python shape (1, 2)
    alphas = np.array([[[a, b]] for a in p1 for b in p2])
    costs = np.asarray(generate_cost((true_, data), alphas, **solver_kwargs)).reshape(p1.size, p2.size)
    out = os.path.join(out_root or default_save_prefix, full)
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, full + "_cost_2d.npz"), parameter_range_1=p1, parameter_range_2=p2, costs=costs)
    return costs


def validate_tv_parameter(parameter, dataset_name="cameraman_128_5", datasets_root=None, npz=None, out_root=None,
                          denoise_function=None, ngpus=None, devices=None, **solver_kwargs):
    """TVDenoise of the whole validation set with a learned parameter; cost + quality table + PNGs
    (src/BPLDenoising.jl:381-415).  Over `ngpus` devices the images of the set are sharded (forward solves only: no
    collective)."""
    _use(ngpus, devices)
    if denoise_function is None:
        from .learning_function import TVDenoise as denoise_function
    full = _full(dataset_name)
    img, noisy = _dataset(dataset_name, None, datasets_root, npz)
    u = np.asarray(denoise_function(noisy, parameter, **solver_kwargs))
    cost = 0.5 * float(np.sum((u - img) ** 2))
    params = dict(save_results=True, dataset_name=full,
                  save_prefix="val_tv_optimal_parameter_scalar_%s_%s" % (tuple(np.shape(parameter)[::-1]), full))
    written = save_results(params, img, noisy, np.asarray(parameter).mean(), u, [], out_root)
    os.remove(written.pop("perf"))               # the reference writes only the quality file and the PNGs here
    return u, cost, written


def _full(name):
    from .datasets import full_datasetname
    return full_datasetname(name)
