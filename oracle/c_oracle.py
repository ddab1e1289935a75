"""ctypes binding of oracle/libbpltv_oracle.so  --  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
Arrays follow the C ABI convention: numpy (O, N, M) C-contiguous == Julia (M, N, O) column major.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbpltv_oracle.so")
_dp = C.POINTER(C.c_double)
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("bpltv_oracle.c", "sumregs_oracle.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libbpltv_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.bplo_step_table.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, _dp]
        L.bplo_step_table.restype = None
        L.bplo_pdhg.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double,
                                C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int]
        L.bplo_pdhg.restype = C.c_int
        L.bplo_pdhg_f32.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_double, C.c_int, C.c_int, _dp]
        L.bplo_pdhg_f32.restype = C.c_int
        L.bplo_pdhg_rows.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double,
                                     C.c_double, C.c_double, C.c_int, C.c_int, _dp, C.c_int, C.c_int]
        L.bplo_pdhg_rows.restype = C.c_int
        L.bplo_pdhg_variant.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, C.c_int]
        L.bplo_pdhg_variant.restype = C.c_int
        L.bplo_cost.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.bplo_cost.restype = C.c_double
        L.bplo_gap.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp]
        L.bplo_gap.restype = None
        L.bplo_grad_fwd.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.bplo_grad_fwd_T.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.bplo_patch_upsample.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.bplo_patch_adjoint.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.bplo_gradient_image.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int,
                                          C.c_double, C.c_int, _dp, _dp, _dp]
        L.bplo_gradient_image.restype = C.c_int
        L.bplo_gradient.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_int, _dp, _dp]
        L.bplo_gradient.restype = C.c_int
        L.bplo_max_threads.restype = C.c_int
        # sum of regularisers (sumregs_oracle.c)
        L.bplo_sr_grad.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.bplo_sr_gradT.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.bplo_sumregs_pdhg.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_double, C.c_int, C.c_int, _dp, _dp, C.c_int]
        L.bplo_sumregs_pdhg.restype = C.c_int
        L.bplo_sumregs_gap.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp]
        L.bplo_sumregs_gradient_image.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                                                  _dp, _dp, _dp]
        L.bplo_sumregs_gradient_image.restype = C.c_int
        L.bplo_sumregs_gradient.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                            C.c_double, C.c_int, _dp, _dp]
        L.bplo_sumregs_gradient.restype = C.c_int
        L.bplo_rsqrt_nr.argtypes = [C.c_double]
        L.bplo_rsqrt_nr.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def alpha_arg(alpha):
    """-> (array, am, an): scalar -> 1x1; numpy (an, am) array == column-major am x an."""
    a = np.asarray(alpha, dtype=np.float64)
    if a.ndim == 0:
        return a.reshape(1), 1, 1
    a = np.ascontiguousarray(a)
    an, am = a.shape
    return a, am, an


def step_table(maxiter, tau0=5.0, sigma0=0.99 / 5, accel=True):
    tab = np.empty((maxiter, 5))
    lib().bplo_step_table(maxiter, tau0, sigma0, int(accel), _p(tab))
    return tab


def pdhg(f, alpha, maxiter=5000, rho=0.0, tau0=5.0, sigma0=0.99 / 5, accel=True, nthreads=1,
         return_dual=False):
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = alpha_arg(alpha)
    x = np.empty_like(f3)
    y1 = np.empty_like(f3) if return_dual else None
    y2 = np.empty_like(f3) if return_dual else None
    rc = lib().bplo_pdhg(M, N, O, _p(f3), _p(a), am, an, rho, tau0, sigma0, int(accel), maxiter,
                         _p(x), _p(y1), _p(y2), nthreads)
    if rc:
        raise RuntimeError("bplo_pdhg rc=%d" % rc)
    x = x.reshape(f.shape)
    if return_dual:
        return x, y1.reshape(f.shape), y2.reshape(f.shape)
    return x


def pdhg_f32(f, alpha, maxiter=5000, rho=0.0, tau0=5.0, sigma0=0.99 / 5, accel=True):
    """"spec v2f": the recurrence in single precision (bplo_pdhg_f32), the checker of the library's opt-in
    dtype = 32 mode.  Returns the primal widened to float64."""
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = alpha_arg(alpha)
    x = np.empty_like(f3)
    rc = lib().bplo_pdhg_f32(M, N, O, _p(f3), _p(a), am, an, rho, tau0, sigma0, int(accel), maxiter, _p(x))
    if rc:
        raise RuntimeError("bplo_pdhg_f32 rc=%d" % rc)
    return x.reshape(f.shape)


VARIANT_FLAGS = {"x0_zero": 1, "dual_first": 2, "ieee_sqrt_div": 4, "max_form": 8, "omega_of_new_tau": 16}


def pdhg_variant(f, alpha, maxiter=5000, flags=0, L=None, tau0=5.0, sigma0=0.99 / 5, accel=True, nthreads=1,
                 return_dual=False):
    """The restated recurrence with unpinned choices flipped (bplo_pdhg_variant; flags: VARIANT_FLAGS).
    flags = 0, L = sqrt(8) is the oracle's recurrence in unfused arithmetic."""
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = alpha_arg(alpha)
    x = np.empty_like(f3); y1 = np.empty_like(f3); y2 = np.empty_like(f3)
    rc = lib().bplo_pdhg_variant(M, N, O, _p(f3), _p(a), am, an, tau0, sigma0, int(accel), maxiter, int(flags),
                                 float(np.sqrt(8.0) if L is None else L), _p(x), _p(y1), _p(y2), nthreads)
    if rc:
        raise RuntimeError("bplo_pdhg_variant rc=%d" % rc)
    if return_dual:
        return x.reshape(f.shape), y1.reshape(f.shape), y2.reshape(f.shape)
    return x.reshape(f.shape)


def pdhg_opts(f, alpha, maxiter=5000, init=0, order=0, L=None, rho=0.0, tau0=5.0, sigma0=0.99 / 5, accel=True,
              return_dual=False):
    """bplo_pdhg_opts: the oracle's own arithmetic with the run-time choices of bpltv_params.init / order / opnorm."""
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = alpha_arg(alpha)
    x = np.empty_like(f3); y1 = np.empty_like(f3); y2 = np.empty_like(f3)
    fn = lib().bplo_pdhg_opts
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 3 + [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                   C.c_int, C.c_double, _dp, _dp, _dp]
    rc = fn(M, N, O, _p(f3), _p(a), am, an, rho, tau0, sigma0, int(accel), maxiter, int(init), int(order),
            float(np.sqrt(8.0) if L is None else L), _p(x), _p(y1), _p(y2))
    if rc:
        raise RuntimeError("bplo_pdhg_opts rc=%d" % rc)
    if return_dual:
        return x.reshape(f.shape), y1.reshape(f.shape), y2.reshape(f.shape)
    return x.reshape(f.shape)


_native = None
_SO_NATIVE = os.path.join(_HERE, "libbpltv_oracle_native.so")


def native_lib():
    """The -O3 -march=native build of the same file, compiled on the machine that
    runs it.  bench.py's cpu_baseline leg only: never a checker."""
    global _native
    if _native is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libbpltv_oracle_native.so"])
        L = C.CDLL(_SO_NATIVE)
        L.bplo_pdhg.argtypes = lib().bplo_pdhg.argtypes
        L.bplo_pdhg.restype = C.c_int
        L.bplo_pdhg_rows.argtypes = lib().bplo_pdhg_rows.argtypes
        L.bplo_pdhg_rows.restype = C.c_int
        L.bplo_max_threads.restype = C.c_int
        _native = L
    return _native


def pdhg_rows(f, alpha, maxiter=5000, rho=0.0, tau0=5.0, sigma0=0.99 / 5, accel=True, nthreads=1, colblock=8,
              native=False):
    """pdhg() with every iteration spread over images x column blocks (bplo_pdhg_rows): same bits in the
    checker build; `native=True` runs the -O3 -march=native build (timing only)."""
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = alpha_arg(alpha)
    x = np.empty_like(f3)
    L = native_lib() if native else lib()
    rc = L.bplo_pdhg_rows(M, N, O, _p(f3), _p(a), am, an, rho, tau0, sigma0, int(accel), maxiter, _p(x),
                          nthreads, colblock)
    if rc:
        raise RuntimeError("bplo_pdhg_rows rc=%d" % rc)
    return x.reshape(f.shape)


def pdhg_native(f, alpha, maxiter=5000, nthreads=1):
    """bplo_pdhg of the -O3 -march=native build (timing only)."""
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = alpha_arg(alpha)
    x = np.empty_like(f3)
    rc = native_lib().bplo_pdhg(M, N, O, _p(f3), _p(a), am, an, 0.0, 5.0, 0.99 / 5, 1, maxiter, _p(x), None, None,
                                nthreads)
    if rc:
        raise RuntimeError("bplo_pdhg rc=%d" % rc)
    return x.reshape(f.shape)


def cost(u, ubar, per_image=False):
    u = _c(u); ubar = _c(ubar)
    u3 = u.reshape((-1,) + u.shape[-2:])
    O, N, M = u3.shape
    pi = np.empty(O)
    tot = lib().bplo_cost(M, N, O, _p(u3), _p(ubar), _p(pi))
    return (tot, pi) if per_image else tot


def gap(u, y1, y2, f, alpha):
    u = _c(u); y1 = _c(y1); y2 = _c(y2); f = _c(f)
    u3 = u.reshape((-1,) + u.shape[-2:])
    O, N, M = u3.shape
    a, am, an = alpha_arg(alpha)
    out = np.empty(O)
    lib().bplo_gap(M, N, O, _p(u3), _p(y1), _p(y2), _p(f), _p(a), am, an, _p(out))
    return out


def grad_fwd(x):
    x = _c(x); N, M = x.shape
    d1 = np.empty_like(x); d2 = np.empty_like(x)
    lib().bplo_grad_fwd(M, N, _p(x), _p(d1), _p(d2))
    return d1, d2


def grad_fwd_T(y1, y2):
    y1 = _c(y1); y2 = _c(y2); N, M = y1.shape
    r = np.empty_like(y1)
    lib().bplo_grad_fwd_T(M, N, _p(y1), _p(y2), _p(r))
    return r


def patch_upsample(alpha, M, N):
    a, am, an = alpha_arg(alpha)
    out = np.empty((N, M))
    lib().bplo_patch_upsample(_p(a), am, an, M, N, _p(out))
    return out


def patch_adjoint(g, am, an):
    g = _c(g); N, M = g.shape
    out = np.empty((an, am))
    lib().bplo_patch_adjoint(_p(g), M, N, am, an, _p(out))
    return out


def gradient_image(u, ubar, amap, patch=False, reg=False, kappa_cap=1e14, nref=3):
    u = _c(u); ubar = _c(ubar); amap = _c(amap)
    N, M = u.shape
    gpix = np.empty_like(u); p = np.empty_like(u)
    res = C.c_double(0.0)
    rc = lib().bplo_gradient_image(M, N, _p(u), _p(ubar), _p(amap), int(patch), int(reg), kappa_cap, nref,
                                   _p(gpix), _p(p), C.byref(res))
    if rc:
        raise RuntimeError("bplo_gradient_image rc=%d" % rc)
    return gpix, p, res.value


def gradient(alpha, u, ubar, reg=False, kappa_cap=1e14, nref=3, per_image=False):
    u = _c(u); ubar = _c(ubar)
    O, N, M = u.shape
    a, am, an = alpha_arg(alpha)
    out = np.empty(am * an)
    pi = np.empty((O, am * an))
    rc = lib().bplo_gradient(M, N, O, _p(u), _p(ubar), _p(a), am, an, int(reg), kappa_cap, nref, _p(out), _p(pi))
    if rc:
        raise RuntimeError("bplo_gradient rc=%d" % rc)
    g = float(out[0]) if (am == 1 and an == 1) else out.reshape(an, am)
    return (g, pi) if per_image else g


def tv_op_learning_function(x, data, delta, delta_t=1e-6, maxiter=5000, rho=0.0, tau0=5.0,
                            sigma0=0.99 / 5, accel=True, nthreads=1):
    """(u, cost, grad) of /root/reference/src/TVLearningFunctionVec.jl:14-27, C restatement."""
    ubar, f = data
    u = pdhg(f, x, maxiter=maxiter, rho=rho, tau0=tau0, sigma0=sigma0, accel=accel, nthreads=nthreads)
    c = cost(u, ubar)
    g = gradient(x, u, ubar, reg=not (delta > delta_t))
    return u, c, g


# ---------------------------------------------------------------------------------------------------
# Sum of regularisers (oracle/sumregs_oracle.c; /root/reference/src/SumRegsLearningFunction.jl)
# Parameter convention: the Julia Vector x = [a1; a2; a3] is a numpy array of shape (3,); the Julia m x n x 3 array
# is a numpy array of shape (3, n, m) (C-contiguous == the three column-major m x n slices one after the other).
# ---------------------------------------------------------------------------------------------------
def sr_alpha_arg(alpha):
    a = np.ascontiguousarray(alpha, dtype=np.float64)
    if a.ndim == 1:
        if a.shape[0] != 3:
            raise ValueError("sum-of-regularisers parameter vector must have 3 entries")
        return a, 1, 1
    if a.ndim != 3 or a.shape[0] != 3:
        raise ValueError("sum-of-regularisers parameter must have shape (3,) or (3, n, m)")
    return a, a.shape[2], a.shape[1]


def sr_grad(k, x):
    x = _c(x); N, M = x.shape
    d1 = np.empty_like(x); d2 = np.empty_like(x)
    lib().bplo_sr_grad(k, M, N, _p(x), _p(d1), _p(d2))
    return d1, d2


def sr_gradT(k, y1, y2):
    y1 = _c(y1); y2 = _c(y2); N, M = y1.shape
    r = np.empty_like(y1)
    lib().bplo_sr_gradT(k, M, N, _p(y1), _p(y2), _p(r))
    return r


def sumregs_pdhg(f, alpha, maxiter=5000, rho=0.0, tau0=5.0, sigma0=0.99 / 5, accel=True, nthreads=1, return_dual=False):
    f = _c(f)
    f3 = f.reshape((-1,) + f.shape[-2:])
    O, N, M = f3.shape
    a, am, an = sr_alpha_arg(alpha)
    x = np.empty_like(f3)
    y = np.empty((O, 6, N, M)) if return_dual else None
    rc = lib().bplo_sumregs_pdhg(M, N, O, _p(f3), _p(a), am, an, rho, tau0, sigma0, int(accel), maxiter, _p(x), _p(y), nthreads)
    if rc:
        raise RuntimeError("bplo_sumregs_pdhg rc=%d" % rc)
    return (x.reshape(f.shape), y) if return_dual else x.reshape(f.shape)


def sumregs_gap(u, y, f, alpha):
    u = _c(u); y = _c(y); f = _c(f)
    u3 = u.reshape((-1,) + u.shape[-2:])
    O, N, M = u3.shape
    a, am, an = sr_alpha_arg(alpha)
    out = np.empty(O)
    lib().bplo_sumregs_gap(M, N, O, _p(u3), _p(y), _p(f), _p(a), am, an, _p(out))
    return out


def sumregs_gradient(alpha, u, ubar, reg=False, kappa_cap=1e14, nref=3, per_image=False):
    u = _c(u); ubar = _c(ubar)
    O, N, M = u.shape
    a, am, an = sr_alpha_arg(alpha)
    out = np.empty(3 * am * an)
    pi = np.empty((O, 3 * am * an))
    rc = lib().bplo_sumregs_gradient(M, N, O, _p(u), _p(ubar), _p(a), am, an, int(reg), kappa_cap, nref, _p(out), _p(pi))
    if rc:
        raise RuntimeError("bplo_sumregs_gradient rc=%d" % rc)
    g = out.copy() if (am == 1 and an == 1) else out.reshape(3, an, am)
    return (g, pi) if per_image else g


def sumregs_learning_function(x, data, delta, delta_t=1e-3, maxiter=5000, nthreads=1):
    """(u, cost, grad) of /root/reference/src/SumRegsLearningFunction.jl:8-36, C restatement."""
    ubar, f = data
    u = sumregs_pdhg(f, x, maxiter=maxiter, nthreads=nthreads)
    return u, cost(u, ubar), sumregs_gradient(x, u, ubar, reg=not (delta > delta_t))


def rsqrt_nr(x):
    return lib().bplo_rsqrt_nr(float(x))


def max_threads():
    return lib().bplo_max_threads()
