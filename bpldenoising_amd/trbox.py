"""Outer trust-region loop as a host-side *harness* (SURVEY.md section 8f, rank 1).

The reference's `bilevel_learn` (/root/reference/src/TRBox.jl:192-273) takes the learning function as
a plain argument, so the product does not replace it; this module restates it so that "the learned
parameter matches" can be demonstrated end to end without Julia: the same loop is driven once by
the HIP learning function and once by the oracle (tests/test_trbox.py, tests/test_gpu_trbox.py).

The scalar path reproduces the reference *as written*, quirks included:
  * the "Newton" step is `pn = B \\ gx` -- no minus sign (TRBox.jl:63);
  * the scalar `B` update is computed and discarded (`updateBFGS!(B::Real, ...)` returns a new value
    that the caller ignores, :181-186,237), so B stays 0.1 for the whole run;
  * `in_bounds(lb, Δ, .)` is called with Δ as the upper bound (:64,68);
  * `step_to_bound` is the elementwise `max(lb/p, ub/p)` (:149-152);
  * the radius is shrunk a second time when `pred < 0` (:247-249); a step is accepted when ρ > 0 (:251);
  * the iteration stops when Δ < tol after an iteration, or after maxiter (BilevelVisualise.jl:190,246).
The array path (L-BFGS model + CG, TRBox.jl:99-114,135-146) depends on LinearOperators.jl's
`LBFGSOperator` and Krylov.jl's `cg_lanczos`, which are external and unpinned (`Project.toml` has no
`[compat]` entry for either) -- PARITY UNPINNED for the array-alpha outer step.  What the text of the
reference does fix is reproduced literally:
  * `init_rest` builds `LBFGSOperator(length(x[:]))` with the package defaults (:50), restated here as
    LinearOperators.jl's documented defaults: memory 5, `scaling = true` (initial matrix (y'y / y's) I from
    the newest pair), no damping, a pair is stored only when y's > 1e-20 (the operator's own curvature test);
  * `updateBFGS!(B, y, s)` pushes only when  y' (B y) > 0  (:174-179) -- the guard is on the CURRENT model
    applied to y = gx_new - gx, not on y's -- and calls `push!(B, y[:], s[:])`, i.e. with y in the slot
    LinearOperators.jl names `s` and s in the slot it names `y` (:176); the same order is kept here;
  * `newton_step` solves B pn = -gx by conjugate gradients (`cg_lanczos`, :135-141; own CG below, exact in
    <= n steps for the n <= 4 parameters of the shipped configs); `cauchy_step` is (:143-146) verbatim.
"""
import numpy as np

EPS = np.finfo(np.float64).eps

# /root/reference/src/BPLDenoising.jl:306-323 (scalar) and :350-357 (patch)
DEFAULT_PARAMS = dict(maxiter=20, tol=1e-5, eta1=0.25, eta2=0.75, beta1=0.25, beta2=1.9)
SCALAR_START = dict(delta0=0.1, alpha0=0.1)
PATCH_START = dict(delta0=1e-4, alpha0=1e-4 * np.ones((2, 2)))


def norm2(v):
    return float(np.sqrt(np.sum(np.asarray(v, dtype=np.float64) ** 2)))


def get_bounds(x, delta):                      # TRBox.jl:160-164
    lb = np.maximum(-delta, EPS - np.asarray(x, dtype=np.float64))
    ub = delta * np.ones(np.shape(x))
    return lb, ub


def in_bounds(lb, ub, x):                      # TRBox.jl:155-157
    return bool(np.all(x >= lb) and np.all(x <= ub))


def step_to_bound(p, lb, ub):                  # TRBox.jl:149-152
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.maximum(lb / p, ub / p)


class LBFGSOperator:
    """Forward limited-memory BFGS operator B ~ Hessian (own implementation, see module docstring)."""

    def __init__(self, n, mem=5):
        self.n, self.mem = n, mem
        self.S, self.Y = [], []

    def push(self, s, y):
        """LinearOperators.jl `push!(op, s, y)`: store the pair when y's > 1e-20 (the operator's own test)."""
        s = np.asarray(s, dtype=np.float64).ravel(); y = np.asarray(y, dtype=np.float64).ravel()
        if float(y @ s) > 1e-20:
            self.S.append(s.copy()); self.Y.append(y.copy())
            if len(self.S) > self.mem:
                self.S.pop(0); self.Y.pop(0)

    def matvec(self, v):
        v = np.asarray(v, dtype=np.float64).ravel()
        if not self.S:
            return v.copy()
        gamma = float(self.Y[-1] @ self.Y[-1]) / float(self.Y[-1] @ self.S[-1])
        a, b = [], []                           # B = gamma I - sum b b' + sum a a'
        for s, y in zip(self.S, self.Y):
            Bs = gamma * s
            for ai, bi in zip(a, b):
                Bs = Bs - bi * float(bi @ s) + ai * float(ai @ s)
            b.append(Bs / np.sqrt(float(s @ Bs)))
            a.append(y / np.sqrt(float(y @ s)))
        out = gamma * v
        for ai, bi in zip(a, b):
            out = out - bi * float(bi @ v) + ai * float(ai @ v)
        return out


def _cg(B, rhs, tol=1e-8, maxit=None):         # stands in for Krylov.cg_lanczos (TRBox.jl:136)
    n = rhs.size
    x = np.zeros(n); r = rhs.copy(); p = r.copy()
    rs = float(r @ r)
    for _ in range(maxit or 2 * n):
        if np.sqrt(rs) <= tol * max(1.0, norm2(rhs)):
            break
        Bp = B.matvec(p)
        a = rs / float(p @ Bp)
        x += a * p; r -= a * Bp
        rs_new = float(r @ r)
        p = r + (rs_new / rs) * p
        rs = rs_new
    return x


def updateBFGS(B, y, s):
    """/root/reference/src/TRBox.jl:174-179: `if y[:]'*(B*y[:]) > 0  push!(B, y[:], s[:])  end`."""
    yv = np.asarray(y, dtype=np.float64).ravel()
    if float(yv @ B.matvec(yv)) > 0:
        B.push(yv, np.asarray(s, dtype=np.float64).ravel())     # (y, s) in the reference's order
    return B


def dogleg_box(x, gx, B, delta):
    lb, _ = get_bounds(x, delta)
    if np.ndim(x) == 0:                        # TRBox.jl:60-76
        pn = gx / B
        if in_bounds(lb, delta, pn):
            return pn
        p = -(abs(gx) ** 2 / (gx * (B * gx))) * gx
        if not in_bounds(lb, delta, p):
            d = p / abs(p)
            return d * step_to_bound(d, lb, delta)
        return p + step_to_bound(pn - p, lb, delta) * (pn - p)
    g = np.asarray(gx, dtype=np.float64)       # TRBox.jl:99-114
    pn = _cg(B, -g.ravel()).reshape(g.shape)
    if in_bounds(lb, delta, pn):
        return pn
    Bg = B.matvec(g.ravel())
    p = (-(norm2(g) ** 2 / float(g.ravel() @ Bg)) * g.ravel()).reshape(g.shape)
    if not in_bounds(lb, delta, p):
        d = p / norm2(p)
        return d * step_to_bound(d, lb, delta)
    return p + step_to_bound(pn - p, lb, delta) * (pn - p)


def pred(B, p, gx):                            # TRBox.jl:166-172
    if np.ndim(p) == 0:
        return -p * gx - 0.5 * p * B * p
    pv = np.ravel(p)
    return float(-pv @ np.ravel(gx) - 0.5 * pv @ B.matvec(pv))


def bilevel_learn(ds, learning_function, xinit, delta0, maxiter=20, tol=1e-5, eta1=0.25, eta2=0.75,
                  beta1=0.25, beta2=1.9, log=None, **lf_kwargs):
    """x, u, history = bilevel_learn((ubar, f), learning_function, xinit, delta0, ...)

    learning_function(x, ds, delta, **lf_kwargs) -> (u, cost, grad), e.g.
    bpldenoising_amd.tv_op_learning_function.  history: one dict per outer iteration with the
    fields of BilevelLogEntry (BilevelVisualise.jl:39-46)."""
    scalar = np.ndim(xinit) == 0
    x = float(xinit) if scalar else np.array(xinit, dtype=np.float64)
    delta = float(delta0)
    u, fx, gx = learning_function(x, ds, delta, **lf_kwargs)           # init_rest, TRBox.jl:34-52
    B = 0.1 if scalar else LBFGSOperator(np.size(x))
    residual = x - x
    hist = []
    for it in range(1, maxiter + 1):
        p = dogleg_box(x, gx, B, delta)                                # TRBox.jl:221
        xb = x + p
        ub_, fxb, gxb = learning_function(xb, ds, delta, **lf_kwargs)  # :227
        predf = pred(B, p, gx)
        with np.errstate(divide="ignore", invalid="ignore"):
            rho = (fx - fxb) / predf                                   # :230
        if not scalar:
            updateBFGS(B, np.asarray(gxb) - np.asarray(gx), p)          # :237 -> :174-179
        if rho < eta1:                                                 # :239-245
            delta = beta1 * delta
        elif rho > eta2:
            if norm2(p) > 0.8 * delta:
                delta = beta2 * delta
        if predf < 0:                                                  # :247-249
            delta = beta1 * delta
        if rho > 0:                                                    # :251-257
            residual = x - xb
            x, u, fx, gx = xb, ub_, fxb, gxb
        entry = dict(iter=it, x=np.array(x).tolist(), function_value=float(fx), gradient_value=norm2(gx),
                     radius_value=delta, stopping_criteria=norm2(residual), rho=float(rho), pred=float(predf))
        hist.append(entry)
        if log:
            log(entry)
        if delta < tol:                                                # BilevelVisualise.jl:246
            break
    return x, u, hist
