"""numpy/scipy twin of the TV learning-function path  --  TEST INFRASTRUCTURE ONLY.

This file is part of the *oracle*: a CPU restatement of the reference algorithm used as the
checker in `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.  Nothing
under `bpldenoising_amd/` may import it; the product path is the HIP library.

PARITY UNPINNED BY THE REFERENCE: the reference (Julia) cannot run here, its PDHG loop lives in
the un-vendored, un-pinned package VariationalImaging (`op_denoise_pdps`), and its tests hold
no expected values.  What this file pins instead:
  * the adjoint gradients are the *literal* sparse systems of
    /root/reference/src/TVLearningFunctionVec.jl:98-135, :137-161, :192-215, :219-254
    assembled with scipy.sparse and solved with a sparse direct LU (SuperLU standing in for
    UMFPACK's `\\`),
  * the PDHG recurrence is the accelerated Chambolle-Pock iteration restated in SURVEY.md
    section 8(a) row A2 with the constants of TVLearningFunctionVec.jl:33-43,
  * mathematical certificates (duality gap, adjointness, finite differences) in tests/.

Array convention (matches the C ABI): a Julia `Array{Float64,3}` of size (M,N,O), column major,
is a C-contiguous numpy array of shape (O, N, M); `a[k, j, i]` is Julia's `a[i+1, j+1, k+1]`.
Gradient component 1 differences Julia dimension 1 (numpy axis -1, the contiguous one),
component 2 Julia dimension 2 (numpy axis -2)  --  /root/reference/src/TVLearningFunctionOp.jl:26-36.
"""
import math
import numpy as np

EPS = np.finfo(np.float64).eps
OPNORM_EST = math.sqrt(8.0)          # bound on ||grad||, SURVEY.md 8(a) A2/A4

DEFAULT_PARAMS = dict(rho=0.0, tau0=5.0, sigma0=0.99 / 5, accel=True, maxiter=5000)
# ^ /root/reference/src/TVLearningFunctionVec.jl:33-43  (TVDenoise uses maxiter=10000,
#   /root/reference/src/BPLDenoising.jl:41-59)


# --------------------------------------------------------------------------- operators (A4)
def grad_fwd(x):
    """Forward differences, Neumann boundary (FwdGradientOp).  x: (..., N, M) -> d1, d2."""
    d1 = np.zeros_like(x)
    d2 = np.zeros_like(x)
    d1[..., :, :-1] = x[..., :, 1:] - x[..., :, :-1]
    d2[..., :-1, :] = x[..., 1:, :] - x[..., :-1, :]
    return d1, d2


def grad_fwd_T(y1, y2):
    """Adjoint of grad_fwd (= minus the backward-difference divergence)."""
    M = y1.shape[-1]
    N = y1.shape[-2]
    r = np.zeros_like(y1)
    # component 1: (G1^T y)(i) = y(i-1) - y(i), with y(M-1) treated as 0
    r[..., :, : M - 1] -= y1[..., :, : M - 1]
    r[..., :, 1:] += y1[..., :, : M - 1]
    r[..., : N - 1, :] -= y2[..., : N - 1, :]
    r[..., 1:, :] += y2[..., : N - 1, :]
    return r


def patch_upsample(x, M, N):
    """PatchOp: m x n parameter -> M x N piecewise-constant map (numpy shape (N, M)).

    x is given in Julia orientation as numpy shape (n, m) (C-contiguous == column-major m x n).
    alpha_map[j, i] = x[(j*n)//N, (i*m)//M].  /root/reference/src/TVLearningFunctionVec.jl:57-60
    (PatchOp itself is external; divisibility behaviour for M % m != 0 is unpinned.)
    """
    x = np.asarray(x, dtype=np.float64)
    n, m = x.shape
    ii = (np.arange(M) * m) // M
    jj = (np.arange(N) * n) // N
    return np.ascontiguousarray(x[np.ix_(jj, ii)])


def patch_adjoint(g, m, n):
    """calc_adjoint(PatchOp, g): sum a (N, M) pixel map over each patch -> (n, m)."""
    N, M = g.shape
    ii = (np.arange(M) * m) // M
    jj = (np.arange(N) * n) // N
    out = np.zeros((n, m))
    np.add.at(out, (jj[:, None], ii[None, :]), g)
    return out


def alpha_to_map(alpha, M, N):
    a = np.asarray(alpha, dtype=np.float64)
    if a.ndim == 0:
        return np.full((N, M), float(a))
    if a.shape == (N, M):
        return a
    return patch_upsample(a, M, N)


# --------------------------------------------------------------------------- PDHG (A2, A3)
def step_table(maxiter, tau0=5.0, sigma0=0.99 / 5, accel=True):
    """Data-independent step sizes: rows (tau_k, sigma_k, omega_k) used in iteration k."""
    tau = tau0 / OPNORM_EST
    sigma = sigma0 / OPNORM_EST
    gamma = 1.0
    tab = np.empty((maxiter, 3))
    for k in range(maxiter):
        omega = 1.0 / math.sqrt(1.0 + 2.0 * gamma * tau) if accel else 1.0
        tab[k] = (tau, sigma, omega)
        if accel:
            tau, sigma = tau * omega, sigma / omega
    return tab


def rsqrt_nr(n2):
    """1/sqrt by four Newton steps from an integer seed ("spec v2", oracle/bpltv_oracle.c); numpy
    has no fma, so this agrees with the C oracle to a few ulp, not bit for bit."""
    n2 = np.ascontiguousarray(n2, dtype=np.float64)
    r = (np.uint64(0x5FE6EB50C7B537A9) - (n2.view(np.uint64) >> np.uint64(1))).view(np.float64)
    h = 0.5 * n2
    for _ in range(4):
        r = r * (1.5 - h * (r * r))
    return r


def pdhg_denoise(f, alpha, maxiter=5000, tau0=5.0, sigma0=0.99 / 5, accel=True, rho=0.0,
                 return_dual=False):
    """ROF denoising of a batch by accelerated PDHG, fixed iteration count.

    f: (O, N, M) or (N, M); alpha: scalar, (n, m) patch parameter or (N, M) map, shared by the
    batch.  Restates SURVEY.md 8(a) A2:  x=f, y=0; per iteration
        x_old=x; x=(x - tau*(G^T y - f))/(1+tau); xb=(1+omega)x - omega*x_old;
        y=(y + sigma*G xb)/(1 + sigma*rho/alpha); y <- proj_{|y_ij|<=alpha_ij}; tau*=omega; sigma/=omega
    (projection factor alpha*rsqrt_nr(|y|^2), see rsqrt_nr)
    """
    f = np.asarray(f, dtype=np.float64)
    N, M = f.shape[-2:]
    amap = alpha_to_map(alpha, M, N)
    tab = step_table(maxiter, tau0, sigma0, accel)
    x = f.copy()
    y1 = np.zeros_like(f)
    y2 = np.zeros_like(f)
    a2 = amap * amap
    for k in range(maxiter):
        tau, sigma, omega = tab[k]
        div = grad_fwd_T(y1, y2)
        xo = x
        x = (x - tau * (div - f)) / (1.0 + tau)
        xb = (1.0 + omega) * x - omega * xo
        d1, d2 = grad_fwd(xb)
        y1 = y1 + sigma * d1
        y2 = y2 + sigma * d2
        if rho != 0.0:
            den = 1.0 + sigma * rho / amap
            y1 = y1 / den
            y2 = y2 / den
        n2 = y1 * y1 + y2 * y2
        with np.errstate(all="ignore"):
            v = np.where(n2 > a2, amap * rsqrt_nr(np.where(n2 > a2, n2, 1.0)), 1.0)
        y1 = y1 * v
        y2 = y2 * v
    if return_dual:
        return x, y1, y2
    return x


def rof_primal(u, f, amap):
    d1, d2 = grad_fwd(u)
    return 0.5 * np.sum((u - f) ** 2, axis=(-1, -2)) + np.sum(amap * np.sqrt(d1 * d1 + d2 * d2), axis=(-1, -2))


def rof_dual(y1, y2, f):
    w = grad_fwd_T(y1, y2)
    return 0.5 * np.sum(f * f, axis=(-1, -2)) - 0.5 * np.sum((f - w) ** 2, axis=(-1, -2))


def rof_gap(u, y1, y2, f, alpha):
    """Duality gap per image; >= 0.5*||u-u*||^2 for feasible y."""
    N, M = f.shape[-2:]
    amap = alpha_to_map(alpha, M, N)
    return rof_primal(u, f, amap) - rof_dual(y1, y2, f)


def l2_cost(u, ubar):
    """cost = 0.5*norm2^2(u - ubar) over the whole batch  (TVLearningFunctionVec.jl:20)."""
    d = np.asarray(u, dtype=np.float64) - np.asarray(ubar, dtype=np.float64)
    return 0.5 * float(np.sum(d * d))


# --------------------------------------------------------------------------- adjoint gradients (A6-A8, A10)
def _sp():
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    return sp, spla


def grad_matrix(M, N):
    """matrix(op, n): sparse [D1; D2] (2MN x MN) on column-major vec (k = i + M*j)."""
    sp, _ = _sp()

    def fwd(n):
        d = sp.lil_matrix((n, n))
        for i in range(n - 1):
            d[i, i] = -1.0
            d[i, i + 1] = 1.0
        return d.tocsr()

    D1 = sp.kron(sp.identity(N), fwd(M), format="csr")
    D2 = sp.kron(fwd(N), sp.identity(M), format="csr")
    return sp.vstack([D1, D2], format="csr")


def xi(v):
    """Per-pixel Euclidean norm of a stacked [v1; v2], duplicated in both halves."""
    n = v.size // 2
    r = np.sqrt(v[:n] ** 2 + v[n:] ** 2)
    return np.concatenate([r, r])


def prodesc(a, b):
    """2x2 block matrix of diagonals: block(k,l) = diag(a_k * b_l)."""
    sp, _ = _sp()
    n = a.size // 2
    a1, a2, b1, b2 = a[:n], a[n:], b[:n], b[n:]
    return sp.bmat([[sp.diags(a1 * b1), sp.diags(a1 * b2)],
                    [sp.diags(a2 * b1), sp.diags(a2 * b2)]], format="csr")


def scalarprod(v, w):
    n = v.size // 2
    return v[:n] * w[:n] + v[n:] * w[n:]


def solve_refined(A, b, iters=10):
    """Sparse LU solve of A x = b followed by iterative refinement with residuals accumulated
    in extended precision (numpy longdouble).  The reference's saddle systems are so badly
    conditioned (entries 1/|grad u| up to 1e12 next to eps()) that a plain double-precision LU
    -- SuperLU here, UMFPACK behind Julia's `\\` -- returns the gradient functional with a
    relative scatter of ~1e-3 (tests/test_oracle_gradient.py documents it); the refined solve is
    the exact solution of the literal system to ~1e-8 and is what the golden fixtures hold."""
    _, spla = _sp()
    A = A.tocsr()
    lu = spla.splu(A.tocsc())
    x = lu.solve(b).astype(np.longdouble)
    data = A.data.astype(np.longdouble)
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    bl = b.astype(np.longdouble)
    for _ in range(iters):
        r = np.zeros(A.shape[0], dtype=np.longdouble)
        np.add.at(r, rows, data * x[A.indices])
        r = bl - r
        x = x + lu.solve(np.asarray(r, dtype=np.float64)).astype(np.longdouble)
    return np.asarray(x, dtype=np.float64)


def gradient_scalar_image(alpha, u, ubar, refine=0):
    """Literal /root/reference/src/TVLearningFunctionVec.jl:98-135 for one image (N, M).
    refine > 0: extended-precision refinement sweeps (see solve_refined)."""
    sp, spla = _sp()
    N, M = u.shape
    n2 = M * N
    G = grad_matrix(M, N)
    uv = u.reshape(-1)
    Gu = G @ uv
    nGu = xi(Gu)
    act = (nGu < 1e-12).astype(np.float64)
    inact = 1.0 - act
    Act = sp.diags(act)
    Inact = sp.diags(inact)
    den = inact * nGu + act
    Den = sp.diags(1.0 / den)
    P = prodesc(Gu / den ** 3, Gu)
    Adj = sp.bmat([[sp.identity(n2), -G.T],
                   [Act @ G + Inact @ (alpha * (Den - P)) @ G, Inact + EPS * Act]], format="csc")
    track = np.concatenate([uv - ubar.reshape(-1), np.zeros(2 * n2)])
    mult = solve_refined(Adj, track, refine) if refine else spla.spsolve(Adj, track)
    p = mult[:n2]
    g = np.sum(scalarprod(G @ p, inact * (1.0 / den) * Gu))
    return -g, p


def gradient_patch_image(amap, u, ubar, refine=0):
    """Literal TVLearningFunctionVec.jl:219-252 for one image; returns the (N, M) pixel map
    before calc_adjoint."""
    sp, spla = _sp()
    N, M = u.shape
    n2 = M * N
    G = grad_matrix(M, N)
    uv = u.reshape(-1)
    Gu = G @ uv
    nGu = xi(Gu)
    act = (nGu < 1e-12).astype(np.float64)
    inact = 1.0 - act
    Act = sp.diags(act)
    Inact = sp.diags(inact)
    den = inact * nGu + act
    Den = sp.diags(1.0 / den)
    P = prodesc(Gu / den ** 3, Gu)
    av = amap.reshape(-1)
    A2 = sp.diags(np.concatenate([av, av]))
    Adj = sp.bmat([[sp.identity(n2), -G.T],
                   [Act @ G + Inact @ A2 @ (Den - P) @ G, Inact + math.sqrt(EPS) * Act]], format="csc")
    track = np.concatenate([uv - ubar.reshape(-1), np.zeros(2 * n2)])
    mult = solve_refined(Adj, track, refine) if refine else spla.spsolve(Adj, track)
    p = mult[:n2]
    g = -scalarprod(G @ p, inact * (1.0 / den) * Gu)
    return g.reshape(N, M), p


def gradient_reg_scalar_image(alpha, u, ubar, gamma=1e8):
    """Literal TVLearningFunctionVec.jl:137-161."""
    sp, spla = _sp()
    N, M = u.shape
    n2 = M * N
    G = grad_matrix(M, N)
    uv = u.reshape(-1)
    Gu = G @ uv
    nGu = xi(Gu)
    act = (np.maximum(0.0, nGu - 1.0 / gamma) != 0).astype(np.float64)
    inact = 1.0 - act
    Act = sp.diags(act)
    den = act * nGu + inact
    Den = sp.diags(1.0 / den)
    P = prodesc(Gu / den ** 3, Gu)
    B = gamma * sp.diags(inact)
    C = Act @ (P - Den)
    A = sp.identity(n2) + alpha * (G.T @ (B - C) @ G)
    p = spla.spsolve(A.tocsc(), ubar.reshape(-1) - uv)
    g = np.sum(scalarprod(G @ p, act * (1.0 / den) * Gu + gamma * inact * Gu))
    return g, p


def gradient_reg_patch_image(amap, u, ubar, gamma=1e8):
    """Literal TVLearningFunctionVec.jl:192-213; returns (N, M) map before calc_adjoint."""
    sp, spla = _sp()
    N, M = u.shape
    n2 = M * N
    G = grad_matrix(M, N)
    uv = u.reshape(-1)
    Gu = G @ uv
    nGu = xi(Gu)
    act = (np.maximum(0.0, nGu - 1.0 / gamma) != 0).astype(np.float64)
    inact = 1.0 - act
    Act = sp.diags(act)
    den = act * nGu + inact
    Den = sp.diags(1.0 / den)
    P = prodesc(Gu / den ** 3, Gu)
    B = gamma * sp.diags(inact)
    C = Act @ (P - Den)
    A = sp.identity(n2) + sp.diags(amap.reshape(-1)) @ (G.T @ (B - C) @ G)
    p = spla.spsolve(A.tocsc(), ubar.reshape(-1) - uv)
    g = p * (G.T @ (act * (1.0 / den) * Gu + gamma * inact * Gu))
    return g.reshape(N, M), p


def batch_gradient(alpha, u, ubar, reg=False, refine=0):
    """Batch wrappers TVLearningFunctionVec.jl:72-96 (scalar) and :163-190 (patch)."""
    u = np.asarray(u, dtype=np.float64)
    ubar = np.asarray(ubar, dtype=np.float64)
    O, N, M = u.shape
    a = np.asarray(alpha, dtype=np.float64)
    if a.ndim == 0:
        if reg:
            return float(sum(gradient_reg_scalar_image(float(a), u[k], ubar[k])[0] for k in range(O)))
        return float(sum(gradient_scalar_image(float(a), u[k], ubar[k], refine)[0] for k in range(O)))
    n, m = a.shape
    amap = alpha_to_map(a, M, N)
    out = np.zeros((n, m))
    for k in range(O):
        if reg:
            gp = gradient_reg_patch_image(amap, u[k], ubar[k])[0]
        else:
            gp = gradient_patch_image(amap, u[k], ubar[k], refine)[0]
        out += patch_adjoint(gp, m, n)
    return out


def tv_op_learning_function(x, data, delta, delta_t=1e-6, **kw):
    """Twin of /root/reference/src/TVLearningFunctionVec.jl:14-27 -> (u, cost, grad)."""
    ubar, f = data
    prm = dict(DEFAULT_PARAMS)
    prm.update(kw)
    u = pdhg_denoise(f, x, **prm)
    cost = l2_cost(u, ubar)
    grad = batch_gradient(x, u, ubar, reg=not (delta > delta_t))
    return u, cost, grad


# --------------------------------------------------------------------------- data
def load_dataset(npz_path, name, num_samples=None):
    """Reference dataset -> (ubar, f) float64 (O, N, M) in [0,1]  (Datasets.jl:54-65 semantics:
    gray/255, Julia matrix [row, col] => numpy (N, M) is the transposed PIL image)."""
    z = np.load(npz_path)
    t = z[name + "/true"].astype(np.float64) / 255.0
    d = z[name + "/data"].astype(np.float64) / 255.0
    t = np.ascontiguousarray(np.transpose(t, (0, 2, 1)))
    d = np.ascontiguousarray(np.transpose(d, (0, 2, 1)))
    if num_samples is not None:
        t, d = t[:num_samples], d[:num_samples]
    return t, d
