"""numpy/scipy twin of the sum-of-regularisers learning function  --  TEST INFRASTRUCTURE ONLY.

Literal scipy.sparse assembly of the adjoint systems of /root/reference/src/SumRegsLearningFunction.jl:
    :264-327  sumregs_gradient (vector x)          the 7n^2 saddle system [I -G1' -G2' -G3'; ...] (:318-324)
    :330-407  sumregs_gradient (patch x)           the same with spdiagm([x_k; x_k]) (:388-394), per-pixel p .* (G_k' ...)
    :112-167  sumregs_gradient_reg (vector x)      gamma = 1e3, (I + sum x_k G_k'(B_k - C_k)G_k) \\ (ubar - u)
    :195-262  sumregs_gradient_reg (patch x)       gamma = 1e8, row scaling x_k[:] .* G_k'(B_k - C_k)G_k (non-symmetric)
solved by sparse LU (+ extended-precision refinement for the saddle systems, as in np_twin.py) -- an independent
check of oracle/sumregs_oracle.c, whose reduced banded systems the HIP kernels reproduce.

PARITY UNPINNED: the three operators' matrices (`matrix(op, n)` of the absent package VariationalImaging) are
restated here with the boundary conventions of sumregs_oracle.c (forward / backward differences with a zero row at
the far / near border, centred half-differences with a mirrored border); the PDHG recurrence is a numpy twin of the
same file.  Array convention as np_twin.py: Julia (M, N) column major == numpy (N, M) C order; the parameter
x[:, :, k] (m x n x 3) is numpy (3, n, m); the vector x = [a1; a2; a3] is numpy (3,).
"""
import math
import numpy as np

from . import np_twin as T

EPS = np.finfo(np.float64).eps
SR_L = math.sqrt(18.0)


def _sp():
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    return sp, spla


def diff_matrix(kind, n):
    """1-D difference matrix (n x n) of operator `kind` (0 fwd, 1 bwd, 2 ctr)."""
    sp, _ = _sp()
    d = sp.lil_matrix((n, n))
    for i in range(n):
        if kind == 0 and i < n - 1:
            d[i, i] = -1.0; d[i, i + 1] = 1.0
        elif kind == 1 and i > 0:
            d[i, i] = 1.0; d[i, i - 1] = -1.0
        elif kind == 2:
            ip, im = min(i + 1, n - 1), max(i - 1, 0)
            if ip != im:
                d[i, ip] += 0.5; d[i, im] -= 0.5
    return d.tocsr()


def grad_matrix(kind, M, N):
    """matrix(op_k, n): sparse [D1; D2] (2MN x MN) on the column-major vec (q = i + M*j)."""
    sp, _ = _sp()
    D1 = sp.kron(sp.identity(N), diff_matrix(kind, M), format="csr")
    D2 = sp.kron(diff_matrix(kind, N), sp.identity(M), format="csr")
    return sp.vstack([D1, D2], format="csr")


def alpha_maps(alpha, M, N):
    a = np.asarray(alpha, dtype=np.float64)
    if a.ndim == 1:
        return np.stack([np.full((N, M), a[k]) for k in range(3)])
    return np.stack([T.patch_upsample(a[k], M, N) for k in range(3)])


def pdhg(f, alpha, maxiter=5000, tau0=5.0, sigma0=0.99 / 5, accel=True):
    """numpy twin of sumregs_oracle.c: sr_pdhg_image (same recurrence; plain numpy arithmetic, so it agrees with
    the C oracle to rounding, not bit for bit).  f: (N, M) or (O, N, M)."""
    f = np.asarray(f, dtype=np.float64)
    N, M = f.shape[-2:]
    am = alpha_maps(alpha, M, N)
    G = [grad_matrix(k, M, N) for k in range(3)]
    shp = f.shape
    fl = f.reshape(-1, N * M)
    out = np.empty_like(fl)
    n = N * M
    for o in range(fl.shape[0]):
        fo = fl[o]
        x = fo.copy()
        y = [np.zeros(2 * n) for _ in range(3)]
        tau, sigma = tau0 / SR_L, sigma0 / SR_L
        for _ in range(maxiter):
            omega = 1.0 / math.sqrt(1.0 + 2.0 * tau) if accel else 1.0
            div = (G[0].T @ y[0] + G[1].T @ y[1]) + G[2].T @ y[2]
            xo = x
            x = (x - tau * (div - fo)) / (1.0 + tau)
            xb = (1.0 + omega) * x - omega * xo
            for k in range(3):
                yk = y[k] + sigma * (G[k] @ xb)
                a = am[k].reshape(-1)
                nrm = np.sqrt(yk[:n] ** 2 + yk[n:] ** 2)
                sc = np.where(nrm > a, a / np.where(nrm > a, nrm, 1.0), 1.0)
                y[k] = yk * np.concatenate([sc, sc])
            if accel:
                tau, sigma = tau * omega, sigma / omega
        out[o] = x
    return out.reshape(shp)


def _pieces(kind, u, gamma=None, tol=1e-12):
    """Per-operator quantities of the reference: G, Gu, act/inact diagonals, Den, prodKuKu."""
    sp, _ = _sp()
    N, M = u.shape
    G = grad_matrix(kind, M, N)
    Gu = G @ u.reshape(-1)
    nGu = T.xi(Gu)
    if gamma is None:
        act = (nGu < tol).astype(np.float64)                    # :274
        inact = 1.0 - act
        den = inact * nGu + act                                  # :280
    else:
        act = (np.maximum(0.0, nGu - 1.0 / gamma) != 0).astype(np.float64)   # :122-123 ("act" = |Gu| > 1/gamma)
        inact = 1.0 - act
        den = act * nGu + inact                                  # :127
    return G, Gu, act, inact, den, T.prodesc(Gu / den ** 3, Gu)


def gradient_image(alpha, u, ubar, refine=8):
    """Literal :264-327 (vector x) / :330-407 (array x, returns the three (N, M) maps before calc_adjoint)."""
    sp, spla = _sp()
    N, M = u.shape
    n2 = M * N
    a = np.asarray(alpha, dtype=np.float64)
    patch = a.ndim == 3
    maps = alpha_maps(a, M, N)
    rows = [[sp.identity(n2)], ]
    blocks = []
    Z = sp.csr_matrix((2 * n2, 2 * n2))
    pcs = [_pieces(k, u) for k in range(3)]
    for k in range(3):
        G, Gu, act, inact, den, P = pcs[k]
        rows[0].append(-G.T)
        Den = sp.diags(1.0 / den)
        X = sp.diags(np.concatenate([maps[k].reshape(-1)] * 2)) if patch else float(a[k]) * sp.identity(2 * n2)
        lower = sp.diags(act) @ G + sp.diags(inact) @ X @ (Den - P) @ G
        r = [lower] + [Z] * 3
        r[1 + k] = sp.diags(inact) + EPS * sp.diags(act)
        blocks.append(r)
    Adj = sp.bmat(rows + blocks, format="csc")
    track = np.concatenate([u.reshape(-1) - ubar.reshape(-1), np.zeros(6 * n2)])
    mult = T.solve_refined(Adj, track, refine) if refine else spla.spsolve(Adj, track)
    p = mult[:n2]
    out = []
    for k in range(3):
        G, Gu, act, inact, den, P = pcs[k]
        w = G.T @ (inact * (1.0 / den) * Gu)
        out.append(-(p * w).reshape(N, M) if patch else -float(p @ w))
    return (np.stack(out) if patch else np.array(out)), p


def gradient_reg_image(alpha, u, ubar):
    """Literal :112-167 (vector x, gamma = 1e3) / :195-262 (array x, gamma = 1e8)."""
    sp, spla = _sp()
    N, M = u.shape
    n2 = M * N
    a = np.asarray(alpha, dtype=np.float64)
    patch = a.ndim == 3
    gamma = 1e8 if patch else 1e3
    maps = alpha_maps(a, M, N)
    A = sp.identity(n2, format="csr")
    pcs = [_pieces(k, u, gamma=gamma) for k in range(3)]
    for k in range(3):
        G, Gu, act, inact, den, P = pcs[k]
        B = gamma * sp.diags(inact)
        C = sp.diags(act) @ (P - sp.diags(1.0 / den))
        K = G.T @ (B - C) @ G
        A = A + (sp.diags(maps[k].reshape(-1)) @ K if patch else float(a[k]) * K)
    p = spla.spsolve(A.tocsc(), ubar.reshape(-1) - u.reshape(-1))
    out = []
    for k in range(3):
        G, Gu, act, inact, den, P = pcs[k]
        w = G.T @ (act * (1.0 / den) * Gu + gamma * inact * Gu)
        out.append((p * w).reshape(N, M) if patch else float(p @ w))
    return (np.stack(out) if patch else np.array(out)), p


def batch_gradient(alpha, u, ubar, reg=False, refine=8):
    """Batch wrappers :87-110, :169-193: sum over images; array x -> calc_adjoint per slice, shape (3, n, m)."""
    u = np.asarray(u, dtype=np.float64); ubar = np.asarray(ubar, dtype=np.float64)
    a = np.asarray(alpha, dtype=np.float64)
    out = np.zeros(a.shape)
    for k in range(u.shape[0]):
        g = gradient_reg_image(a, u[k], ubar[k])[0] if reg else gradient_image(a, u[k], ubar[k], refine)[0]
        if a.ndim == 1:
            out += g
        else:
            _, n, m = a.shape
            out += np.stack([T.patch_adjoint(g[s], m, n) for s in range(3)])
    return out
