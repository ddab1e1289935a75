"""Numpy prototype of block cyclic reduction (nested-dissection ordered block Cholesky) for the
reduced adjoint system -- de-risks the HIP implementation (explicit inverses of the diagonal
Cholesky factors, refinement with the sparse matrix).  Dev tool, imports oracle/ (test infra)."""
import sys, time
import numpy as np
import scipy.sparse as sp
import scipy.linalg as sl
sys.path.insert(0, "/root/repo")
from oracle import c_oracle as co, np_twin as nt


def assemble(u, ubar, amap, patch=False, kappa_cap=1e14):
    N, M = u.shape            # python arrays are (N, M): row j = image column j, contiguous i
    n = M * N
    uf = u.ravel()
    idx = np.arange(n).reshape(N, M)
    g1 = np.zeros((N, M)); g2 = np.zeros((N, M))
    g1[:, :-1] = u[:, 1:] - u[:, :-1]
    g2[:-1, :] = u[1:, :] - u[:-1, :]
    ng = np.sqrt(g1 * g1 + g2 * g2)
    act = ng < 1e-12
    eps = np.finfo(float).eps
    kp = min(1.0 / (np.sqrt(eps) if patch else eps), kappa_cap)
    ngs = np.where(act, 1.0, ng)
    t1 = np.where(act, 0.0, -g2 / ngs); t2 = np.where(act, 0.0, g1 / ngs)
    c = np.where(act, 0.0, amap / ngs)
    kap = np.where(act, kp, 0.0)
    h1 = np.where(act, 0.0, g1 / ngs); h2 = np.where(act, 0.0, g2 / ngs)
    hb = np.zeros((N, M), bool); hb[:, :-1] = True
    hc = np.zeros((N, M), bool); hc[:-1, :] = True
    a = idx.ravel()
    # B rows: b_k^T p = t1*(p[a+1]-p[a]) [hb] + t2*(p[a+M]-p[a]) [hc]
    e1 = np.where(hb, t1, 0.0).ravel(); e2 = np.where(hc, t2, 0.0).ravel()
    rows = np.concatenate([a, a[hb.ravel()], a[hc.ravel()]])
    cols = np.concatenate([a, a[hb.ravel()] + 1, a[hc.ravel()] + M])
    vals = np.concatenate([-(e1 + e2), e1[hb.ravel()], e2[hc.ravel()]])
    B = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    # G1, G2 rows
    r1 = a[hb.ravel()]
    G1 = sp.csr_matrix((np.concatenate([-np.ones(r1.size), np.ones(r1.size)]),
                        (np.concatenate([r1, r1]), np.concatenate([r1, r1 + 1]))), shape=(n, n))
    r2 = a[hc.ravel()]
    G2 = sp.csr_matrix((np.concatenate([-np.ones(r2.size), np.ones(r2.size)]),
                        (np.concatenate([r2, r2]), np.concatenate([r2, r2 + M]))), shape=(n, n))
    A = sp.identity(n) + B.T @ sp.diags(c.ravel()) @ B + G1.T @ sp.diags(kap.ravel()) @ G1 + G2.T @ sp.diags(kap.ravel()) @ G2
    rhs = (u - ubar).ravel()
    def apply(p):
        P = p.reshape(N, M)
        d1 = np.zeros((N, M)); d2 = np.zeros((N, M))
        d1[:, :-1] = P[:, 1:] - P[:, :-1]; d2[:-1, :] = P[1:, :] - P[:-1, :]
        bp = t1 * d1 + t2 * d2
        w1 = c * bp * t1 + kap * d1; w2 = c * bp * t2 + kap * d2
        w1[:, -1] = 0.0; w2[-1, :] = 0.0
        gt = -w1 - w2
        gt[:, 1:] += w1[:, :-1]; gt[1:, :] += w2[:-1, :]
        return (P + gt).ravel()
    return A.tocsr(), rhs, h1, h2, apply


def bcr_factor(A, M, N, explicit_inverse=True):
    """returns per-level records; D, C dense blocks."""
    A = A.tocsr()
    D = [A[j * M:(j + 1) * M, j * M:(j + 1) * M].toarray() for j in range(N)]
    C = [A[(j + 1) * M:(j + 2) * M, j * M:(j + 1) * M].toarray() if j + 1 < N else None for j in range(N)]
    Linv = [None] * N; XA = [None] * N; XB = [None] * N
    levels = []
    s = 1
    while s < N:
        elim = list(range(s, N, 2 * s))
        newD = {}
        for j in elim:
            a, b = j - s, j + s
            L = np.linalg.cholesky(D[j])
            if explicit_inverse:
                Li = sl.solve_triangular(L, np.eye(M), lower=True)
                Linv[j] = Li
                XA[j] = Li @ C[a]
                XB[j] = Li @ C[j].T if b < N else None
            else:
                Linv[j] = L
                XA[j] = sl.solve_triangular(L, C[a], lower=True)
                XB[j] = sl.solve_triangular(L, C[j].T, lower=True) if b < N else None
        for j in elim:
            a, b = j - s, j + s
            D[a] = D[a] - XA[j].T @ XA[j]
            if b < N:
                D[b] = D[b] - XB[j].T @ XB[j]
                C[a] = -XB[j].T @ XA[j]
            else:
                C[a] = None
        levels.append((s, elim))
        s *= 2
    L = np.linalg.cholesky(D[0])
    Linv[0] = sl.solve_triangular(L, np.eye(M), lower=True) if explicit_inverse else L
    return dict(M=M, N=N, Linv=Linv, XA=XA, XB=XB, levels=levels, explicit=explicit_inverse)


def bcr_solve(F, rhs):
    M, N = F["M"], F["N"]
    r = rhs.reshape(N, M).copy()
    z = np.zeros_like(r)
    ex = F["explicit"]
    def fs(j, v):
        return F["Linv"][j] @ v if ex else sl.solve_triangular(F["Linv"][j], v, lower=True)
    def bs(j, v):
        return F["Linv"][j].T @ v if ex else sl.solve_triangular(F["Linv"][j].T, v, lower=False)
    for s, elim in F["levels"]:
        for j in elim:
            z[j] = fs(j, r[j])
        for j in elim:
            a, b = j - s, j + s
            r[a] -= F["XA"][j].T @ z[j]
            if b < N:
                r[b] -= F["XB"][j].T @ z[j]
    z[0] = fs(0, r[0])
    p = np.zeros_like(r)
    p[0] = bs(0, z[0])
    for s, elim in reversed(F["levels"]):
        for j in elim:
            a, b = j - s, j + s
            v = z[j] - F["XA"][j] @ p[a]
            if b < N:
                v = v - F["XB"][j] @ p[b]
            p[j] = bs(j, v)
    return p.ravel()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cameraman_128_10"
    alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    ub, f = nt.load_dataset("/root/repo/tests/golden/datasets.npz", name, 1)
    f = f[:1]; ub = ub[:1]
    u = np.asarray(co.pdhg(f, alpha, maxiter=5000))
    O, N, M = f.shape
    amap = np.full((N, M), alpha)
    gp, p_or, res = co.gradient_image(u[0], ub[0], amap)
    print("oracle grad", gp.sum(), "res", res)
    gp1, _, res1 = co.gradient_image(u[0], ub[0], amap, nref=6)
    print("oracle grad nref=6", gp1.sum(), "res", res1)
    A, rhs, h1, h2, apply = assemble(u[0], ub[0], amap)
    for ex in (True, False):
        t = time.time()
        F = bcr_factor(A, M, N, explicit_inverse=ex)
        p = bcr_solve(F, rhs)
        for it in range(3):
            rr = rhs - apply(p)
            p = p + bcr_solve(F, rr)
            print("  sweep", it, "res", np.linalg.norm(rhs - apply(p)) / np.linalg.norm(rhs))
        P = p.reshape(N, M)
        d1 = np.zeros((N, M)); d2 = np.zeros((N, M))
        d1[:, :-1] = P[:, 1:] - P[:, :-1]; d2[:-1, :] = P[1:, :] - P[:-1, :]
        g = -(d1 * h1 + d2 * h2).sum()
        print("explicit" if ex else "trsm", "grad", g, "rel diff vs oracle", abs(g - gp.sum()) / abs(gp.sum()),
              "max|dp|", np.abs(p - np.asarray(p_or).ravel()).max(), "time", time.time() - t)


if __name__ == "__main__":
    main()
