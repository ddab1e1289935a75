"""Oracle self-checks (CPU): operators, step table, rsqrt -- SURVEY.md 8c (ii)."""
import numpy as np
from oracle import np_twin as T


def test_adjointness(oracle):
    rng = np.random.default_rng(0)
    for (N, M) in [(17, 23), (32, 32), (1, 9), (9, 1)]:
        x = rng.standard_normal((N, M)); y1 = rng.standard_normal((N, M)); y2 = rng.standard_normal((N, M))
        d1, d2 = oracle.grad_fwd(x)
        lhs = np.sum(d1 * y1 + d2 * y2)
        rhs = np.sum(x * oracle.grad_fwd_T(y1, y2))
        assert abs(lhs - rhs) <= 1e-12 * max(1.0, abs(lhs))
        # numpy twin is the same operator
        t1, t2 = T.grad_fwd(x)
        assert np.array_equal(t1, d1) and np.array_equal(t2, d2)
        assert np.allclose(T.grad_fwd_T(y1, y2), oracle.grad_fwd_T(y1, y2), atol=1e-14)


def test_neumann_boundary_and_matrix_form(oracle):
    rng = np.random.default_rng(1)
    N, M = 11, 7
    x = rng.standard_normal((N, M))
    d1, d2 = oracle.grad_fwd(x)
    assert np.all(d1[:, -1] == 0) and np.all(d2[-1, :] == 0)
    G = T.grad_matrix(M, N)   # matrix(op, n): stacked [D1; D2] on the column-major vec
    assert np.array_equal(G @ x.reshape(-1), np.concatenate([d1.reshape(-1), d2.reshape(-1)]))
    assert np.allclose(G.T @ np.concatenate([d1.reshape(-1), d2.reshape(-1)]),
                       oracle.grad_fwd_T(d1, d2).reshape(-1), atol=1e-13)


def test_operator_norm_below_8(oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((40, 40))
    for _ in range(200):
        d1, d2 = oracle.grad_fwd(x)
        x = oracle.grad_fwd_T(d1, d2)
        lam = np.linalg.norm(x)
        x /= lam
    assert 7.0 < lam < 8.0


def test_step_table(oracle):
    tab = oracle.step_table(5000)
    L2 = 8.0
    assert np.allclose(tab[:, 0] * tab[:, 1] * L2, 5 * 0.99 / 5, rtol=1e-12)   # tau*sigma*L^2 = 0.99
    assert np.all(np.diff(tab[:, 0]) < 0) and np.all(np.diff(tab[:, 1]) > 0)
    assert np.allclose(tab[:, 2], 1 / np.sqrt(1 + 2 * tab[:, 0]), rtol=1e-15)
    assert np.allclose(tab[:, :3], T.step_table(5000), rtol=1e-15)
    noacc = oracle.step_table(10, accel=False)
    assert np.all(noacc[:, 2] == 1.0) and np.all(noacc[:, 0] == noacc[0, 0])


def test_rsqrt_nr(oracle):
    rng = np.random.default_rng(3)
    xs = np.concatenate([10.0 ** rng.uniform(-60, 60, 5000), rng.random(5000), [1.0, 4.0, 0.25, 1e-300, 1e300]])
    r = np.array([oracle.rsqrt_nr(x) for x in xs])
    assert np.max(np.abs(r * np.sqrt(xs) - 1.0)) < 5e-16
    assert np.max(np.abs(T.rsqrt_nr(xs) / r - 1.0)) < 1e-15


def test_patch_upsample_and_adjoint(oracle):
    rng = np.random.default_rng(4)
    M, N = 12, 10
    for (n, m) in [(2, 2), (1, 2), (5, 3), (10, 12)]:
        a = rng.random((n, m))
        up = oracle.patch_upsample(a, M, N)
        assert np.array_equal(up, T.patch_upsample(a, M, N))
        g = rng.standard_normal((N, M))
        adj = oracle.patch_adjoint(g, m, n)
        assert np.allclose(adj, T.patch_adjoint(g, m, n), atol=1e-13)
        assert abs(np.sum(up * g) - np.sum(a * adj)) < 1e-12      # <P a, g> = <a, P^T g>
    a = rng.random((2, 2))
    up = oracle.patch_upsample(a, 128, 128)   # 2x2 -> four 64x64 blocks
    assert np.all(up[:64, :64] == a[0, 0]) and np.all(up[:64, 64:] == a[0, 1])
    assert np.all(up[64:, :64] == a[1, 0]) and np.all(up[64:, 64:] == a[1, 1])
