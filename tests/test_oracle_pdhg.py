"""Oracle PDHG: C restatement vs numpy twin, closed-form cases, duality-gap certificate
(SURVEY.md 8c (i), (iii))."""
import numpy as np
import pytest
from oracle import np_twin as T
from conftest import synth_batch


@pytest.mark.parametrize("alpha", [0.1, np.array([[0.05, 0.1], [0.2, 0.08]]), "map"])
def test_c_matches_numpy_twin(oracle, alpha):
    ub, f = synth_batch(2, 30, 22, seed=5)
    if isinstance(alpha, str):
        alpha = 0.05 + 0.1 * np.random.default_rng(0).random((30, 22))
    u = oracle.pdhg(f, alpha, maxiter=400)
    ut = T.pdhg_denoise(f, alpha, maxiter=400)
    assert np.abs(u - ut).max() < 1e-12


def test_rho_and_no_accel(oracle):
    ub, f = synth_batch(1, 20, 20, seed=6)
    u = oracle.pdhg(f, 0.1, maxiter=300, rho=0.05, accel=False)
    ut = T.pdhg_denoise(f, 0.1, maxiter=300, rho=0.05, accel=False)
    assert np.abs(u - ut).max() < 1e-12


def test_closed_form_cases(oracle):
    ub, f = synth_batch(1, 16, 16, seed=7)
    assert np.abs(oracle.pdhg(f, 0.0, maxiter=50) - f).max() < 1e-15   # alpha = 0 => u = f (to rounding)
    c = np.full((1, 16, 16), 0.37)
    assert np.abs(oracle.pdhg(c, 0.2, maxiter=200) - c).max() < 1e-15  # constant f => u = f
    u = oracle.pdhg(f, 50.0, maxiter=4000)                             # alpha >= alpha_max => mean(f)
    assert np.abs(u - f.mean()).max() < 1e-6
    assert np.array_equal(oracle.pdhg(f, 0.1, maxiter=0), f)           # maxiter = 0


def test_transposition_equivariance(oracle):
    ub, f = synth_batch(1, 24, 18, seed=8)
    u = oracle.pdhg(f, 0.1, maxiter=300)
    ut = oracle.pdhg(np.ascontiguousarray(f.transpose(0, 2, 1)), 0.1, maxiter=300)
    assert np.abs(u - ut.transpose(0, 2, 1)).max() < 1e-12


def test_gap_certificate(oracle):
    ub, f = synth_batch(1, 32, 32, seed=9)
    alpha = 0.1
    ustar = oracle.pdhg(f, alpha, maxiter=40000)
    prev = None
    for it in (50, 200, 1000, 5000):
        u, y1, y2 = oracle.pdhg(f, alpha, maxiter=it, return_dual=True)
        assert np.max(np.sqrt(y1 ** 2 + y2 ** 2)) <= alpha * (1 + 1e-15)     # dual feasible
        gap = oracle.gap(u, y1, y2, f, alpha)[0]
        assert gap >= -1e-12
        assert 0.5 * np.sum((u - ustar) ** 2) <= gap + 1e-9                # strong convexity bound
        assert np.isclose(gap, T.rof_gap(u, y1, y2, f, alpha)[0], rtol=1e-6, atol=1e-12)
        if prev is not None:
            assert gap < prev
        prev = gap
    assert prev < 1e-5


def test_cost(oracle):
    ub, f = synth_batch(3, 10, 12, seed=10)
    tot, per = oracle.cost(f, ub, per_image=True)
    assert np.isclose(tot, T.l2_cost(f, ub), rtol=1e-14)
    assert np.isclose(per.sum(), tot, rtol=1e-15)


def test_f32_twin_tracks_the_f64_recurrence(oracle):
    """"spec v2f" (the checker of the library's opt-in dtype = 32 mode) stays within single-precision distance of
    the Float64 recurrence; alpha = 0 leaves f to rounding."""
    rng = np.random.default_rng(3)
    f = rng.random((2, 24, 20))
    a64 = oracle.pdhg(f, 0.1, maxiter=400)
    a32 = oracle.pdhg_f32(f, 0.1, maxiter=400)
    assert 0 < np.abs(a64 - a32).max() < 1e-5
    assert np.abs(oracle.pdhg_f32(f, 0.0, maxiter=20) - f).max() < 1e-6    # (x + tau f) / (1 + tau) in float
    amap = 0.05 + 0.1 * rng.random((24, 20))
    assert np.abs(oracle.pdhg(f, amap, maxiter=200) - oracle.pdhg_f32(f, amap, maxiter=200)).max() < 1e-5


def test_pdhg_opts_is_the_oracle_recurrence_with_the_unpinned_choices(oracle):
    """bplo_pdhg_opts (checker of bpltv_params.init / order / opnorm): defaults == bplo_pdhg bit for bit; each choice
    flipped agrees with the unfused study code bplo_pdhg_variant to rounding."""
    from conftest import synth_batch
    ub, f = synth_batch(2, 40, 36, seed=5)
    alpha = np.array([[0.05, 0.1], [0.2, 0.08]])
    assert np.array_equal(oracle.pdhg_opts(f, alpha, maxiter=120), oracle.pdhg(f, alpha, maxiter=120))
    for init, order, flags in ((1, 0, 1), (0, 1, 2), (1, 1, 3)):
        a = oracle.pdhg_opts(f, alpha, maxiter=120, init=init, order=order)
        b = oracle.pdhg_variant(f, alpha, maxiter=120, flags=flags)
        assert np.abs(a - b).max() < 1e-12
        assert np.abs(a - oracle.pdhg(f, alpha, maxiter=120)).max() > 1e-9      # and it is a different sequence
    L = 2 * np.sqrt(2) * (1 - 1 / 36)
    assert np.abs(oracle.pdhg_opts(f, alpha, maxiter=120, L=L) - oracle.pdhg_variant(f, alpha, maxiter=120, L=L)).max() < 1e-12
