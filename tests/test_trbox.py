"""Outer trust-region harness driven by the oracle: the reference's control
flow and its quirks (/root/reference/src/TRBox.jl:60-76,149-172,192-273)."""
import numpy as np
import pytest
from bpldenoising_amd import trbox
from conftest import synth_batch


def test_scalar_step_rules():
    # large positive gradient: "Newton" step pn = +g/B is uphill and out of bounds, the Cauchy step is
    # clipped to the lower bound max(-D, eps - x)
    assert trbox.dogleg_box(0.1, 346.0, 0.1, 0.1) == pytest.approx(max(-0.1, trbox.EPS - 0.1))
    assert trbox.dogleg_box(0.05, 346.0, 0.1, 0.1) == pytest.approx(trbox.EPS - 0.05)
    # large negative gradient: full step +D
    assert trbox.dogleg_box(0.1, -38.0, 0.1, 0.1) == pytest.approx(0.1)
    # small positive gradient: pn = g/B lies inside the box and is returned as is (no minus sign, TRBox.jl:63)
    assert trbox.dogleg_box(0.1, 0.005, 0.1, 0.1) == pytest.approx(0.05)
    assert trbox.pred(0.1, 0.05, 0.005) < 0            # ... for which the model predicts an increase
    lb, ub = trbox.get_bounds(np.array([[0.1, 1e-20]]), 0.05)
    assert np.allclose(lb, [[-0.05, trbox.EPS - 1e-20]]) and np.all(ub == 0.05)


def test_lbfgs_operator_is_spd_and_secant():
    rng = np.random.default_rng(0)
    B = trbox.LBFGSOperator(6)
    A = rng.standard_normal((6, 6)); A = A @ A.T + 6 * np.eye(6)
    for _ in range(8):
        s = rng.standard_normal(6)
        B.push(s, A @ s)
    s_last, y_last = B.S[-1], B.Y[-1]
    assert np.allclose(B.matvec(s_last), y_last, rtol=1e-10)          # secant equation
    Bm = np.array([B.matvec(e) for e in np.eye(6)]).T
    assert np.allclose(Bm, Bm.T, atol=1e-10) and np.all(np.linalg.eigvalsh(Bm) > 0)
    assert np.allclose(trbox._cg(B, A[:, 0]), np.linalg.solve(Bm, A[:, 0]), rtol=1e-6)


def test_update_guard_is_the_references():
    """TRBox.jl:174-179: the pair is pushed iff y'(B y) > 0 -- with B SPD that is any y != 0, including pairs
    with y's < 0 that a y's test would drop before the operator's own curvature test sees them."""
    B = trbox.LBFGSOperator(2)
    y, s = np.array([1.0, 0.5]), np.array([0.2, 0.1])
    trbox.updateBFGS(B, y, s)
    assert len(B.S) == 1 and np.array_equal(B.S[0], y) and np.array_equal(B.Y[0], s)   # (y, s) order of :176
    trbox.updateBFGS(B, np.zeros(2), s)                  # y'By = 0: not pushed
    assert len(B.S) == 1
    trbox.updateBFGS(B, np.array([1.0, 0.0]), np.array([-1.0, 0.0]))   # guard passes, the operator's y's > 1e-20 test drops it
    assert len(B.S) == 1


def oracle_lf(oracle, maxiter):
    def lf(x, ds, delta, **kw):
        return oracle.tv_op_learning_function(x, ds, delta, maxiter=maxiter, nthreads=4)
    return lf


def test_scalar_loop_with_oracle(oracle):
    ub, f = synth_batch(2, 32, 32, seed=50)
    x, u, hist = trbox.bilevel_learn((ub, f), oracle_lf(oracle, 400), 0.1, 0.1, maxiter=20, tol=1e-5)
    assert 1 <= len(hist) <= 20 and x > 0
    assert hist[-1]["radius_value"] < 1e-5 or len(hist) == 20          # stop rule
    fvals = [h["function_value"] for h in hist]
    f0 = oracle.tv_op_learning_function(0.1, (ub, f), 0.1, maxiter=400)[1]
    assert fvals[-1] <= f0 + 1e-12                                     # accepted steps never increase fx ... unless rho>0 with pred<0
    u2, c2, _ = oracle.tv_op_learning_function(x, (ub, f), 0.1, maxiter=400)
    assert np.array_equal(u, u2) and c2 == fvals[-1]                   # returned u belongs to returned x
    # deterministic
    x2, _, hist2 = trbox.bilevel_learn((ub, f), oracle_lf(oracle, 400), 0.1, 0.1, maxiter=20, tol=1e-5)
    assert x2 == x and hist2 == hist


def test_patch_loop_with_oracle(oracle):
    ub, f = synth_batch(1, 32, 32, seed=51)
    x0 = 1e-4 * np.ones((2, 2))
    x, u, hist = trbox.bilevel_learn((ub, f), oracle_lf(oracle, 300), x0, 1e-4, maxiter=6, tol=1e-9)
    assert x.shape == (2, 2) and np.all(x > 0) and len(hist) >= 1
    f0 = oracle.tv_op_learning_function(x0, (ub, f), 1e-4, maxiter=300)[1]
    assert hist[-1]["function_value"] <= f0
