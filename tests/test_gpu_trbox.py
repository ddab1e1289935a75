"""End to end: the outer trust-region loop driven by the HIP learning function learns the same
parameter as when it is driven by the oracle (BASELINE configs 1 and 4, reduced outer budget)."""
import numpy as np
import pytest
from oracle import np_twin as T
from bpldenoising_amd import trbox
from conftest import DATASETS_NPZ

pytestmark = pytest.mark.gpu


def _oracle_lf(oracle, maxiter):
    return lambda x, ds, delta, **kw: oracle.tv_op_learning_function(x, ds, delta, maxiter=maxiter, nthreads=8)


def test_scalar_bilevel_circle(gpu_solver_cls, oracle):
    """scalar_bilevel_tv_learn on circle_128 (/root/reference/src/BPLDenoising.jl:325-343): a0 = 0.1, D0 = 0.1."""
    import bpldenoising_amd as B
    ub, f = T.load_dataset(DATASETS_NPZ, "circle_128_10")
    xg, ug, hg = trbox.bilevel_learn((ub, f), lambda x, ds, d, **kw: B.tv_op_learning_function(x, ds, d, maxiter=5000), 0.1, 0.1, maxiter=8)
    xo, uo, ho = trbox.bilevel_learn((ub, f), _oracle_lf(oracle, 5000), 0.1, 0.1, maxiter=8)
    assert len(hg) == len(ho)
    for a, b in zip(hg, ho):
        assert a["x"] == pytest.approx(b["x"], rel=1e-9)
        assert a["function_value"] == pytest.approx(b["function_value"], rel=1e-12)
        assert a["radius_value"] == b["radius_value"]
    assert xg == pytest.approx(xo, rel=1e-9)
    assert np.array_equal(ug, uo)


def test_patch_bilevel_cameraman(gpu_solver_cls, oracle):
    """patch_bilevel_tv_learn start (/root/reference/src/BPLDenoising.jl:350-376): a0 = 1e-4*ones(2,2), D0 = 1e-4."""
    import bpldenoising_amd as B
    ub, f = T.load_dataset(DATASETS_NPZ, "cameraman_128_10")
    x0 = 1e-4 * np.ones((2, 2))
    xg, ug, hg = trbox.bilevel_learn((ub, f), lambda x, ds, d, **kw: B.tv_op_learning_function(x, ds, d, maxiter=2000), x0, 1e-4, maxiter=4)
    xo, uo, ho = trbox.bilevel_learn((ub, f), _oracle_lf(oracle, 2000), x0, 1e-4, maxiter=4)
    assert np.allclose(xg, xo, rtol=1e-6, atol=1e-12)
    assert [h["radius_value"] for h in hg] == [h["radius_value"] for h in ho]
    assert np.abs(ug - uo).max() < 1e-9
