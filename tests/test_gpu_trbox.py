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


def test_patch_bilevel_cameraman_full_budget(gpu_solver_cls, oracle):
    """The reference's whole outer budget (maxiter = 20, tol = 1e-5, /root/reference/src/BPLDenoising.jl:308-312,
    350-357) for the 2x2 patch parameter on cameraman_128_10 with the reference's 5000 inner iterations: the HIP-driven
    and the oracle-driven loop take the same accept/reject decisions (same radius sequence) and learn the same
    parameter.  The L-BFGS operator / CG of the harness are restatements of unpinned packages (trbox.py)."""
    import bpldenoising_amd as B
    ub, f = T.load_dataset(DATASETS_NPZ, "cameraman_128_10")
    x0 = 1e-4 * np.ones((2, 2))
    xg, ug, hg = trbox.bilevel_learn((ub, f), lambda x, ds, d, **kw: B.tv_op_learning_function(x, ds, d), x0, 1e-4,
                                     maxiter=20, tol=1e-5)
    xo, uo, ho = trbox.bilevel_learn((ub, f), _oracle_lf(oracle, 5000), x0, 1e-4, maxiter=20, tol=1e-5)
    assert len(hg) == len(ho) == 20 or hg[-1]["radius_value"] < 1e-5
    assert [h["radius_value"] for h in hg] == [h["radius_value"] for h in ho]
    assert np.allclose(xg, xo, rtol=1e-5, atol=1e-12)
    assert np.allclose([h["function_value"] for h in hg], [h["function_value"] for h in ho], rtol=1e-9)
    assert np.abs(ug - uo).max() < 1e-8


def test_scalar_driver_end_to_end(gpu_solver_cls, oracle, tmp_path):
    """`scalar_bilevel_tv_learn(dataset_name=..., num_samples=...)` with its default (HIP) learning function:
    the same learned parameter and log as when the driver is run on the oracle, and the artefacts exist."""
    import os
    import bpldenoising_amd as B
    kw = dict(npz=DATASETS_NPZ, dataset_name="faces_train_128_10", num_samples=3, maxiter=4, verbose_iter=0,
              lf_kwargs=dict(maxiter=1500))
    xg, ug, lg, wg = B.scalar_bilevel_tv_learn(out_root=str(tmp_path / "hip"), **kw)
    kw["lf_kwargs"] = dict(maxiter=1500, nthreads=8)
    xo, uo, lo, wo = B.scalar_bilevel_tv_learn(learning_function=oracle.tv_op_learning_function,
                                               out_root=str(tmp_path / "cpu"), **kw)
    assert xg == pytest.approx(xo, rel=1e-8) and np.abs(ug - uo).max() < 1e-9
    assert [e["radius_value"] for e in lg] == [e["radius_value"] for e in lo]
    assert os.path.exists(wg["perf"]) and os.path.exists(wg["quality"]) and len(wg["png"]) == 9
    qg, qo = open(wg["quality"]).read().splitlines()[-1].split(), open(wo["quality"]).read().splitlines()[-1].split()
    assert np.allclose([float(v) for v in qg], [float(v) for v in qo], rtol=1e-7)


def test_cost_sweeps_and_validation(gpu_solver_cls, oracle, tmp_path):
    """generate_scalar_tv_cost / generate_2d_tv_cost / validate_tv_parameter (src/BPLDenoising.jl:92-178,381-415)
    through the batched sweep: every cost equals the one-by-one oracle value."""
    import bpldenoising_amd.experiments as E
    ub, f = T.load_dataset(DATASETS_NPZ, "circle_128_10")
    rng1 = np.array([0.02, 0.1, 0.3])
    c1 = E.generate_scalar_tv_cost("circle", rng1, npz=DATASETS_NPZ, out_root=str(tmp_path), maxiter=600)
    ref = [oracle.cost(oracle.pdhg(f[:1], a, maxiter=600), ub[:1]) for a in rng1]
    assert np.allclose(c1, ref, rtol=1e-12)
    z = np.load(str(tmp_path / "circle_128_10" / "circle_128_10_cost.npz"))
    assert np.array_equal(z["parameter_range"], rng1) and np.array_equal(z["costs"], c1)
    c2 = E.generate_2d_tv_cost("circle", [0.05, 0.2], [0.1], npz=DATASETS_NPZ, out_root=str(tmp_path), maxiter=400)
    ref2 = [oracle.cost(oracle.pdhg(f[:1], np.array([[a, 0.1]]), maxiter=400), ub[:1]) for a in (0.05, 0.2)]
    assert c2.shape == (2, 1) and np.allclose(c2[:, 0], ref2, rtol=1e-12)
    u, cost, w = E.validate_tv_parameter(0.1, dataset_name="circle", npz=DATASETS_NPZ, out_root=str(tmp_path), maxiter=500)
    assert np.array_equal(u, oracle.pdhg(f, 0.1, maxiter=500)) and np.isclose(cost, oracle.cost(u, ub), rtol=1e-13)
    # the same drivers over a multi-device handle (rehearsed on one GPU with a repeated device): the ONE image of the set
    # (num_samples = 1, src/BPLDenoising.jl:313) leaves only the parameter axis to split -- replicas, bitwise the same costs
    import bpldenoising_amd.learning_function as LF
    try:
        c1m = E.generate_scalar_tv_cost("circle", rng1, npz=DATASETS_NPZ, out_root=str(tmp_path), maxiter=600, devices=[0, 0])
        assert LF._cache["s"]["solver"].stats()["sweep_shards"] == 2
        assert np.array_equal(c1m, c1)
        c2m = E.generate_2d_tv_cost("circle", [0.05, 0.2], [0.1], npz=DATASETS_NPZ, out_root=str(tmp_path), maxiter=400, devices=[0, 0])
        assert np.array_equal(c2m, c2)
        um, costm, _ = E.validate_tv_parameter(0.1, dataset_name="circle", npz=DATASETS_NPZ, out_root=str(tmp_path), maxiter=500, devices=[0, 0])
        assert np.array_equal(um, u) and costm == cost
    finally:
        LF.use_devices()        # back to one GPU for the tests that follow

