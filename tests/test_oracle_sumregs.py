"""The sum-of-regularisers oracle (oracle/sumregs_oracle.c) pinned by what can be pinned without the reference
(absent package, no fixtures -- PARITY UNPINNED): operator adjointness, the numpy twin of the recurrence, the duality
gap certificate, the LITERAL scipy assembly of the reference's adjoint systems
(/root/reference/src/SumRegsLearningFunction.jl:112-167, :195-262, :264-327, :330-407), finite differences, and the
committed golden vectors on the reference's images."""
import json
import os
import zlib
import numpy as np
import pytest
from conftest import DATASETS_NPZ, GOLDEN, synth_batch
from oracle import np_twin as T
from oracle import np_twin_sumregs as S

A3 = np.array([0.03, 0.02, 0.05])
P3 = np.stack([np.array([[0.03, 0.05], [0.02, 0.04]]), np.array([[0.02, 0.03], [0.05, 0.02]]),
               np.array([[0.04, 0.02], [0.03, 0.06]])])


def test_operators_adjoint_and_norm(oracle):
    rng = np.random.default_rng(0)
    N, M = 13, 9
    for k in range(3):
        x = rng.standard_normal((N, M)); y1 = rng.standard_normal((N, M)); y2 = rng.standard_normal((N, M))
        d1, d2 = oracle.sr_grad(k, x)
        G = S.grad_matrix(k, M, N)
        assert np.allclose(np.concatenate([d1.ravel(), d2.ravel()]), G @ x.ravel(), atol=1e-14)      # C == scipy matrix
        if k == 0:
            y1[:, -1] = 0; y2[-1, :] = 0          # the gather form of G^T assumes y = 0 where G u = 0 structurally
        if k == 1:
            y1[:, 0] = 0; y2[0, :] = 0
        gt = oracle.sr_gradT(k, y1, y2)
        assert np.allclose(gt.ravel(), G.T @ np.concatenate([y1.ravel(), y2.ravel()]), atol=1e-13)
        assert abs(np.sum(d1 * y1 + d2 * y2) - np.sum(x * gt)) < 1e-12
    K = np.vstack([S.grad_matrix(k, 16, 16).toarray() for k in range(3)])
    assert np.linalg.norm(K, 2) ** 2 < 18.0                                  # the step-size bound L^2 = 18
    assert np.abs(oracle.sr_grad(2, np.ones((5, 7)))[0]).max() == 0          # constants are in every null space


@pytest.mark.parametrize("alpha", [A3, P3], ids=["vector", "patch"])
def test_pdhg_twin_gap_and_limits(oracle, alpha):
    ub, f = synth_batch(2, 24, 20, seed=3)
    u, y = oracle.sumregs_pdhg(f, alpha, maxiter=400, return_dual=True)
    assert np.abs(u - S.pdhg(f, alpha, maxiter=400)).max() < 1e-13
    g400 = oracle.sumregs_gap(u, y, f, alpha)
    u2, y2 = oracle.sumregs_pdhg(f, alpha, maxiter=3000, return_dual=True)
    g3000 = oracle.sumregs_gap(u2, y2, f, alpha)
    assert np.all(g400 > 0) and np.all(g3000 >= -1e-12) and np.all(g3000 < g400)
    assert np.abs(oracle.sumregs_pdhg(f, np.zeros(3), maxiter=50) - f).max() < 1e-15           # alpha = 0: u = f
    big = oracle.sumregs_pdhg(f, np.array([50.0, 50.0, 50.0]), maxiter=20000)
    assert np.abs(big - f.mean(axis=(1, 2), keepdims=True)).max() < 1e-10                      # alpha large: u = mean(f)
    # with only the forward term the model is ROF; its minimiser is unique, so a (different) TV solve agrees within the gaps
    a1 = np.array([0.07, 0.0, 0.0])
    us, ys = oracle.sumregs_pdhg(f, a1, maxiter=6000, return_dual=True)
    ut, t1, t2 = oracle.pdhg(f, 0.07, maxiter=6000, return_dual=True)
    bound = np.sqrt(2 * oracle.sumregs_gap(us, ys, f, a1)) + np.sqrt(2 * oracle.gap(ut, t1, t2, f, 0.07))
    assert np.all(np.sqrt(((us - ut) ** 2).sum(axis=(1, 2))) <= bound + 1e-12)


@pytest.mark.parametrize("alpha", [A3, P3], ids=["vector", "patch"])
def test_gradients_match_the_literal_reference_systems(oracle, alpha):
    ub, f = synth_batch(2, 24, 20, seed=3)
    u = oracle.sumregs_pdhg(f, alpha, maxiter=3000)
    for reg in (False, True):
        g = oracle.sumregs_gradient(alpha, u, ub, reg=reg)
        gl = S.batch_gradient(alpha, u, ub, reg=reg)
        assert np.shape(g) == np.shape(gl) == np.shape(alpha)
        assert np.abs(g - gl).max() <= 5e-8 * np.abs(gl).max()


def test_gradient_against_finite_differences(oracle):
    ub, f = synth_batch(1, 20, 20, seed=8)
    a = np.array([0.04, 0.03, 0.05])
    J = lambda x: oracle.cost(oracle.sumregs_pdhg(f, x, maxiter=30000), ub)
    g = oracle.sumregs_gradient(a, oracle.sumregs_pdhg(f, a, maxiter=30000), ub)
    for k in range(3):
        e = np.zeros(3); e[k] = 2e-4
        fd = (J(a + e) - J(a - e)) / 4e-4
        assert abs(fd - g[k]) <= 0.03 * abs(fd) + 1e-3, (k, fd, g[k])       # the solution map is only piecewise smooth


def test_golden_vectors_on_reference_images(oracle):
    z = np.load(os.path.join(GOLDEN, "golden_sumregs.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    for m in meta[:3]:                                           # the cameraman cases (one image each)
        ub, f = T.load_dataset(DATASETS_NPZ, m["dataset"])
        ub, f = ub[m["lo"]:m["hi"]], f[m["lo"]:m["hi"]]
        alpha = np.asarray(m["alpha"])
        u = oracle.sumregs_pdhg(f, alpha, maxiter=m["maxiter"], nthreads=4)
        assert zlib.crc32(np.ascontiguousarray(u).tobytes()) == int(z[m["name"] + "/u_crc32"])
        assert np.isclose(oracle.cost(u, ub), float(z[m["name"] + "/cost"]), rtol=1e-13)
        g = oracle.sumregs_gradient(alpha, u, ub)
        assert np.abs(g - z[m["name"] + "/grad"]).max() <= 2e-7 * np.abs(z[m["name"] + "/grad"]).max()
        gr = oracle.sumregs_gradient(alpha, u, ub, reg=True)
        assert np.abs(gr - z[m["name"] + "/grad_reg"]).max() <= 1e-8 * np.abs(z[m["name"] + "/grad_reg"]).max()
