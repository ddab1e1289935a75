"""The oracle against the committed golden vectors (tests/golden/golden_v2.npz, produced by
tests/golden/make_golden.py from the reference's own images)."""
import zlib
import numpy as np
import pytest
from oracle import np_twin as T
from conftest import DATASETS_NPZ

FAST = ["circle_scalar", "cameraman10_scalar", "cameraman10_patch22", "cameraman10_short", "faces_val_patch22"]


@pytest.mark.parametrize("name", FAST)
def test_oracle_reproduces_golden(oracle, golden, name):
    z, meta = golden
    m = meta[name]
    ub, f = T.load_dataset(DATASETS_NPZ, m["dataset"])
    ub, f = ub[m["lo"]:m["hi"]], f[m["lo"]:m["hi"]]
    alpha = np.asarray(m["alpha"]) if isinstance(m["alpha"], list) else m["alpha"]
    u, y1, y2 = oracle.pdhg(f, alpha, maxiter=m["maxiter"], return_dual=True, nthreads=4)
    assert zlib.crc32(np.ascontiguousarray(u).tobytes()) == int(z[name + "/u_crc32"])   # bit exact
    if name + "/u" in z:
        assert np.array_equal(u, z[name + "/u"])
    assert np.isclose(oracle.cost(u, ub), float(z[name + "/cost"]), rtol=1e-14)
    assert np.allclose(oracle.gap(u, y1, y2, f, alpha), z[name + "/gap"], rtol=1e-9, atol=1e-13)
    g = oracle.gradient(alpha, u, ub)
    assert np.allclose(g, z[name + "/grad"], rtol=2e-6)
    assert np.allclose(oracle.gradient(alpha, u, ub, reg=True), z[name + "/grad_reg"], rtol=1e-7)


def test_golden_metadata(golden):
    z, meta = golden
    for name, m in meta.items():
        assert m["twin_max_du"] < 1e-13              # numpy twin agreed with the C oracle
        assert m["c_oracle_vs_literal_grad_rel"] < 2e-6


def test_dataset_statistics(datasets_npz):
    """0.5||f - ubar||^2 of the shipped pairs."""
    ub, f = T.load_dataset(datasets_npz, "cameraman_128_10")
    assert np.isclose(T.l2_cost(f, ub), 114.876, atol=1e-3)
    ub, f = T.load_dataset(datasets_npz, "cameraman_128_5")
    assert np.isclose(T.l2_cost(f, ub), 39.5, atol=0.1)
    ub, f = T.load_dataset(datasets_npz, "circle_128_10")
    assert np.isclose(T.l2_cost(f, ub), 40.3, atol=0.1)
    assert set(np.unique(ub)) == {0.0, 1.0}
    ub, f = T.load_dataset(datasets_npz, "faces_train_128_10")
    assert ub.shape == (10, 128, 128) and f.min() == 0.0 and f.max() == 1.0
