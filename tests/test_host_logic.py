"""Host-side logic above the C ABI (no GPU): sharding arithmetic, argument conventions, dataset
format, parameter aliases."""
import numpy as np
import pytest
from conftest import DATASETS_NPZ


def test_shard_range():
    from bpldenoising_amd import shard_range
    assert [shard_range(10, 8, r) for r in range(8)] == [(0, 2), (2, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10)]
    assert [shard_range(1, 4, r) for r in range(4)] == [(0, 1), (1, 1), (1, 1), (1, 1)]
    for O in (1, 7, 10, 64):
        for w in (1, 2, 3, 8):
            r = [shard_range(O, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == O
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_alpha_argument_convention():
    from bpldenoising_amd.learning_function import _alpha_arg
    a, am, an, scalar = _alpha_arg(0.1)
    assert (am, an, scalar) == (1, 1, True) and a[0] == 0.1
    a, am, an, scalar = _alpha_arg(np.ones((3, 2)))     # numpy (n, m) == Julia m x n
    assert (am, an, scalar) == (2, 3, False)
    a, am, an, scalar = _alpha_arg(np.ones(4))          # Julia vector
    assert (am, an) == (4, 1)
    with pytest.raises(ValueError):
        _alpha_arg(np.ones((2, 2, 2)))


def test_dataset_loader_matches_reference_format(tmp_path):
    """filelist.txt lines `true.png,noisy.png`, gray/255, Julia [row, col] column major
    (/root/reference/src/Datasets.jl:54-65)."""
    from PIL import Image
    from bpldenoising_amd import testdataset, load_filelist_dataset
    from bpldenoising_amd.datasets import full_datasetname
    assert full_datasetname("circle_128") == "circle_128_10"         # prefix match (:27-31)
    assert full_datasetname("cameraman") == "cameraman_128_5"        # first match in list order
    with pytest.raises(ValueError):
        full_datasetname("lena")
    t, d = testdataset("faces_train", npz=DATASETS_NPZ)
    assert t.shape == d.shape == (10, 128, 128) and t.dtype == np.float64
    z = np.load(DATASETS_NPZ)
    assert np.array_equal(t[3], z["faces_train_128_10/true"][3].T / 255.0)
    # round trip through real PNG files, including a 1-bit image and no trailing newline
    dd = tmp_path / "circle_128_10"
    dd.mkdir()
    a = (np.arange(12 * 9).reshape(12, 9) % 255).astype(np.uint8)
    b = (a > 100)
    Image.fromarray(a).save(dd / "n.png")
    Image.fromarray(b).save(dd / "t.png")
    (dd / "filelist.txt").write_text("t.png,n.png")
    tt, nn = load_filelist_dataset(str(dd))
    assert tt.shape == (1, 9, 12)
    assert np.array_equal(nn[0], a.T / 255.0) and set(np.unique(tt)) <= {0.0, 1.0}
    t2, n2 = testdataset("circle", root=str(tmp_path))
    assert np.array_equal(n2, nn)


def test_parameter_aliases_need_library():
    """Reference NamedTuple keys (unicode) map onto bpltv_params; unknown keys are rejected."""
    from bpldenoising_amd import _lib
    from bpldenoising_amd.learning_function import TVSolver, _PARAM_ALIASES
    assert _PARAM_ALIASES["τ₀"] == "tau0" and _PARAM_ALIASES["σ₀"] == "sigma0" and _PARAM_ALIASES["ρ"] == "rho"
    s = TVSolver.__new__(TVSolver)          # no handle: only exercise params()
    s._lib = _lib.load()
    p = s.params(maxiter=10000, **{"τ₀": 4.0, "verbose_iter": 10001, "save_results": False})
    assert p.maxiter == 10000 and p.tau0 == 4.0 and p.sigma0 == 0.99 / 5
    with pytest.raises(TypeError):
        s.params(bogus=1)


class FakeSolver:
    """Stands in for the HIP solver in host-logic tests: per-shard partials from the oracle."""

    def __init__(self, M, N, O):
        self.M, self.N, self.O = M, N, O

    def set_data(self, ubar, f):
        self.ubar, self.f = np.array(ubar), np.array(f)

    def evaluate_partial(self, x, delta, fetch_u=True, **kw):
        from oracle import c_oracle as co
        u, c, g = co.tv_op_learning_function(x, (self.ubar, self.f), delta, maxiter=kw.get("maxiter", 5000))
        self._rows = []
        for k in range(self.O):   # per-image rows, as bpltv_per_image returns them
            _, ck, gk = co.tv_op_learning_function(x, (self.ubar[k:k + 1], self.f[k:k + 1]), delta, maxiter=kw.get("maxiter", 5000))
            self._rows.append(np.concatenate([[ck], np.atleast_1d(np.asarray(gk)).reshape(-1)]))
        return (u if fetch_u else None), np.concatenate([[c], np.atleast_1d(np.asarray(g)).reshape(-1)])

    def per_image(self):
        return np.array(self._rows)


def test_sharded_learning_function_single_process(oracle):
    from bpldenoising_amd import ShardedLearningFunction
    from conftest import synth_batch
    ub, f = synth_batch(3, 20, 16, seed=21)
    fn = ShardedLearningFunction((ub, f), solver_factory=FakeSolver)
    u, cost, grad = fn(0.1, 0.1, maxiter=200)
    u0, c0, g0 = oracle.tv_op_learning_function(0.1, (ub, f), 0.1, maxiter=200)
    assert np.array_equal(u, u0) and cost == c0 and grad == g0
    P = np.array([[0.05, 0.1], [0.2, 0.08]])
    u, cost, grad = fn(P, 0.1, maxiter=200)
    u0, c0, g0 = oracle.tv_op_learning_function(P, (ub, f), 0.1, maxiter=200)
    assert grad.shape == (2, 2) and np.allclose(grad, g0, rtol=1e-13) and np.isclose(cost, c0, rtol=1e-15)
