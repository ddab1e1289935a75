"""GPU parity of the evaluate surface: loss, adjoint gradients (both branches, scalar / patch /
pixelwise alpha), the sharded partial forms, error behaviour and the reference-named entry points.

Gradient tolerance: the reference's saddle system is so ill conditioned that its own plain LU is
reproducible only to ~1e-3 (tests/test_oracle_gradient.py).  Stated bar here: HIP vs the C oracle
(same reduced system, same u bit for bit) rtol 1e-6; HIP vs the golden vectors (exact solution of
the reference's literal system) rtol 5e-6.
"""
import numpy as np
import pytest
from oracle import np_twin as T
from conftest import DATASETS_NPZ, synth_batch

pytestmark = pytest.mark.gpu

P22 = np.array([[0.08, 0.12], [0.1, 0.05]])


@pytest.mark.parametrize("alpha", [0.1, P22, np.array([[0.06, 0.15]])], ids=["scalar", "patch22", "patch21"])
def test_evaluate_matches_oracle(gpu_solver_cls, oracle, alpha):
    ub, f = synth_batch(3, 64, 48, seed=20)
    s = gpu_solver_cls(48, 64, 3)
    s.set_data(ub, f)
    u, cost, grad = s.evaluate(alpha, 0.1, maxiter=800)
    u0 = oracle.pdhg(f, alpha, maxiter=800)
    assert np.array_equal(u, u0)
    assert np.isclose(cost, oracle.cost(u0, ub), rtol=1e-13)
    g0 = oracle.gradient(alpha, u0, ub)
    assert np.shape(grad) == np.shape(g0)
    assert np.allclose(grad, g0, rtol=1e-6, atol=1e-10)
    st = s.stats()
    assert st["reg_gradient_used"] == 0 and st["adjoint_residual"] <= 1e-8, st
    eps = np.finfo(float).eps
    assert st["adjoint_attempts"] == 1 and st["kappa_used"] == (1e14 if np.ndim(alpha) == 0 else 1 / np.sqrt(eps))
    # Delta <= Delta_t takes the gradient_reg branch (TVLearningFunctionVec.jl:21-25)
    u, cost, greg = s.evaluate(alpha, 1e-7, maxiter=800)
    assert s.stats()["reg_gradient_used"] == 1
    assert np.allclose(greg, oracle.gradient(alpha, u0, ub, reg=True), rtol=1e-8, atol=1e-12)
    s.close()


@pytest.mark.parametrize("name,ds,lo,hi", [
    ("cameraman10_scalar", "cameraman_128_10", 0, 1),
    ("cameraman10_patch22", "cameraman_128_10", 0, 1),
    ("faces_train_scalar", "faces_train_128_10", 0, 10),
    ("faces_val_patch22", "faces_val_128_10", 0, 3),
])
def test_evaluate_matches_golden(gpu_solver_cls, golden, name, ds, lo, hi):
    """BASELINE configs 2-4 end to end on the reference's images against the golden vectors."""
    z, meta = golden
    m = meta[name]
    ub, f = T.load_dataset(DATASETS_NPZ, ds)
    ub, f = ub[lo:hi], f[lo:hi]
    alpha = np.asarray(m["alpha"]) if isinstance(m["alpha"], list) else m["alpha"]
    s = gpu_solver_cls(128, 128, hi - lo)
    s.set_data(ub, f)
    u, cost, grad = s.evaluate(alpha, 0.1, maxiter=m["maxiter"])
    assert np.isclose(cost, float(z[name + "/cost"]), rtol=1e-13)
    assert np.allclose(grad, z[name + "/grad"], rtol=5e-6)
    u, cost, greg = s.evaluate(alpha, 0.0, maxiter=m["maxiter"], fetch_u=False)
    assert u is None
    assert np.allclose(greg, z[name + "/grad_reg"], rtol=1e-7)
    s.close()


def test_pixelwise_alpha_gradient(gpu_solver_cls, oracle):
    ub, f = synth_batch(2, 40, 32, seed=21)
    amap = 0.05 + 0.1 * np.random.default_rng(3).random((40, 32))
    s = gpu_solver_cls(32, 40, 2)
    s.set_data(ub, f)
    u, cost, grad = s.evaluate(amap, 0.1, maxiter=600)
    u0 = oracle.pdhg(f, amap, maxiter=600)
    g0 = oracle.gradient(amap, u0, ub)
    assert grad.shape == (40, 32) and np.array_equal(u, u0)
    assert np.allclose(grad, g0, rtol=1e-5, atol=1e-8 * np.abs(g0).max())
    s.close()


def test_standalone_gradient_entry(gpu_solver_cls, oracle):
    ub, f = synth_batch(2, 48, 48, seed=22)
    u0 = oracle.pdhg(f, 0.1, maxiter=500)
    s = gpu_solver_cls(48, 48, 2)
    g = s.gradient(u0, ub, 0.1)
    assert np.isclose(g, oracle.gradient(0.1, u0, ub), rtol=1e-6)
    g = s.gradient(u0, ub, P22, reg=True)
    assert np.allclose(g, oracle.gradient(P22, u0, ub, reg=True), rtol=1e-8)
    s.close()


def test_partial_forms_sum_to_the_batch(gpu_solver_cls):
    """Shards: [cost, grad] partials of two handles add up to the one-handle result; the device
    form writes the same vector into caller-owned HBM."""
    import torch
    ub, f = synth_batch(5, 64, 64, seed=23)
    full = gpu_solver_cls(64, 64, 5)
    full.set_data(ub, f)
    _, c, g = full.evaluate(P22, 0.1, maxiter=300)
    a = gpu_solver_cls(64, 64, 3); a.set_data(ub[:3], f[:3])
    b = gpu_solver_cls(64, 64, 2); b.set_data(ub[3:], f[3:])
    ua, pa = a.evaluate_partial(P22, 0.1, maxiter=300)
    ub_, pb = b.evaluate_partial(P22, 0.1, maxiter=300)
    tot = pa + pb
    assert np.isclose(tot[0], c, rtol=1e-13) and np.allclose(tot[1:].reshape(2, 2), g, rtol=1e-12)
    t = torch.zeros(5, dtype=torch.float64, device="cuda")
    a.evaluate_device(P22, 0.1, t.data_ptr(), maxiter=300)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), pa)
    ud = torch.empty((3, 64, 64), dtype=torch.float64, device="cuda")
    a.copy_u_device(ud.data_ptr())
    assert np.array_equal(ud.cpu().numpy(), ua)
    # dataset handed over from HBM
    c2 = gpu_solver_cls(64, 64, 3)
    tu, tf = torch.from_numpy(ub[:3]).cuda(), torch.from_numpy(f[:3]).cuda()
    c2.set_data_device(tu.data_ptr(), tf.data_ptr())
    _, pc = c2.evaluate_partial(P22, 0.1, maxiter=300)
    assert np.array_equal(pc, pa)
    for s in (full, a, b, c2):
        s.close()


def test_error_behaviour(gpu_solver_cls):
    from bpldenoising_amd._lib import BpltvError
    s = gpu_solver_cls(32, 32, 1)
    with pytest.raises(BpltvError) as e:
        s.denoise(0.1, maxiter=10)                     # no data yet
    assert e.value.code == 3
    ub, f = synth_batch(1, 32, 32, seed=24)
    s.set_data(ub, f)
    with pytest.raises(BpltvError) as e:
        s.denoise(np.ones((40, 2)), maxiter=10)        # parameter larger than the image
    assert e.value.code == 1
    with pytest.raises(BpltvError):
        s.denoise(0.1, maxiter=-1)
    with pytest.raises(ValueError):
        s.set_data(ub[:, :16], f)
    assert s.denoise(0.1, maxiter=10).shape == (1, 32, 32)   # handle still usable after errors
    s.close()


def test_breakdown_retry_is_visible_and_arguments_are_validated(gpu_solver_cls):
    """The literal 1/eps() = 4.5e15 active-set weight (kappa_cap lifted) breaks the Cholesky down on converged
    images; the retry with a 100x smaller weight succeeds and bpltv_stats says so (kappa_used, adjoint_attempts)
    instead of returning a gradient of an unnamed system.  Also: NaN / negative parameters, a zero entry where
    the arithmetic divides by alpha, and the experiment switch of tools/ builds are BPLTV_E_ARG."""
    from bpldenoising_amd._lib import BpltvError
    from oracle import np_twin as T2
    ub, f = T2.load_dataset(DATASETS_NPZ, "faces_train_128_10")
    s = gpu_solver_cls(128, 128, 10)
    s.set_data(ub, f)
    _, _, g = s.evaluate(0.1, 0.1)
    st = s.stats()
    assert st["adjoint_attempts"] == 1 and st["kappa_used"] == 1e14 and st["adjoint_residual"] <= 1e-8
    _, _, g2 = s.evaluate(0.1, 0.1, kappa_cap=1e300)
    st = s.stats()
    eps = np.finfo(float).eps
    assert st["adjoint_attempts"] == 2 and np.isclose(st["kappa_used"], 1e-2 / eps) and st["adjoint_residual"] <= 1e-8
    assert np.isclose(g2, g, rtol=1e-6)                       # both weights sit in the hard-constraint limit
    for bad in (np.nan, -0.1, np.array([[0.1, -1e-3], [0.1, 0.1]]), np.array([[np.inf, 0.1]])):
        with pytest.raises(BpltvError) as e:
            s.denoise(bad, maxiter=10)
        assert e.value.code == 1
    with pytest.raises(BpltvError) as e:
        s.evaluate(np.array([[0.1, 0.0], [0.1, 0.1]]), 0.0, maxiter=50)     # gradient_reg, patch: sqrt(alpha) scaling
    assert e.value.code == 1 and "alpha" in str(e.value)
    with pytest.raises(BpltvError) as e:
        s.denoise(np.array([[0.1, 0.0]]), maxiter=10, rho=0.01)             # rho != 0 divides by alpha
    assert e.value.code == 1
    assert np.abs(s.denoise(0.0, maxiter=10) - f).max() < 1e-15             # alpha = 0 itself is fine: u = f
    p = s.params(maxiter=10)
    p.reserved[3] = 4
    import ctypes as C
    a = np.array([0.1])
    rc = s._lib.bpltv_denoise(s._h, a.ctypes.data_as(C.POINTER(C.c_double)), 1, 1, C.byref(p), None)
    assert rc == 1 and b"reserved[3]" in s._lib.bpltv_last_error(s._h)
    s.close()


@pytest.mark.parametrize("shape,alpha", [((2, 40, 200), 0.1), ((1, 70, 150), np.array([[0.06, 0.15], [0.1, 0.2]])),
                                         ((1, 36, 160), "map"), ((1, 45, 139), 0.08), ((3, 20, 257), "map")],
                         ids=["scalar200", "patch150", "map160", "scalar139_odd_band", "map257_three_images"])
@pytest.mark.parametrize("method", ["nd", "band"])
def test_wide_images_use_the_hbm_band_path(gpu_solver_cls, oracle, shape, alpha, method):
    """M > 138 does not fit the LDS window.  Default: the nested-dissection (multifrontal) Cholesky (O(M^3) flop);
    cross-check: the band factored in place in HBM (O(M^4), `adjoint_method="band"`).  Same reduced system, same
    tolerance against the oracle."""
    O, N, M = shape
    ub, f = synth_batch(O, N, M, seed=70 + M)
    if isinstance(alpha, str):
        alpha = 0.05 + 0.1 * np.random.default_rng(4).random((N, M))
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    kw = {} if method == "nd" else {"adjoint_method": "band"}       # nd is the automatic choice for this shape
    u, cost, grad = s.evaluate(alpha, 0.1, maxiter=500, **kw)
    st = s.stats()
    assert st["adjoint_method"] == ("nd" if method == "nd" else "band-hbm") and st["adjoint_residual"] <= 1e-8, st
    u0 = oracle.pdhg(f, alpha, maxiter=500)
    g0 = oracle.gradient(alpha, u0, ub)
    assert np.array_equal(u, u0)
    if np.ndim(g0) == 2 and g0.shape == (N, M):
        assert np.allclose(grad, g0, rtol=1e-4, atol=2e-6 * np.abs(g0).max()) and np.isclose(grad.sum(), g0.sum(), rtol=1e-6)
    else:
        assert np.allclose(grad, g0, rtol=2e-6, atol=1e-9)
    _, _, greg = s.evaluate(alpha, 0.0, maxiter=500, **kw)
    gr0 = oracle.gradient(alpha, u0, ub, reg=True)
    assert np.allclose(greg, gr0, rtol=1e-6, atol=1e-8 * np.abs(gr0).max())
    s.close()


def test_reference_named_entry_points(oracle):
    """tv_op_learning_function / denoise / TVDenoise with the reference's call shapes."""
    import bpldenoising_amd as B
    ub, f = T.load_dataset(DATASETS_NPZ, "circle_128_10")
    u, cost, grad = B.tv_op_learning_function(0.1, (ub, f), 0.1, maxiter=400)
    u0 = oracle.pdhg(f, 0.1, maxiter=400)
    assert np.array_equal(u, u0) and isinstance(grad, float)
    assert np.isclose(cost, B.L2CostFunction(u, ub), rtol=1e-13)
    x = 1e-2 * np.ones((2, 2))
    u, cost, grad = B.tv_op_learning_function(x, (ub, f), 0.1, Δt=1e-6, maxiter=400)
    assert grad.shape == x.shape                                       # grad has the shape of x
    assert np.array_equal(B.denoise(f, x, B.FwdGradientOp(), maxiter=400), u)
    ud = B.TVDenoise(f, 0.1, maxiter=400)
    assert np.array_equal(ud, u0)
    with pytest.raises(TypeError):
        B.denoise(f, 0.1, op=object())


def test_parameter_sweep_matches_individual_solves(gpu_solver_cls, oracle):
    """generate_cost / generate_2d_cost (/root/reference/src/BPLDenoising.jl:92-111,136-158) as one
    batched solve: every (parameter, image) problem is bit-identical to its stand-alone solve."""
    ub, f = synth_batch(2, 64, 48, seed=26)
    s = gpu_solver_cls(48, 64, 2)
    s.set_data(ub, f)
    alphas = np.array([0.0, 0.02, 0.05, 0.1, 0.2, 0.4, 0.8])
    costs, u = s.sweep(alphas, fetch_u=True, maxiter=400)
    assert costs.shape == (7,) and u.shape == (7, 2, 64, 48)
    for k, a in enumerate(alphas):
        uk = s.denoise(float(a), maxiter=400)
        assert np.array_equal(u[k], uk)
        assert np.isclose(costs[k], oracle.cost(uk, ub), rtol=1e-13)
    assert np.array_equal(u[3], oracle.pdhg(f, 0.1, maxiter=400))
    # 2-D sweep with a 2x1 parameter [a; b] (generate_2d_cost)
    grid = np.array([[[a, b]] for a in (0.05, 0.1) for b in (0.02, 0.2, 0.3)])      # (K, n=1, m=2)
    costs2, u2 = s.sweep(grid, fetch_u=True, maxiter=300)
    for k in range(grid.shape[0]):
        assert np.array_equal(u2[k], s.denoise(grid[k], maxiter=300))
    # a default-context solve afterwards is unaffected
    assert np.array_equal(s.denoise(0.1, maxiter=400), u[3])
    s.close()


def test_generate_cost_entry_point(oracle):
    import bpldenoising_amd as B
    ub, f = T.load_dataset(DATASETS_NPZ, "cameraman_128_5")
    par = np.linspace(0.0, 0.055, 12)          # minimum of the curve near alpha = 0.015-0.02
    costs = B.generate_cost((ub, f), par, maxiter=1000)
    k = int(np.argmin(costs))
    assert 0 < k < 11                                   # the cost curve has an interior minimum
    assert np.isclose(costs[5], oracle.cost(oracle.pdhg(f, par[5], maxiter=1000), ub), rtol=1e-13)


def test_sharded_learning_function_rccl_single_rank(gpu_solver_cls, oracle):
    """The N-rank host path with the RCCL backend on one rank: partial vector written to HBM by
    bpltv_evaluate_device, all-reduced by torch.distributed (nccl == RCCL), u fetched from HBM."""
    import os
    import torch.distributed as dist
    from bpldenoising_amd import ShardedLearningFunction
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ub, f = synth_batch(3, 48, 48, seed=27)
        fn = ShardedLearningFunction((ub, f))
        assert fn.device_reduce and (fn.lo, fn.hi) == (0, 3)
        u, cost, grad = fn(P22, 0.1, maxiter=300)
        u0 = oracle.pdhg(f, P22, maxiter=300)
        assert np.array_equal(u, u0)
        assert np.isclose(cost, oracle.cost(u0, ub), rtol=1e-13)
        assert np.allclose(grad, oracle.gradient(P22, u0, ub), rtol=1e-6)
        u, cost, g = fn(0.1, 0.1, maxiter=300, fetch_u=False)
        assert u is None and isinstance(g, float)
    finally:
        if created:
            dist.destroy_process_group()


def test_plain_c_demo_runs(gpu_solver_cls, tmp_path):
    """The C demo of the ABI runs on the GPU without Python in the loop."""
    import os, subprocess
    from conftest import ROOT
    from bpldenoising_amd import _lib
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
                           "-o", str(exe), "-L", libdir, "-lbpltv", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-lm"])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "scalar alpha" in out.stdout and "cost curve" in out.stdout
    cost = float(out.stdout.split("cost ")[1].split()[0])
    assert 0 < cost < 1e4


@pytest.mark.parametrize("seed", range(6))
def test_randomised_configurations(gpu_solver_cls, oracle, seed):
    """Random shapes / parameters / iteration counts: u bit-exact, cost and gradient within tolerance."""
    rng = np.random.default_rng(1000 + seed)
    for _ in range(4):
        O = int(rng.integers(1, 4)); M = int(rng.integers(2, 120)); N = int(rng.integers(2, 120))
        ub, f = synth_batch(O, N, M, seed=int(rng.integers(1 << 30)))
        mode = rng.integers(0, 3)
        if mode == 0:
            alpha = float(rng.uniform(0.01, 0.3))
        elif mode == 1:
            alpha = rng.uniform(0.02, 0.3, size=(int(rng.integers(1, min(4, N) + 1)), int(rng.integers(1, min(4, M) + 1))))
        else:
            alpha = rng.uniform(0.02, 0.3, size=(N, M))
        kw = dict(maxiter=int(rng.integers(0, 260)), accel=bool(rng.integers(0, 2)), rho=float(rng.choice([0.0, 0.0, 0.02])),
                  tau0=float(rng.uniform(2, 6)), sigma0=float(rng.uniform(0.1, 0.19)))
        s = gpu_solver_cls(M, N, O)
        s.set_data(ub, f)
        u = s.denoise(alpha, tile_iters=int(rng.integers(0, 9)), **kw)
        u0 = oracle.pdhg(f, alpha, **kw)
        assert np.array_equal(u, u0), (O, N, M, mode, kw)
        if kw["maxiter"] >= 50 and mode != 2:
            _, cost, grad = s.evaluate(alpha, 0.1, **kw)
            assert np.isclose(cost, oracle.cost(u0, ub), rtol=1e-12)
            assert np.allclose(grad, oracle.gradient(alpha, u0, ub), rtol=2e-6, atol=1e-9)
        s.close()


def _gpu_rank_worker(rank, world, port, q):
    import os, sys
    from conftest import ROOT, synth_batch as sb
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bpldenoising_amd import ShardedLearningFunction
    ub, f = sb(5, 64, 64, seed=28)
    fn = ShardedLearningFunction((ub, f))          # HIP solver per rank, all ranks on cuda:0
    u, cost, grad = fn(np.array([[0.08, 0.12], [0.1, 0.05]]), 0.1, maxiter=300)
    fd = ShardedLearningFunction((ub, f), solver_factory=lambda *a: fn.solver, deterministic=True)
    fd.solver = fn.solver                          # same handle, rows added in global image order
    _, cd, gd = fd(np.array([[0.08, 0.12], [0.1, 0.05]]), 0.1, maxiter=300, fetch_u=False)
    q.put((rank, fn.lo, fn.hi, u, cost, np.asarray(grad), cd, np.asarray(gd)))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharded_evaluate_on_one_gpu(gpu_solver_cls, oracle):
    """Two processes, each with its own libbpltv handle over its block of images (here both on the
    one visible GPU), one all-reduce of [cost, grad] -- the N-GPU path with the device count of this box."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000
    procs = [ctx.Process(target=_gpu_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ub, f = synth_batch(5, 64, 64, seed=28)
    P = np.array([[0.08, 0.12], [0.1, 0.05]])
    u0 = oracle.pdhg(f, P, maxiter=300)
    c0, g0 = oracle.cost(u0, ub), oracle.gradient(P, u0, ub)
    s1 = gpu_solver_cls(64, 64, 5)                 # one handle over the whole batch
    s1.set_data(ub, f)
    _, c1, g1 = s1.evaluate(P, 0.1, maxiter=300, fetch_u=False)
    assert np.allclose(s1.per_image().sum(axis=0), np.concatenate([[c1], np.ravel(g1)]), rtol=1e-13)
    s1.close()
    for rank, lo, hi, u, cost, grad, cd, gd in res:
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        assert np.array_equal(u, u0[lo:hi])
        assert np.isclose(cost, c0, rtol=1e-13) and np.allclose(grad, g0, rtol=1e-6)
        # deterministic mode: bitwise what a single handle over all five images returns
        assert cd == c1 and np.array_equal(gd, g1)


@pytest.mark.parametrize("N", [1, 2, 3, 5, 17])
def test_adjoint_split_edge_cases(gpu_solver_cls, oracle, N):
    """M = 128 with very few image columns: the two-sided factorisation degenerates (sides of 0, 64,
    ... columns); results must not depend on which path runs."""
    ub, f = synth_batch(2, N, 128, seed=60 + N)
    s = gpu_solver_cls(128, N, 2)
    s.set_data(ub, f)
    for alpha in (0.1, np.array([[0.06, 0.15]])):
        u, cost, grad = s.evaluate(alpha, 0.1, maxiter=400)
        u0 = oracle.pdhg(f, alpha, maxiter=400)
        assert np.array_equal(u, u0)
        assert np.allclose(grad, oracle.gradient(alpha, u0, ub), rtol=2e-6, atol=1e-9)
        _, _, greg = s.evaluate(alpha, 0.0, maxiter=400)
        assert np.allclose(greg, oracle.gradient(alpha, u0, ub, reg=True), rtol=1e-7, atol=1e-11)
    s.close()


@pytest.mark.parametrize("shape", [(2, 2, 5), (2, 3, 16), (1, 29, 37), (2, 64, 100), (1, 13, 128), (3, 40, 1)])
def test_both_adjoint_factorisations_agree(gpu_solver_cls, oracle, shape):
    """The adjoint system is solved by nested dissection (the default), by block cyclic reduction (M <= 128, N >= 2:
    level 0 in operator form, dense MFMA levels above) or by the banded Cholesky; all must reproduce the oracle."""
    O, N, M = shape
    ub, f = synth_batch(O, N, M, seed=70 + N + M)
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    for alpha in (0.1, np.array([[0.06], [0.15]]) if N >= 2 else np.array([[0.06, 0.15]])):
        u0 = oracle.pdhg(f, alpha, maxiter=400)
        g0 = oracle.gradient(alpha, u0, ub)
        r0 = oracle.gradient(alpha, u0, ub, reg=True)
        res = {}
        for meth in ("band", "bcr", "nd"):
            _, _, g = s.evaluate(alpha, 0.1, maxiter=400, adjoint_method=meth)
            assert s.stats()["adjoint_method"] == meth
            _, _, r = s.evaluate(alpha, 0.0, maxiter=400, adjoint_method=meth)
            assert np.allclose(g, g0, rtol=2e-6, atol=1e-9), (meth, shape)
            assert np.allclose(r, r0, rtol=1e-7, atol=1e-11), (meth, shape)
            res[meth] = (np.asarray(g), np.asarray(r))
        assert np.allclose(res["band"][0], res["bcr"][0], rtol=1e-7, atol=1e-11)
        assert np.allclose(res["nd"][0], res["bcr"][0], rtol=1e-7, atol=1e-11)
        # the same call twice is bitwise reproducible (no atomics anywhere in the factorisations)
        for meth in ("bcr", "nd"):
            _, _, g2 = s.evaluate(alpha, 0.1, maxiter=400, adjoint_method=meth)
            assert np.array_equal(np.asarray(g2), res[meth][0])
        _, _, g3 = s.evaluate(alpha, 0.1, maxiter=400)              # automatic choice
        assert s.stats()["adjoint_method"] == "nd" and np.array_equal(np.asarray(g3), res["nd"][0])
    s.close()


def test_bcr_is_refused_where_it_does_not_apply(gpu_solver_cls):
    from bpldenoising_amd._lib import BpltvError
    ub, f = synth_batch(1, 3, 130, seed=5)       # M = 130 > 128
    s = gpu_solver_cls(130, 3, 1)
    s.set_data(ub, f)
    with pytest.raises(BpltvError) as e:
        s.evaluate(0.1, 0.1, maxiter=50, adjoint_method="bcr")
    assert e.value.code == 6
    u, c, g = s.evaluate(0.1, 0.1, maxiter=50)   # automatic choice: nested dissection
    assert np.isfinite(g) and s.stats()["adjoint_method"] == "nd"
    u, c, g1 = s.evaluate(0.1, 0.1, maxiter=50, adjoint_method="band")
    assert s.stats()["adjoint_method"] == "band" and np.isclose(g1, g, rtol=1e-7)
    s.close()


def test_bcr_kernel_unit_checks(gpu_solver_cls):
    """tools/bcr_unit.hip: Cholesky+inverse tile kernel, the two MFMA product kernels and complete
    factor/solve sequences (dense and operator-form level 0) against plain host loops."""
    import os, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "_bin", "bcr_unit")
    assert os.path.exists(exe), "built by __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all ok" in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


def test_band_solver_unit_checks(gpu_solver_cls):
    """tools/lu_unit.hip: the HBM band solvers (banded Cholesky plain and twisted, block LU without pivoting) on
    random band matrices against dense host elimination, incl. the diagonal sets of the sum-of-regularisers model."""
    import os, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "_bin", "lu_unit")
    assert os.path.exists(exe), "built by __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all ok" in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


def test_full_size_gradient_properties_1024(gpu_solver_cls):
    """BASELINE config 5 image size (1024 x 1024): properties that need no oracle run -- the refined adjoint solve
    reaches the residual level of the 128^2 cases, the stand-alone gradient entry reproduces the gradient of evaluate
    from its (u, ubar), and the two factorisations of the same system -- nested dissection (default, 0.65 GB of factor)
    and the HBM-resident band (8.6 GB) -- give the same gradient."""
    N = M = 1024
    ub, f = synth_batch(1, N, M, seed=12)
    s = gpu_solver_cls(M, N, 1)
    s.set_data(ub, f)
    u, cost, g = s.evaluate(0.1, 0.1, maxiter=300)
    st = s.stats()
    assert st["adjoint_method"] == "nd" and st["reg_gradient_used"] == 0 and st["adjoint_chunks"] == 1
    assert np.isfinite(g) and st["adjoint_residual"] <= 1e-8, st
    assert np.isclose(cost, 0.5 * np.sum((u - ub) ** 2), rtol=1e-12)
    g2 = s.gradient(u, ub, 0.1)
    assert np.isclose(g2, g, rtol=1e-12), (g, g2)            # same kernels, same data: reproducible
    gb = s.gradient(u, ub, 0.1, adjoint_method="band")
    assert s.stats()["adjoint_method"] == "band-hbm" and np.isclose(gb, g, rtol=1e-6), (g, gb)
    s.close()


def test_config5_share_evaluate_8x1024(gpu_solver_cls):
    """BASELINE config 5's share of one GPU through the whole learning function: 8 x 1024 x 1024, pixelwise
    alpha, PDHG + loss + the nested-dissection adjoint (8 x 0.65 GB factors).  Properties that need
    no oracle run: the factorisation is nested dissection, the scaled residual passes the gate, the loss equals the
    host sum, the pixel-map gradient is finite with the sign of the reference (increasing alpha from a
    small value lowers the loss on noisy data: sum of the map gradient < 0), and the stand-alone gradient entry
    reproduces it from (u, ubar)."""
    O, N, M = 8, 1024, 1024
    ub, f = synth_batch(O, N, M, seed=3)
    jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
    amap = 0.02 + 0.01 * np.sin(2 * np.pi * ii / M) * np.cos(2 * np.pi * jj / N)
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    u, cost, g = s.evaluate(amap, 0.1, maxiter=400)
    st = s.stats()
    assert st["adjoint_method"] == "nd" and st["adjoint_attempts"] == 1 and st["adjoint_chunks"] == 1
    assert st["adjoint_residual"] <= 1e-8, st
    assert st["adjoint_ms"] < 150.0, st["adjoint_ms"]      # the banded Cholesky needed 500 ms here
    assert g.shape == (N, M) and np.isfinite(g).all() and g.sum() < 0
    assert np.isclose(cost, 0.5 * np.sum((u - ub) ** 2), rtol=1e-12)
    g2 = s.gradient(u, ub, amap)
    assert np.array_equal(g2, g)                          # same kernels, same data, no atomics: reproducible
    s.close()


@pytest.mark.parametrize("O", [16, 40])
def test_larger_batches_of_small_images_same_bits_whatever_the_plan(gpu_solver_cls, oracle, O):
    """Batches beyond the reference's ten images: the launch-cost model moves between the 32x32 kernel with two launch
    chains, 48x48 regions (3 px per thread) and the 64-lane rows kernels as the batch grows; whatever it picks, same bits."""
    ub, f = synth_batch(O, 128, 128, seed=90)
    s = gpu_solver_cls(128, 128, O)
    s.set_data(ub, f)
    u = s.denoise(0.07, maxiter=300)
    st = s.stats()
    assert st["region_i"] in (32, 48, 64) and 2 <= st["tile_iters"] <= 12 and st["graph_used"] == 1, st
    assert np.array_equal(u, oracle.pdhg(f, 0.07, maxiter=300, nthreads=8))
    assert np.array_equal(u, s.denoise(0.07, maxiter=300, variant=13, tile_iters=10, chains=1))      # the 48x48 plan explicitly
    s.close()


def test_hbm_pipeline_event_and_value_waits_agree(gpu_solver_cls, monkeypatch):
    """The three-stream factorisation resolves its cross-stream dependencies by stream memory operations (default) or
    by HIP events (bpltv_set_option "hb_sync" = 1, what rocprofv3 runs need): same kernels in the same order, so the gradients
    are bitwise equal -- a missing dependency in either would show here (twisted, 3 images, 6 half-problems)."""
    O, N, M = 3, 48, 300
    ub, f = synth_batch(O, N, M, seed=77)
    amap = 0.05 + 0.1 * np.random.default_rng(9).random((N, M))
    res = []
    for mode in ("value", "event", "single-stream"):   # the last: everything on one stream (the profiling aid)
        s = gpu_solver_cls(M, N, O)
        s.set_option("hb_sync", 2 if mode == "value" else 1)
        s.set_option("hb_single_stream", 1 if mode == "single-stream" else 0)
        s.set_data(ub, f)
        _, _, g = s.evaluate(amap, 0.1, maxiter=300, adjoint_method="band")
        st = s.stats()
        assert st["adjoint_method"] == "band-hbm" and st["adjoint_residual"] <= 1e-8
        # the mode that ran, as the library reports it: hb_sync = 2 (value) fails hard when stream memory operations
        # are unavailable, so the 'value' leg cannot pass as a second 'event' leg
        assert st["hb_sync"] == ("value" if mode == "value" else "event"), st["hb_sync"]
        res.append(g)
        s.close()
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])


def test_skinny_fronts_whatever_the_batch_size(gpu_solver_cls):
    """nd_front_skinny2_kernel (fronts of 17..32 pivots with two pivot block columns in LDS) takes a level only from 256
    (front, image) pairs on; option "nd_skinny2_min" = 0 sends the 32 such fronts of two 150 x 139 images through it: the
    same gradient as the large-regime kernels to 1e-9 (another order of the same operations), the same quality gate, and
    against the oracle like every other path."""
    from oracle import c_oracle as co
    O, N, M = 2, 139, 150
    ub, f = synth_batch(O, N, M, seed=139)
    alpha = 0.04 + 0.1 * np.random.default_rng(3).random((N, M))
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    u0, c0, g0 = s.evaluate(alpha, 0.1, maxiter=300)
    r0 = s.stats()["adjoint_residual"]
    s.set_option("nd_skinny2_min", 0)
    u1, c1, g1 = s.evaluate(alpha, 0.1, maxiter=300)
    st = s.stats()
    assert np.array_equal(u0, u1) and c0 == c1 and st["adjoint_method"] == "nd"
    assert np.allclose(g1, g0, rtol=1e-9, atol=1e-12 * np.abs(g0).max()) and st["adjoint_residual"] < 1e-8 and r0 < 1e-8
    s.set_option("nd_skinny", 0)            # and none of the skinny kernels at all
    _, _, g2 = s.evaluate(alpha, 0.1, maxiter=300)
    assert np.allclose(g2, g0, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    s.close()


def test_nd_solver_unit_checks(gpu_solver_cls):
    """tools/nd_unit.hip: the nested-dissection Cholesky (fronts in LDS, fronts in HBM with multi-panel pivot blocks,
    gather-form substitutions) against the host restatement tools/nd_ref.hpp: factor entries and solutions on random
    SPD stencil matrices, several images per call, both stencils."""
    import os, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "_bin", "nd_unit")
    assert os.path.exists(exe), "built by __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all ok" in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.parametrize("shape,alpha,method,budget_mb", [
    ((7, 128, 128), 0.1, "bcr", 300.0),                                  # 117 MB per image: groups of 2
    ((5, 48, 300), "map", "nd", None),                                   # budget computed below: groups of 2
    ((5, 128, 128), np.array([[0.06, 0.15], [0.1, 0.2]]), "nd", None),   # nested dissection on a small image
], ids=["bcr_scalar", "nd_map_wide", "nd_patch_128"])
def test_adjoint_in_image_groups_is_bitwise_the_whole_batch(gpu_solver_cls, monkeypatch, shape, alpha, method, budget_mb):
    """When the factor workspace of all images does not fit in HBM the gradient runs in image groups (the
    reference's loop is sequential and has no such limit, /root/reference/src/TVLearningFunctionVec.jl:76-81).
    bpltv_set_option "adjoint_budget_mb" forces that on a small case: cost and gradient are BITWISE those of the whole batch at once
    (per-image results do not depend on the other images of a launch; sums run per image, in image order)."""
    O, N, M = shape
    ub, f = synth_batch(O, N, M, seed=5 + M)
    if isinstance(alpha, str):
        alpha = 0.05 + 0.1 * np.random.default_rng(8).random((N, M))
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    u0, c0, g0 = s.evaluate(alpha, 0.1, maxiter=300, adjoint_method=method)
    st0 = s.stats()
    assert st0["adjoint_chunks"] == 1 and st0["adjoint_method"] == method
    rows0 = s.per_image() if np.size(alpha) < M * N else None
    s.close()
    if budget_mb is None:       # two images' worth of this shape's nested-dissection workspace
        import ctypes as C
        budget_mb = 2.5 * _nd_bytes_per_image(M, N) / 1e6
    s = gpu_solver_cls(M, N, O)
    s.set_option("adjoint_budget_mb", budget_mb)
    s.set_data(ub, f)
    u1, c1, g1 = s.evaluate(alpha, 0.1, maxiter=300, adjoint_method=method)
    st1 = s.stats()
    assert st1["adjoint_chunks"] == (O + 1) // 2, st1
    assert np.array_equal(u1, u0) and c1 == c0 and np.array_equal(np.asarray(g1), np.asarray(g0))
    if rows0 is not None:
        assert np.array_equal(s.per_image(), rows0)
    _, _, r1 = s.evaluate(alpha, 0.0, maxiter=300, adjoint_method=method)       # gradient_reg in groups too
    s.close()
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    _, _, r0 = s.evaluate(alpha, 0.0, maxiter=300, adjoint_method=method)
    assert np.array_equal(np.asarray(r1), np.asarray(r0))
    # a budget below one image's workspace is an error that says so, not a crash
    from bpldenoising_amd._lib import BpltvError
    s2 = gpu_solver_cls(M, N, O)
    s2.set_option("adjoint_budget_mb", 0.001)
    s2.set_data(ub, f)
    with pytest.raises(BpltvError) as e:
        s2.evaluate(alpha, 0.1, maxiter=10, adjoint_method=method)
    assert e.value.code == 5 and "ONE" in str(e.value)
    s2.close()
    s.close()


def _nd_bytes_per_image(M, N):
    """Workspace of the nested-dissection solver per image, from the host check tool (the same symbolic code)."""
    import os, re, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "_bin", "nd_host_check")
    out = subprocess.run([exe, "bytes", str(M), str(N)], capture_output=True, text=True, timeout=120).stdout
    return float(re.search(r"bytes_per_image tv (\d+)", out).group(1))


def test_config5_whole_batch_evaluate_64x1024(gpu_solver_cls):
    """BASELINE config 5's WHOLE batch on one GPU: 64 x 1024 x 1024, pixelwise alpha -- PDHG state 4 GB, adjoint
    planes 8 GB, nested-dissection factors 64 x 0.94 GB.  The banded path returned BPLTV_E_NOMEM here (8.6 GB per
    image); the reference's per-image loop has no such limit.  Few iterations: this is about capacity."""
    O, N, M = 64, 1024, 1024
    rng = np.random.default_rng(1)
    base_ub, base_f = synth_batch(4, N, M, seed=9)
    idx = np.arange(O) % 4
    ub = base_ub[idx] * (1.0 - 0.01 * (np.arange(O) // 4))[:, None, None]
    f = np.clip(base_f[idx] + 0.002 * rng.standard_normal((O, 1, 1)), 0, 1)
    jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
    amap = 0.02 + 0.01 * np.sin(2 * np.pi * ii / M) * np.cos(2 * np.pi * jj / N)
    s = gpu_solver_cls(M, N, O)
    s.set_data(ub, f)
    u, cost, g = s.evaluate(amap, 0.1, maxiter=40, fetch_u=False)
    st = s.stats()
    assert st["adjoint_method"] == "nd" and st["adjoint_residual"] <= 1e-8, st
    assert g.shape == (N, M) and np.isfinite(g).all() and np.isfinite(cost)
    s.close()
