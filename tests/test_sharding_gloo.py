"""N > 1 path on CPU: world_size-2 gloo processes, images sharded, one all-reduce of [cost, grad]."""
import os
import sys
import numpy as np
import pytest
from conftest import ROOT, synth_batch


def _worker(rank, world, port, q, alpha, nimg=5, deterministic=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bpldenoising_amd import ShardedLearningFunction
    from test_host_logic import FakeSolver
    ub, f = synth_batch(nimg, 18, 14, seed=33)
    fn = ShardedLearningFunction((ub, f), solver_factory=FakeSolver, deterministic=deterministic)
    u, cost, grad = fn(alpha, 0.1, maxiter=150)
    q.put((rank, fn.lo, fn.hi, None if u is None else u.copy(), cost, np.asarray(grad).copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("alpha", [0.1, np.array([[0.05, 0.1], [0.2, 0.08]])], ids=["scalar", "patch"])
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_evaluate(oracle, alpha, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, alpha)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ub, f = synth_batch(5, 18, 14, seed=33)
    u0, c0, g0 = oracle.tv_op_learning_function(alpha, (ub, f), 0.1, maxiter=150)
    covered = np.zeros(5, bool)
    for rank, lo, hi, u, cost, grad in res:
        assert np.isclose(cost, c0, rtol=1e-14)                    # identical totals on every rank
        assert np.allclose(grad, g0, rtol=1e-12)
        assert np.array_equal(u, u0[lo:hi])                        # each rank holds its block of u
        covered[lo:hi] = True
    assert covered.all()


def test_more_ranks_than_images(oracle):
    """cameraman-style datasets hold ONE image: ranks without images contribute zeros to the all-reduce
    and still receive the batch totals."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 0.1, 1)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ub, f = synth_batch(1, 18, 14, seed=33)
    u0, c0, g0 = oracle.tv_op_learning_function(0.1, (ub, f), 0.1, maxiter=150)
    for rank, lo, hi, u, cost, grad in res:
        assert np.isclose(cost, c0, rtol=1e-14) and np.allclose(grad, g0, rtol=1e-12)
        if rank == 0:
            assert (lo, hi) == (0, 1) and np.array_equal(u, u0)
        else:
            assert lo == hi and u is None


def test_deterministic_totals_do_not_depend_on_the_world_size(oracle):
    """deterministic=True: per-image rows all-gathered and added in global image order -- bitwise the same
    totals for world sizes 1, 2 and 3 (an all-reduce of per-rank sums only agrees to rounding)."""
    import torch.multiprocessing as mp
    from bpldenoising_amd import ShardedLearningFunction
    from test_host_logic import FakeSolver
    P = np.array([[0.05, 0.1], [0.2, 0.08]])
    ub, f = synth_batch(5, 18, 14, seed=33)
    fn = ShardedLearningFunction((ub, f), solver_factory=FakeSolver, deterministic=True)     # world size 1
    _, c1, g1 = fn(P, 0.1, maxiter=150)
    ctx = mp.get_context("spawn")
    for world in (2, 3):
        q = ctx.Queue()
        port = 29300 + (os.getpid() % 2000) + world
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, P, 5, True)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=180) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, lo, hi, u, cost, grad in res:
            assert cost == c1 and np.array_equal(grad, g1), (world, rank)
