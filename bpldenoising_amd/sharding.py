"""Images of a training batch sharded over ranks (one process per GPU), one all-reduce of the
per-shard [cost, grad...] vector per evaluation.

Why this is exact: images are independent ROF problems sharing only alpha
(/root/reference/src/TVLearningFunctionVec.jl:57-65), the loss is a plain sum over all entries
(:20) and both gradient wrappers sum per-image contributions (:76-82, :168-173).  No halo, no
other collective.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import numpy as np


def shard_range(O, world, rank):
    """Block distribution of O images over `world` ranks: the first O % world ranks get one more."""
    base, rem = divmod(int(O), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ShardedLearningFunction:
    """tv_op_learning_function over a batch sharded across the ranks of a torch.distributed group.

    data = (ubar, f): the full batch (every rank slices its own block) as (O, N, M) arrays.
    solver_factory(M, N, O_local) -> object with set_data / evaluate_partial[/evaluate_device];
    default: the HIP TVSolver.  Calling the object returns (u_local, cost, grad) where cost and
    grad are the all-reduced batch totals and u_local is this rank's block of denoised images.

    deterministic=True (scalar / patch parameters): instead of all-reducing per-rank sums, the per-image
    rows [cost_k, grad_k...] (solver.per_image()) are all-gathered and added in global image order on
    every rank -- cost and grad are then bitwise the same for every world size, and equal to what a
    single handle over the whole batch returns (O <= 64 rows of a few doubles).
    """

    def __init__(self, data, group=None, solver_factory=None, device_reduce=None, deterministic=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ubar, f = np.asarray(data[0], dtype=np.float64), np.asarray(data[1], dtype=np.float64)
        if f.ndim == 2:
            ubar, f = ubar[None], f[None]
        self.O, self.N, self.M = f.shape
        self.lo, self.hi = shard_range(self.O, self.world, self.rank)
        self.solver = None
        if solver_factory is None:
            from .learning_function import TVSolver
            solver_factory = TVSolver
        if self.hi > self.lo:
            self.solver = solver_factory(self.M, self.N, self.hi - self.lo)
            self.solver.set_data(ubar[self.lo:self.hi], f[self.lo:self.hi])
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self.device_reduce = (backend == "nccl") if device_reduce is None else device_reduce
        self.deterministic = bool(deterministic)

    def __call__(self, x, delta, fetch_u=True, **kw):
        import torch
        a = np.asarray(x, dtype=np.float64)
        scalar = a.ndim == 0
        npar = 1 if scalar else a.size
        u = None
        if self.device_reduce:
            t = torch.zeros(1 + npar, dtype=torch.float64, device="cuda")
            if self.solver is not None:
                self.solver.evaluate_device(x, delta, t.data_ptr(), **kw)
                if fetch_u:
                    u = torch.empty((self.hi - self.lo, self.N, self.M), dtype=torch.float64, device="cuda")
                    self.solver.copy_u_device(u.data_ptr())
                    u = u.cpu().numpy()
        else:
            part = np.zeros(1 + npar)
            if self.solver is not None:
                u, part = self.solver.evaluate_partial(x, delta, fetch_u=fetch_u, **kw)
            t = torch.from_numpy(np.ascontiguousarray(part))
        if self.deterministic and a.size < self.M * self.N:
            tot = self._ordered_total(npar, on_device=self.device_reduce)
        else:
            if self.world > 1:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            tot = t.cpu().numpy()
        cost = float(tot[0])
        grad = float(tot[1]) if scalar else tot[1:].reshape(a.shape if a.ndim == 2 else (1, -1)).copy()
        return u, cost, grad

    def _ordered_total(self, npar, on_device):
        """All-gather the per-image rows and add them in global image order (plain left-to-right sums)."""
        import torch
        maxloc = -(-self.O // self.world)
        rows = np.zeros((maxloc, 1 + npar))
        if self.solver is not None:
            rows[:self.hi - self.lo] = self.solver.per_image()
        if self.world > 1:
            t = torch.from_numpy(rows)
            if on_device:
                t = t.cuda()
            parts = [torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(parts, t, group=self.group)
            parts = [q.cpu().numpy() for q in parts]
        else:
            parts = [rows]
        tot = np.zeros(1 + npar)
        for r, q in enumerate(parts):
            lo, hi = shard_range(self.O, self.world, r)
            for k in range(hi - lo):
                tot = tot + q[k]
        return tot
