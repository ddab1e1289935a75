#!/usr/bin/env python3
"""Developer probe: three-dual PDHG kernels (params variant 1..5) on larger images -- ms per 1000 iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
A3 = np.array([0.03, 0.02, 0.05])
rng = np.random.default_rng(0)
for O, n in ((10, 128), (4, 256), (4, 512), (2, 1024)):
    ub = rng.random((O, n, n)); f = ub + 0.1 * rng.standard_normal((O, n, n))
    s = TVSolver(n, n, O); s.set_data(ub, f)
    ref = None
    for var, T in ((1, 4), (2, 4), (0, 0)):
        for _ in range(2):
            s.sumregs_denoise(A3, maxiter=1000, tile_iters=T, variant=var, fetch=False)
        st = s.stats()
        u = s.sumregs_denoise(A3, maxiter=1000, tile_iters=T, variant=var)
        if ref is None:
            ref = u
        print("%d x %d^2 variant %d region %d T %d: %.2f ms per 1000 iterations (%d tiles)  same bits %s" % (
            O, n, var, st["region_i"], st["tile_iters"], st["pdhg_ms"], st["tiles"], np.array_equal(u, ref)), flush=True)
    s.close()
