#!/usr/bin/env python3
"""Developer probe: pdhg_rows_kernel variants against the 32x32 tile kernel on random shapes, depths, iteration counts and
parameter forms (same bits expected).  usage: gpu_fuzz_rows.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
RJ = {19: 64, 20: 48, 21: 128, 22: 64, 23: 96, 24: 64, 25: 48, 26: 80, 27: 96, 28: 48, 29: 64}
bad = 0
for c in range(ncase):
    var = int(rng.choice(list(RJ)))
    M = int(rng.integers(64, 330)); N = int(rng.integers(RJ[var], RJ[var] + 300)); O = int(rng.integers(1, 4))
    T = int(rng.integers(1, 13)); it = int(rng.integers(1, 60)); rho = float(rng.choice([0.0, 0.0, 0.2]))
    mode = int(rng.integers(0, 3))
    alpha = [0.1, 0.03 + 0.1 * rng.random((2, 3)), 0.03 + 0.15 * rng.random((N, M))][mode]
    ub = rng.random((O, N, M)); f = ub + 0.1 * rng.standard_normal((O, N, M))
    s = TVSolver(M, N, O); s.set_data(ub, f)
    u0 = s.denoise(alpha, maxiter=it, rho=rho, variant=1)
    u1 = s.denoise(alpha, maxiter=it, rho=rho, variant=var, tile_iters=T)
    ok = np.array_equal(u0, u1)
    bad += not ok
    print("%s case %2d: variant %d T %2d (used %d) %dx%dx%d iters %d rho %.1f alpha %s" % ("ok  " if ok else "FAIL", c, var, T, s.stats()["tile_iters"], O, N, M, it, rho, ["scalar", "patch", "map"][mode]), flush=True)
    s.close()
print("gpu_fuzz_rows: %d cases, %d failed" % (ncase, bad))
sys.exit(1 if bad else 0)
