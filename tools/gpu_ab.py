#!/usr/bin/env python3
"""Developer probe: timing of the default configuration (median of repeats) + parity on the batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from oracle import c_oracle as co
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10)
s.set_data(ub, f)
u = s.denoise(0.1, maxiter=1000)
print("bitexact vs oracle (1000 it):", np.array_equal(u, co.pdhg(f, 0.1, maxiter=1000, nthreads=8)))
cfgs = [dict(), dict(chains=2), dict(tile_iters=7), dict(tile_iters=6), dict(tile_iters=9), dict(tile_iters=10)]
res = {i: [] for i in range(len(cfgs))}
for rep in range(10):
    for i, c in enumerate(cfgs):
        s.denoise(0.1, fetch=False, maxiter=5000, **c)
        res[i].append(s.stats()["pdhg_ms"])
for i, c in enumerate(cfgs):
    a = np.array(res[i][2:])
    print("%-24s median %.3f ms min %.3f -> %.0f it/s" % (c, np.median(a), a.min(), 5e6 / np.median(a)))
