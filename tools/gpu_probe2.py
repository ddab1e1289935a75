#!/usr/bin/env python3
"""Developer probe: where does the fixed per-launch time go? (timing-only debug flags)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10)
s.set_data(ub, f)
for T in (4, 8):
    for chains in (1, 2):
        for dbg, name in ((0, "full"), (4, "no-iterations"), (5, "no-iter,no-state-loads"), (6, "no-iter,no-stores"), (7, "no-iter,no-loads,no-stores"), (1, "iter,no-state-loads"), (2, "iter,no-stores"), (3, "iter,no-loads,no-stores")):
            t = []
            for _ in range(4):
                s.denoise(0.1, fetch=False, maxiter=5000, variant=1, tile_iters=T, chains=chains, dbg=dbg)
                st = s.stats(); t.append(st["pdhg_ms"])
            print("T %d chains %d %-28s: %.3f ms  per-launch %.2f us" % (T, chains, name, min(t), 1e3 * min(t) / (st["launches"] / chains)), flush=True)
