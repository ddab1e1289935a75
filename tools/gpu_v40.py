#!/usr/bin/env python3
"""Developer probe: 40x40 / 48x48 region variants vs the 32x32 default (10x128x128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10); s.set_data(ub, f)
ref = s.denoise(0.1, maxiter=1000)
cfgs = [(1, 8), (11, 8), (11, 6), (11, 10), (11, 12), (12, 8), (12, 6), (13, 8), (13, 12), (13, 16)]
res = {c: [] for c in cfgs}
for c in cfgs:
    ok = np.array_equal(s.denoise(0.1, maxiter=1000, variant=c[0], tile_iters=c[1]), ref)
    res[c].append(ok)
for rep in range(5):
    for c in cfgs:
        s.denoise(0.1, fetch=False, maxiter=5000, variant=c[0], tile_iters=c[1]); st = s.stats()
        res[c].append(st["pdhg_ms"]); res[c].append(st["tiles"])
for c in cfgs:
    print("variant %2d T %2d tiles %4d bitexact %s median %.3f ms -> %.0f it/s" % (c[0], c[1], res[c][2], res[c][0], np.median(res[c][3::2]), 5e6 / np.median(res[c][3::2])), flush=True)
