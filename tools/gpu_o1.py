#!/usr/bin/env python3
"""Developer probe: small batches (O = 1, 2, 4) -- which fusion depth / variant is best?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
for O in (1, 2, 4, 20, 40):
    ub, f = synth_batch(O, 128, 128, seed=1)
    s = TVSolver(128, 128, O)
    s.set_data(ub, f)
    best = None
    for var in (1, 3, 4):
        for T in (4, 6, 8, 10, 12):
            try:
                t = []
                for _ in range(4):
                    s.denoise(0.1, fetch=False, maxiter=5000, variant=var, tile_iters=T)
                    st = s.stats(); t.append(st["pdhg_ms"])
                m = min(t[1:])
                if T == st["tile_iters"]:
                    print("O %2d var %d T %2d tiles %4d: %.3f ms" % (O, var, T, st["tiles"], m), flush=True)
                    if best is None or m < best[0]: best = (m, var, T)
            except Exception as e:
                print("ERR", O, var, T, e)
    print("O", O, "best", best, flush=True)
    s.close()
