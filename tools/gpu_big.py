#!/usr/bin/env python3
"""Developer probe: large batches of 128x128 images -- 32x32 vs 64x64 regions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
for O in (16, 40, 100):
    ub, f = synth_batch(O, 128, 128, seed=1)
    s = TVSolver(128, 128, O); s.set_data(ub, f)
    for var in (1, 2, 5, 10):
        for T in (3, 4, 6, 8):
            t = []
            for _ in range(3):
                s.denoise(0.1, fetch=False, maxiter=5000, variant=var, tile_iters=T); st = s.stats(); t.append(st["pdhg_ms"])
            print("O %3d variant %2d T %d tiles %5d: %.3f ms" % (O, var, st["tile_iters"], st["tiles"], min(t[1:])), flush=True)
    s.close()
