#!/usr/bin/env python3
"""Developer probe: automatic fusion depth vs batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
for (O, n) in ((1, 128), (2, 128), (3, 128), (5, 128), (10, 128), (16, 128), (20, 128), (40, 128), (100, 128), (4, 256), (1, 64), (50, 64)):
    ub, f = synth_batch(O, n, n, seed=1)
    s = TVSolver(n, n, O); s.set_data(ub, f)
    t = []
    for _ in range(4):
        s.denoise(0.1, fetch=False, maxiter=5000); st = s.stats(); t.append(st["pdhg_ms"])
    print("O %3d %dx%d: auto T %2d tiles %5d: %.3f ms  -> %.0f it/s (x%d images)" % (O, n, n, st["tile_iters"], st["tiles"], min(t[1:]), 5e6 / min(t[1:]), O), flush=True)
    s.close()
