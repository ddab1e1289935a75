"""Dev probe: large-image default variant: 64x64 (variant 2) vs 48x48 (variant 13) vs 64x32 (variant 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
for (O, n, m) in ((1, 1024, 1024), (8, 512, 512), (2, 2048, 2048), (4, 300, 260), (1, 257, 700), (16, 1024, 1024)):
    ub, f = synth_batch(O, n, m, seed=3)
    s = TVSolver(m, n, O); s.set_data(ub, f)
    for alpha, nm in ((0.1, "scalar"),):
        for var in (2, 13, 5):
            for T in (6, 8):
                t = []
                for _ in range(2):
                    s.denoise(alpha, fetch=False, maxiter=1000, variant=var, tile_iters=T); st = s.stats(); t.append(st["pdhg_ms"])
                print("O %2d %4dx%-4d %s variant %2d T %d tiles %6d: %8.2f ms per 1000 its" % (O, m, n, nm, var, st["tile_iters"], st["tiles"], t[-1]), flush=True)
    s.close()
