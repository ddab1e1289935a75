"""Dev probe: 128^2 batches: 32x32 (variant 1) vs 40x40 (11, 12) vs 48x48 (13) regions, fusion depths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
for O in (1, 10, 16, 32, 64):
    ub, f = synth_batch(O, 128, 128, seed=1)
    s = TVSolver(128, 128, O); s.set_data(ub, f)
    res = []
    for var in (1, 11, 12, 13):
        for T in (4, 6, 8, 10, 12):
            try:
                t = []
                for _ in range(4):
                    s.denoise(0.1, fetch=False, maxiter=5000, variant=var, tile_iters=T); st = s.stats(); t.append(st["pdhg_ms"])
                if st["tile_iters"] == T: res.append((min(t[1:]), var, T, st["tiles"]))
            except Exception as e:
                pass
    s.denoise(0.1, fetch=False, maxiter=5000); s.denoise(0.1, fetch=False, maxiter=5000); st = s.stats()
    res.sort()
    print("O %2d auto: variant? T %d tiles %d %.3f ms | best:" % (O, st["tile_iters"], st["tiles"], st["pdhg_ms"]), " ".join("v%d/T%d/%d:%.3f" % (v, T, n, m) for m, v, T, n in res[:5]), flush=True)
    s.close()
