"""Dev probe: PDHG kernel variants x fusion depth on the config-5 per-GPU shape (8 x 1024^2, alpha map)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
O, n = 8, 1024
ub, f = synth_batch(O, n, n, seed=3)
jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
amap = 0.11 + 0.09 * np.sin(2 * np.pi * ii / n) * np.cos(2 * np.pi * jj / n)
s = TVSolver(n, n, O); s.set_data(ub, f)
for var in range(1, 14):
    for T in (6, 8, 10):
        try:
            t = []
            for _ in range(2):
                s.denoise(amap, fetch=False, maxiter=1000, variant=var, tile_iters=T); st = s.stats(); t.append(st["pdhg_ms"])
            if st["tile_iters"] == T:
                print("variant %2d T %2d tiles %6d: %.2f ms per 1000 its -> %.0f it/s" % (var, T, st["tiles"], t[-1], 1e6 / t[-1]), flush=True)
        except Exception as e:
            print("variant", var, "T", T, "ERR", str(e)[-60:], flush=True)
