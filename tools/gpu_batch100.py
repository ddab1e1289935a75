"""Dev probe: 100-problem batches (100 images, and a 100-parameter sweep over one image)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
ub, f = synth_batch(100, 128, 128, seed=1)
s = TVSolver(128, 128, 100); s.set_data(ub, f)
for _ in range(3):
    s.denoise(0.1, fetch=False, maxiter=5000); st = s.stats()
print("100 images x 5000 its: %.2f ms (T %d, tiles %d)" % (st["pdhg_ms"], st["tile_iters"], st["tiles"]), flush=True)
s.close()
s = TVSolver(128, 128, 1); s.set_data(ub[:1], f[:1])
al = np.linspace(0.001, 0.2, 100)
for _ in range(2):
    t = time.time(); c = s.sweep(al, maxiter=10000); dt = time.time() - t
print("sweep 100 alpha x 10000 its on one image: %.3f s, argmin alpha %.4f" % (dt, al[int(np.argmin(c))]), flush=True)
