#!/usr/bin/env python3
"""Developer probe: evaluate (with gradient) on 1024x1024 images -- HBM band path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
for (O, n) in ((1, 1024), (8, 1024)):
    ub, f = synth_batch(O, n, n, seed=3)
    jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    amap = 0.11 + 0.09 * np.sin(2 * np.pi * ii / n) * np.cos(2 * np.pi * jj / n)
    s = TVSolver(n, n, O); s.set_data(ub, f)
    for alpha, name in ((0.1, "scalar"), (amap, "map")):
        try:
            t = time.time(); u, c, g = s.evaluate(alpha, 0.1, fetch_u=False, maxiter=2000); dt = time.time() - t
        except Exception as e:
            print("O %d %dx%d %s FAILED: %s" % (O, n, n, name, e), flush=True); continue
        st = s.stats()
        print("O %d %dx%d %-6s: evaluate %.2f s  pdhg %.1f ms adjoint %.1f ms residual %.2e cost %.4f grad %s" % (
            O, n, n, name, dt, st["pdhg_ms"], st["adjoint_ms"], st["adjoint_residual"], c,
            ("%.5f" % g) if np.ndim(g) == 0 else ("map sum %.5f" % g.sum())), flush=True)
    s.close()
