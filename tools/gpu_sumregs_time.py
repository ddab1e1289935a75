#!/usr/bin/env python3
"""Developer probe: sum-of-regularisers learning function on the reference's batch shapes (timing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver, testdataset
NPZ = os.path.join(ROOT, "tests/golden/datasets.npz")
A3 = np.array([0.03, 0.02, 0.05])
P3 = 0.02 + 0.03 * np.random.default_rng(0).random((3, 2, 2))
for ds, O in (("faces_train", 10), ("cameraman_128_10", 1)):
    ub, f = testdataset(ds, npz=NPZ)
    s = TVSolver(128, 128, O); s.set_data(ub[:O], f[:O])
    for name, a, delta in (("vector", A3, 0.1), ("vector reg", A3, 1e-4), ("patch", P3, 0.1), ("patch reg (LU)", P3, 1e-4)):
        for T, var in ((0, 0), (4, 1), (4, 2), (5, 2), (2, 1)):
            best = None
            for _ in range(3):
                t = time.time(); s.sumregs_evaluate(a, delta, fetch_u=False, tile_iters=T, variant=var); dt = time.time() - t
                st = s.stats()
                if best is None or dt < best[0]:
                    best = (dt, st)
            dt, st = best
            print("%-12s O %2d %-15s region %d T %d: evaluate %.1f ms  pdhg %.2f ms (%d launches, %d tiles) adjoint %.2f ms residual %.1e" % (
                ds, O, name, st["region_i"], st["tile_iters"], 1e3 * dt, st["pdhg_ms"], st["launches"], st["tiles"], st["adjoint_ms"], st["adjoint_residual"]), flush=True)
            if name != "vector":
                break
    s.close()
