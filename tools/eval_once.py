#!/usr/bin/env python3
"""Developer probe: one full evaluate on the 10x128x128 batch (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10)
s.set_data(ub, f)
for alpha in (0.1, np.array([[0.08, 0.12], [0.1, 0.05]])):
    for delta in (0.1, 0.0):
        u, c, g = s.evaluate(alpha, delta, fetch_u=False)
        st = s.stats()
        print("alpha", np.shape(alpha), "delta", delta, "cost", c, "grad", np.ravel(g), "pdhg_ms %.2f adjoint_ms %.2f res %.2e" % (st["pdhg_ms"], st["adjoint_ms"], st["adjoint_residual"]), flush=True)
