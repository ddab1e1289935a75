#!/usr/bin/env python3
"""Developer probe: the banded LU path of the sum-of-regularisers adjoint on symmetric systems (BPLTV_SR_FORCE_LU=1)
against the Cholesky path and the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from oracle import c_oracle as co
from conftest import synth_batch
A3 = np.array([0.03, 0.02, 0.05])
P3 = np.stack([np.array([[0.03, 0.05], [0.02, 0.04]]), np.array([[0.02, 0.03], [0.05, 0.02]]), np.array([[0.04, 0.02], [0.03, 0.06]])])
for (O, N, M) in ((1, 48, 40), (2, 64, 64)):
    ub, f = synth_batch(O, N, M, seed=21)
    s = TVSolver(M, N, O); s.set_data(ub, f)
    for name, a in (("vector", A3), ("patch", P3)):
        u0 = co.sumregs_pdhg(f, a, maxiter=800, nthreads=4)
        for delta in (0.1, 1e-4):
            g0 = co.sumregs_gradient(a, u0, ub, reg=not (delta > 1e-3))
            try:
                u, c, g = s.sumregs_evaluate(a, delta, maxiter=800)
                st = s.stats()
                print("%dx%dx%d %-6s delta %g: residual %.2e attempts %d  max rel diff vs oracle %.2e" % (
                    O, N, M, name, delta, st["adjoint_residual"], st["adjoint_attempts"], np.abs(g - g0).max() / np.abs(g0).max()), flush=True)
            except Exception as e:
                print("%dx%dx%d %-6s delta %g: FAILED %s" % (O, N, M, name, delta, e), flush=True)
    s.close()
