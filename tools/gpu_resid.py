#!/usr/bin/env python3
"""Developer probe: scaled vs raw adjoint residual, weight used and attempts, per factorisation and parameter kind."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver, testdataset
from conftest import synth_batch
P22 = np.array([[0.08, 0.12], [0.1, 0.05]])
ub, f = testdataset("faces_train", npz=os.path.join(ROOT, "tests/golden/datasets.npz"))
amap = 0.05 + 0.1 * np.random.default_rng(4).random((128, 128))
s = TVSolver(128, 128, 10); s.set_data(ub, f)
for name, a in (("scalar", 0.1), ("patch22", P22), ("map", amap)):
    for meth in ("bcr", "band"):
        for delta in (0.1, 0.0):
            for kw in ({}, {"kappa_cap": 1e300}, {"refine": 0}, {"refine": 1}):
                try:
                    u, c, g = s.evaluate(a, delta, fetch_u=False, adjoint_method=meth, **kw)
                    st = s.stats()
                    print("%-8s %-5s reg=%d %-22s scaled %.2e raw %.2e kappa %.3e attempts %d grad %.8g" % (
                        name, meth, st["reg_gradient_used"], kw, st["adjoint_residual"], st["adjoint_residual_raw"],
                        st["kappa_used"], st["adjoint_attempts"], np.sum(g)), flush=True)
                except Exception as e:
                    st = s.stats()
                    print("%-8s %-5s delta=%g %-22s FAILED %s | scaled %.2e attempts %d" % (name, meth, delta, kw, e, st["adjoint_residual"], st["adjoint_attempts"]), flush=True)
s.close()
ub, f = synth_batch(2, 200, 160, seed=5)
s = TVSolver(160, 200, 2); s.set_data(ub, f)
for name, a in (("scalar", 0.1), ("patch22", P22)):
    for delta in (0.1, 0.0):
        u, c, g = s.evaluate(a, delta, fetch_u=False, maxiter=1000)
        st = s.stats()
        print("hbm %-8s reg=%d scaled %.2e raw %.2e kappa %.3e attempts %d" % (name, st["reg_gradient_used"], st["adjoint_residual"], st["adjoint_residual_raw"], st["kappa_used"], st["adjoint_attempts"]), flush=True)
