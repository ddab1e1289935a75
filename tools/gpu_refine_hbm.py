#!/usr/bin/env python3
"""Developer probe: how many refinement sweeps does the HBM-band adjoint need?  (1024^2, pixel map and scalar)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
O, n = 2, 1024
ub, f = synth_batch(O, n, n, seed=3)
jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
amap = 0.11 + 0.09 * np.sin(2 * np.pi * ii / n) * np.cos(2 * np.pi * jj / n)
s = TVSolver(n, n, O); s.set_data(ub, f)
u = s.denoise(amap, maxiter=2000)
ref = {}
for name, a in (("map", amap), ("scalar", 0.1), ("patch", np.array([[0.08, 0.12], [0.1, 0.05]]))):
    if name != "map":
        u = s.denoise(a, maxiter=2000)
    for reg in (False, True):
        g3 = None
        for nref in (4, 3, 2, 1, 0):
            t = time.time(); g = np.asarray(s.gradient(u, ub, a, reg=reg, refine=nref)); dt = time.time() - t
            st = s.stats()
            if g3 is None:
                g3 = g
            print("%-6s reg=%d refine %d: %.3f s scaled res %.2e raw %.2e  sum %.10g  max rel diff vs refine 4: %.2e" % (
                name, reg, nref, dt, st["adjoint_residual"], st["adjoint_residual_raw"], g.sum(),
                np.abs(g - g3).max() / np.abs(g3).max()), flush=True)
