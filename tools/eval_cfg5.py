#!/usr/bin/env python3
"""Developer probe: BASELINE config 5's share of one GPU through the learning function -- 8 x 1024 x 1024,
pixelwise alpha, PDHG (short) + loss + HBM-band adjoint.  For rocprofv3 --kernel-trace / --pmc and timing.
usage: eval_cfg5.py [images] [repeats] [maxiter] [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
O = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 400
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
ub, f = synth_batch(O, n, n, seed=3)
jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
amap = 0.11 + 0.09 * np.sin(2 * np.pi * ii / n) * np.cos(2 * np.pi * jj / n)
s = TVSolver(n, n, O); s.set_data(ub, f)
for r in range(reps):
    t = time.time(); u, c, g = s.evaluate(amap, 0.1, fetch_u=False, maxiter=maxiter); dt = time.time() - t
    st = s.stats()
    print("O %d %dx%d map: evaluate %.3f s  pdhg %.1f ms adjoint %.1f ms residual %.2e (raw %.2e) cost %.4f grad sum %.6f" % (
        O, n, n, dt, st["pdhg_ms"], st["adjoint_ms"], st["adjoint_residual"], st["adjoint_residual_raw"], c, g.sum()), flush=True)
s.close()
