#!/usr/bin/env python3
"""Developer probe: PDHG kernel variants on config 5's per-GPU share (8 x 1024^2, pixel-map alpha): iterations/s and a
same-bits check against the library's default.  usage: gpu_variants_large.py [variant:T ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
O, n, iters = 8, 1024, 480
rng = np.random.default_rng(0)
ub = rng.random((O, n, n)); f = ub + 0.1 * rng.standard_normal((O, n, n))
amap = 0.05 + 0.1 * rng.random((n, n))
s = TVSolver(n, n, O); s.set_data(ub, f)
ref = s.denoise(amap, maxiter=iters)
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(0, 0), (13, 8), (13, 6), (19, 8), (20, 8), (21, 8), (22, 8), (23, 8), (19, 6), (21, 6), (21, 10), (23, 10)]
for var, T in cases:
    best = 1e9
    for _ in range(3):
        s.denoise(amap, maxiter=iters, variant=var, tile_iters=T, fetch=False)
        best = min(best, s.stats()["pdhg_ms"])
    st = s.stats()
    u = s.denoise(amap, maxiter=iters, variant=var, tile_iters=T)
    print("variant %2d T %2d (region %dx%d, %d tiles): %.2f ms per %d iterations = %.3e it/s  same bits %s" % (
        var, st["tile_iters"], st["region_i"], st["region_j"], st["tiles"], best, iters, iters / best * 1e3, np.array_equal(u, ref)), flush=True)
s.close()
