#!/usr/bin/env python3
"""Developer probe: PDHG kernel variants on config 5's per-GPU share (8 x 1024^2, pixel-map alpha): iterations/s and a
same-bits check against the library's default.  usage: gpu_variants_large.py [shape=OxNxM] [alpha=map|scalar] [variant:T ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
O, N, M, iters, amode = 8, 1024, 1024, 480, "map"
args = []
for a in sys.argv[1:]:
    if a.startswith("shape="):
        O, N, M = (int(x) for x in a[6:].split("x"))
    elif a.startswith("alpha="):
        amode = a[6:]
    else:
        args.append(a)
rng = np.random.default_rng(0)
ub = rng.random((O, N, M)); f = ub + 0.1 * rng.standard_normal((O, N, M))
amap = 0.05 + 0.1 * rng.random((N, M)) if amode == "map" else 0.1
s = TVSolver(M, N, O); s.set_data(ub, f)
ref = s.denoise(amap, maxiter=iters)
print("shape %d x %d x %d, alpha %s" % (O, N, M, amode), flush=True)
cases = [tuple(int(x) for x in a.split(":")) for a in args if not a.startswith("xcd")] or [(0, 0), (13, 8), (13, 6), (19, 8), (20, 8), (21, 8), (22, 8), (23, 8), (19, 6), (21, 6), (21, 10), (23, 10)]
xcds = [int(a[4:]) for a in args if a.startswith("xcd=")] or [None]
for var, T in cases:
  for xcd in xcds:
    kw = {} if xcd is None else {"xcd": xcd}
    best = 1e9
    for _ in range(3):
        s.denoise(amap, maxiter=iters, variant=var, tile_iters=T, fetch=False, **kw)
        best = min(best, s.stats()["pdhg_ms"])
    st = s.stats()
    u = s.denoise(amap, maxiter=iters, variant=var, tile_iters=T, **kw)
    print("variant %2d T %2d xcd %s (region %dx%d, %d tiles): %.2f ms per %d iterations = %.3e it/s  same bits %s" % (
        var, st["tile_iters"], xcd, st["region_i"], st["region_j"], st["tiles"], best, iters, iters / best * 1e3, np.array_equal(u, ref)), flush=True)
s.close()
