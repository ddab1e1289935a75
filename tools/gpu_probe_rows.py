#!/usr/bin/env python3
"""Developer probe: where does the launch time of pdhg_rows_kernel go on config 5's share (8 x 1024^2, pixel map)?
Needs the EXPERIMENTS build (tools/_bin/libbpltv_exp.so; results are wrong when a switch is set).
Bits: 1 skip state loads, 2 skip stores, 128 no barriers."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BPLTV_LIB_PATH", os.path.join(ROOT, "tools", "_bin", "libbpltv_exp.so"))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
O, N, M, iters = 8, 1024, 1024, 480
rng = np.random.default_rng(0)
ub = rng.random((O, N, M)); f = ub + 0.1 * rng.standard_normal((O, N, M))
amap = np.ascontiguousarray(0.05 + 0.1 * rng.random((N, M)))
s = TVSolver(M, N, O); s.set_data(ub, f)


def run(dbg, **kw):
    p = s.params(**kw)
    p.reserved[3] = dbg
    s._check(s._lib.bpltv_denoise(s._h, amap.ctypes.data_as(C.POINTER(C.c_double)), M, N, C.byref(p), None))
    return s.stats()


for var in (19, 20):
    for dbg, name in ((0, "full"), (128, "no barriers"), (1, "no state loads"), (2, "no stores"), (3, "no loads, no stores"), (131, "no barriers, loads, stores"),
                      (8 << 8, "stagger <= 3.4 us"), (24 << 8, "stagger <= 10 us"), (48 << 8, "stagger <= 20 us"), (96 << 8, "stagger <= 41 us")):
        for ch in (1, 2):
            t = []
            for _ in range(3):
                st = run(dbg, maxiter=iters, variant=var, tile_iters=8, chains=ch)
                t.append(st["pdhg_ms"])
            print("variant %d chains %d %-28s: %.3f ms = %.3e it/s, per dispatch %.1f us" % (var, ch, name, min(t), iters / min(t) * 1e3, 1e3 * min(t) / st["launches"]), flush=True)
    continue
    for dbg, name in ():
        t = []
        for _ in range(3):
            st = run(dbg, maxiter=iters, variant=var, tile_iters=8)
            t.append(st["pdhg_ms"])
        print("variant %d %-28s: %.3f ms = %.3e it/s, per launch %.1f us" % (var, name, min(t), iters / min(t) * 1e3, 1e3 * min(t) / st["launches"]), flush=True)
s.close()
