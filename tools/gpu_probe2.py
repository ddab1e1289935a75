#!/usr/bin/env python3
"""Developer probe: where does the fixed per-launch time of pdhg_tile_kernel go?

Needs the EXPERIMENTS build of the library (timing switches compiled in; results are wrong when set):
    python -c "import __graft_entry__ as g; g.build_experiments()"     # -> tools/_bin/libbpltv_exp.so
The product library is compiled without these switches and rejects params.reserved[3] != 0.
Bits: 1 skip state loads, 2 skip stores, 4 no iterations, 16 nt stores, 32 plain stores, 64 nt loads."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BPLTV_LIB_PATH", os.path.join(ROOT, "tools", "_bin", "libbpltv_exp.so"))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10)
s.set_data(ub, f)
a = np.array([0.1])


def run(dbg, **kw):
    p = s.params(**kw)
    p.reserved[3] = dbg
    s._check(s._lib.bpltv_denoise(s._h, a.ctypes.data_as(C.POINTER(C.c_double)), 1, 1, C.byref(p), None))
    return s.stats()


for T in (4, 8):
    for dbg, name in ((0, "full"), (4, "no-iterations"), (5, "no-iter,no-state-loads"), (6, "no-iter,no-stores"),
                      (7, "no-iter,no-loads,no-stores"), (1, "iter,no-state-loads"), (2, "iter,no-stores"),
                      (3, "iter,no-loads,no-stores"), (16, "nt stores"), (32, "plain stores"), (64, "nt loads")):
        t = []
        for _ in range(4):
            st = run(dbg, maxiter=5000, variant=1, tile_iters=T)
            t.append(st["pdhg_ms"])
        print("T %d %-28s: %.3f ms  per-launch %.2f us" % (T, name, min(t), 1e3 * min(t) / st["launches"]), flush=True)
