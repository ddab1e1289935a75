#!/usr/bin/env python3
"""Developer probe: random shapes through evaluate() against the oracle's gradient -- every adjoint path (nested
dissection = the automatic choice, block cyclic reduction, LDS band, HBM band incl. twisted / odd bandwidth / ragged
last panel), both branches, all parameter kinds, with and without forced image groups.
usage: gpu_fuzz_adjoint.py [cases] [seed] [all-kernels]   (all-kernels: nd_front_skinny2_kernel on every eligible level, whatever the batch size)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from oracle import c_oracle as co
from conftest import synth_batch

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
bad = 0
t0 = time.time()
for case in range(cases):
    kind = rng.choice(["bcr", "lds", "hbm", "hbm", "hbm"])
    if kind == "bcr":
        M, N = int(rng.integers(3, 129)), int(rng.integers(2, 70))
    elif kind == "lds":
        M, N = int(rng.integers(129, 139)), int(rng.integers(1, 40))
    else:
        M, N = int(rng.integers(139, 420)), int(rng.integers(1, 48))
    O = int(rng.integers(1, 4))
    ak = rng.choice(["scalar", "patch", "map"])
    if ak == "scalar":
        alpha = float(0.03 + 0.2 * rng.random())
    elif ak == "patch":
        alpha = 0.03 + 0.2 * rng.random((int(rng.integers(1, min(N, 3) + 1)), int(rng.integers(1, min(M, 3) + 1))))
    else:
        alpha = 0.03 + 0.2 * rng.random((N, M))
    reg = bool(rng.integers(0, 2))
    ub, f = synth_batch(O, N, M, seed=int(rng.integers(1, 10000)))
    meth = str(rng.choice(["auto", "auto", "band", "bcr" if (M <= 128 and N >= 2) else "nd"]))
    s = TVSolver(M, N, O); s.set_data(ub, f)
    if "all-kernels" in sys.argv: s.set_option("nd_skinny2_min", 0)
    if meth in ("auto", "nd") and O > 1 and rng.random() < 0.3:
        s.set_option("adjoint_budget_mb", 1.3 * 8e-6 * 40 * M * N * max(4, np.log2(M * N)))   # a group of about one image
    try:
        u, c, g = s.evaluate(alpha, 0.0 if reg else 0.1, maxiter=300, adjoint_method=meth)
        st = s.stats()
        u0 = co.pdhg(f, alpha, maxiter=300)
        g0 = co.gradient(alpha, u0, ub, reg=reg)
        g, g0 = np.asarray(g, dtype=float), np.asarray(g0, dtype=float)
        err = np.abs(g - g0).max() / max(np.abs(g0).max(), 1e-300)
        ok = np.array_equal(u, u0) and err < 1e-4 and st["adjoint_residual"] <= 1e-8
        print("%s case %2d %-3s O %d N %3d M %3d %-6s reg %d  method %-9s groups %d res %.1e  grad err %.1e" % (
            "ok " if ok else "BAD", case, kind, O, N, M, ak, reg, st["adjoint_method"], st["adjoint_chunks"], st["adjoint_residual"], err), flush=True)
        bad += not ok
    except Exception as e:
        print("EXC case %d O %d N %d M %d %s reg %d: %s" % (case, O, N, M, ak, reg, e), flush=True)
        bad += 1
    s.close()
print("%d cases, %d bad, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
