#!/usr/bin/env python3
"""Developer probe: store/load cache-policy experiments (timing + parity)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10)
s.set_data(ub, f)
ref = s.denoise(0.1, maxiter=5000)
for T in (4, 8):
    for chains in (1, 2):
        for dbg, name in ((0, "plain"), (16, "nt stores"), (32, "sc1 stores"), (64, "nt loads"), (16 + 64, "nt stores+loads"), (32 + 64, "sc1 stores + nt loads")):
            t = []
            u = s.denoise(0.1, maxiter=5000, variant=1, tile_iters=T, chains=chains, dbg=dbg)
            ok = np.array_equal(u, ref)
            for _ in range(4):
                s.denoise(0.1, fetch=False, maxiter=5000, variant=1, tile_iters=T, chains=chains, dbg=dbg)
                st = s.stats(); t.append(st["pdhg_ms"])
            print("T %d chains %d %-24s: %.3f ms (%.0f it/s) bitexact %s" % (T, chains, name, min(t), 5e6 / min(t), ok), flush=True)
