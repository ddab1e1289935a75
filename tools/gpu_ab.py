#!/usr/bin/env python3
"""Developer probe: interleaved A/B of launch-chain counts in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
ub, f = synth_batch(10, 128, 128, seed=1)
s = TVSolver(128, 128, 10)
s.set_data(ub, f)
res = {1: [], 2: []}
for rep in range(12):
    for ch in (1, 2):
        s.denoise(0.1, fetch=False, maxiter=5000, chains=ch)
        res[ch].append(s.stats()["pdhg_ms"])
for ch in (1, 2):
    a = np.array(res[ch][2:])
    print("chains %d: median %.3f ms min %.3f max %.3f -> %.0f it/s" % (ch, np.median(a), a.min(), a.max(), 5e6 / np.median(a)))
