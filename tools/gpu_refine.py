"""Dev probe: how many refinement sweeps does the adjoint gradient need?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd.learning_function import TVSolver
from oracle import np_twin as nt
for name in ("cameraman_128_10", "faces_train_128_10", "circle_128_10"):
    ub, f = nt.load_dataset(os.path.join(ROOT, "tests/golden/datasets.npz"), name, 10)
    s = TVSolver(128, 128, ub.shape[0]); s.set_data(ub, f)
    for alpha in (0.1, 0.02, np.array([[0.05, 0.1], [0.2, 0.08]])):
        ref = None
        line = "%-20s alpha %-8s" % (name, "patch" if np.ndim(alpha) else alpha)
        for nref in (6, 0, 1, 2, 3):
            _, _, g = s.evaluate(alpha, 0.1, fetch_u=False, refine=nref)
            g = np.asarray(g, dtype=float)
            if ref is None: ref = g
            else: line += "  nref %d: %.1e" % (nref, np.max(np.abs(g - ref) / np.abs(ref)))
        print(line, flush=True)
    s.close()
