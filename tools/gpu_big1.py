"""Dev probe: one 1024^2 gradient (HBM band path) for rocprofv3 --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver
from conftest import synth_batch
O = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ub, f = synth_batch(O, 1024, 1024, seed=3)
s = TVSolver(1024, 1024, O); s.set_data(ub, f)
u, c, g = s.evaluate(0.1, 0.1, fetch_u=False, maxiter=500)
st = s.stats()
print("O", O, "adjoint_ms", st["adjoint_ms"], "grad", g, flush=True)
