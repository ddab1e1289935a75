"""Dev probe: repeated evaluates (timing stability per dataset)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bpldenoising_amd.learning_function import TVSolver
from oracle import np_twin as nt
for name in ("cameraman_128_10", "faces_train_128_10"):
    ub, f = nt.load_dataset(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/datasets.npz"), name, 10)
    s = TVSolver(128, 128, ub.shape[0]); s.set_data(ub, f)
    for it in range(4):
        u, c, g = s.evaluate(0.1, 0.1, fetch_u=False)
        st = s.stats()
        print(name, it, "pdhg %.3f adjoint %.3f total %.3f res %.2e grad %.10e" % (st["pdhg_ms"], st["adjoint_ms"], st["total_ms"], st["adjoint_residual"], float(g)), flush=True)
