"""Dev probe: two evaluates on faces_train (10 x 128^2) for counter collection."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bpldenoising_amd.learning_function import TVSolver
from oracle import np_twin as nt
ub, f = nt.load_dataset(os.path.join(ROOT, "tests/golden/datasets.npz"), "faces_train_128_10", 10)
s = TVSolver(128, 128, 10); s.set_data(ub, f)
for it in range(2):
    u, c, g = s.evaluate(0.1, 0.1, fetch_u=False, maxiter=200)
    print(it, s.stats()["adjoint_ms"], float(g), flush=True)
