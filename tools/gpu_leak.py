"""Dev probe: create / evaluate / destroy handles repeatedly; free HBM must come back."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bpldenoising_amd.learning_function import TVSolver
from tests.conftest import synth_batch
def free(): return torch.cuda.mem_get_info()[0] / 2**20
for (O, N, M, reps) in ((4, 128, 128, 5), (4, 128, 128, 25), (4, 128, 128, 25), (1, 40, 200, 10), (2, 30, 133, 10)):
    ub, f = synth_batch(O, N, M, seed=1)
    f0 = free()
    for r in range(reps):
        s = TVSolver(M, N, O); s.set_data(ub, f)
        s.evaluate(0.1, 0.1, maxiter=50, fetch_u=False)
        s.evaluate(np.array([[0.1, 0.2]]), 0.0, maxiter=50, fetch_u=False)
        s.sweep(np.array([0.05, 0.1]), maxiter=50)
        meth = s.stats()["adjoint_method"]
        s.close()
    print("O %d %dx%d (%s): free HBM before %.0f MiB, after %d create/evaluate/destroy cycles %.0f MiB" % (O, M, N, meth, f0, reps, free()), flush=True)
