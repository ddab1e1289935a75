"""Dev probe: automatic PDHG plan (region / fusion depth) on random shapes and batch sizes vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bpldenoising_amd.learning_function import TVSolver
from oracle import c_oracle as co
from tests.conftest import synth_batch
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    O = int(rng.choice([rng.integers(1, 12), rng.integers(12, 48)]))
    M = int(rng.integers(3, 300)); N = int(rng.integers(3, 300))
    if M * N * O > 1.2e6: N = max(3, int(1.2e6 / (M * O)))
    ub, f = synth_batch(O, N, M, seed=int(rng.integers(1 << 30)))
    mode = int(rng.integers(0, 3))
    alpha = float(rng.uniform(0.02, 0.3)) if mode == 0 else (rng.uniform(0.02, 0.3, size=(2, 3)) if mode == 1 else rng.uniform(0.02, 0.3, size=(N, M)))
    mi = int(rng.integers(20, 140))
    s = TVSolver(M, N, O); s.set_data(ub, f)
    u = s.denoise(alpha, maxiter=mi)
    st = s.stats()
    ok = np.array_equal(u, co.pdhg(f, alpha, maxiter=mi, nthreads=8))
    bad += (not ok)
    print("%s O=%2d %3dx%-3d mode %d it %3d: T %2d tiles %5d" % ("ok  " if ok else "FAIL", O, M, N, mode, mi, st["tile_iters"], st["tiles"]), flush=True)
    s.close()
print("failures:", bad)
