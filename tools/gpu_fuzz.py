"""Dev probe: random shapes through evaluate (all adjoint paths) against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bpldenoising_amd.learning_function import TVSolver
from oracle import c_oracle as co
from tests.conftest import synth_batch
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    M = int(rng.choice([rng.integers(1, 20), rng.integers(20, 129), rng.integers(129, 139), rng.integers(139, 210)]))
    N = int(rng.choice([rng.integers(1, 6), rng.integers(6, 40)]))
    O = int(rng.integers(1, 4))
    ub, f = synth_batch(O, N, M, seed=int(rng.integers(1 << 30)))
    mode = int(rng.integers(0, 2))
    alpha = float(rng.uniform(0.02, 0.3)) if mode == 0 else rng.uniform(0.02, 0.3, size=(int(rng.integers(1, min(3, N) + 1)), int(rng.integers(1, min(3, M) + 1))))
    delta = float(rng.choice([0.1, 0.0]))
    mi = int(rng.integers(100, 400))
    s = TVSolver(M, N, O); s.set_data(ub, f)
    try:
        u, c, g = s.evaluate(alpha, delta, maxiter=mi)
        meth = s.stats()["adjoint_method"]
        u0 = co.pdhg(f, alpha, maxiter=mi)
        g0 = co.gradient(alpha, u0, ub, reg=(delta <= 1e-6))
        ok = np.array_equal(u, u0) and np.allclose(g, g0, rtol=2e-6, atol=1e-9)
        err = float(np.max(np.abs(np.asarray(g) - np.asarray(g0)) / (np.abs(np.asarray(g0)) + 1e-9)))
    except Exception as e:
        ok, meth, err = False, "EXC " + str(e)[-70:], -1
    if not ok:
        bad += 1
    print("%s M=%3d N=%2d O=%d mode=%d delta=%.1f it=%3d %-8s relerr %.1e" % ("ok  " if ok else "FAIL", M, N, O, mode, delta, mi, meth, err), flush=True)
    s.close()
print("failures:", bad)
