"""Dev probe: adjoint gradient by block cyclic reduction vs banded Cholesky vs the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpldenoising_amd as B
from bpldenoising_amd.learning_function import TVSolver
from oracle import c_oracle as co, np_twin as nt
from tests.conftest import synth_batch

def run(name, ub, f, alpha, maxiter=2000, oracle=True, delta=0.1):
    O, N, M = f.shape
    s = TVSolver(M, N, O)
    s.set_data(ub, f)
    out = {}
    for meth in ("band", "bcr"):
        try:
            t = time.time()
            u, c, g = s.evaluate(alpha, delta, maxiter=maxiter, adjoint_method=meth)
            t1 = time.time() - t
            u, c, g = s.evaluate(alpha, delta, maxiter=maxiter, adjoint_method=meth)
            st = s.stats()
            out[meth] = (np.array(g, dtype=float), st["adjoint_ms"], st["adjoint_residual"])
        except Exception as e:
            print(name, meth, "ERROR", e); out[meth] = None
    go = None
    if oracle:
        uo = co.pdhg(f, alpha, maxiter=maxiter)
        go = np.array(co.gradient(alpha, uo, ub, reg=(delta <= 1e-6)), dtype=float)
    for meth in ("band", "bcr"):
        if out[meth] is None: continue
        g, ms, res = out[meth]
        line = "%-28s %-5s adjoint %8.3f ms  res %.2e  |g| %.6e" % (name, meth, ms, res, np.linalg.norm(g))
        if go is not None:
            line += "  rel vs oracle %.2e" % (np.linalg.norm(g.ravel() - go.ravel()) / np.linalg.norm(go))
        if meth == "bcr" and out["band"] is not None:
            line += "  rel vs band %.2e" % (np.linalg.norm(g.ravel() - out["band"][0].ravel()) / np.linalg.norm(out["band"][0]))
        print(line, flush=True)

ub, f = synth_batch(2, 18, 14, seed=3); run("synth 2x18x14 scalar", ub, f, 0.1, 500)
ub, f = synth_batch(3, 37, 29, seed=4); run("synth 3x37x29 scalar", ub, f, 0.05, 500)
ub, f = synth_batch(2, 37, 29, seed=4); run("synth 2x37x29 patch", ub, f, 0.05 * np.ones((2, 3)), 500)
ub, f = synth_batch(2, 64, 48, seed=5); run("synth 2x64x48 map", ub, f, 0.05 + 0.01 * np.random.default_rng(0).random((64, 48)), 500)
ub, f = synth_batch(2, 40, 40, seed=6); run("synth 2x40x40 reg", ub, f, 0.1, 500, delta=1e-7)
ub, f = nt.load_dataset(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/datasets.npz"), "cameraman_128_10", 10)
run("cameraman 10x128x128", ub, f, 0.1, 5000)
run("cameraman 10x128x128 patch", ub, f, 0.02 * np.ones((2, 2)), 5000, delta=1e-4)
ub, f = nt.load_dataset(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/datasets.npz"), "faces_train_128_10", 10)
run("faces 10x128x128", ub, f, 0.1, 5000)
run("faces 1x128x128", ub[:1], f[:1], 0.1, 5000, oracle=False)
