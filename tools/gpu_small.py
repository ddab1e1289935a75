"""Dev probe: adjoint paths on tiny images."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bpldenoising_amd.learning_function import TVSolver
from oracle import c_oracle as co
from tests.conftest import synth_batch
for M in (1, 2, 3, 4, 5, 7, 8, 9, 12):
    for N in (1, 2, 3, 5):
        ub, f = synth_batch(2, N, M, seed=70 + N + M)
        s = TVSolver(M, N, 2); s.set_data(ub, f)
        u0 = co.pdhg(f, 0.1, maxiter=400); g0 = co.gradient(0.1, u0, ub)
        line = "M=%d N=%d oracle %.6e" % (M, N, g0)
        for meth in ("band", "bcr"):
            try:
                _, _, g = s.evaluate(0.1, 0.1, maxiter=400, adjoint_method=meth)
                line += "  %s rel %.1e" % (meth, abs(g - g0) / max(abs(g0), 1e-300))
            except Exception as e:
                line += "  %s ERR %s" % (meth, str(e)[-60:])
        print(line, flush=True)
        s.close()
