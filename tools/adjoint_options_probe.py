import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bpldenoising_amd as B
for O in (1, 10, 40):
    ub, f, _ = bench.load_batch("faces_train_128_10", O, 128, 128, 20211004)
    s = B.TVSolver(128, 128, O); s.set_data(ub, f)
    for opts in ({}, {"nd_skinny": 0}, {"nd_skinny_min": 100000}, {"nd_skinny2_min": 100000}, {"nd_skinny_min": 400, "nd_skinny2_min": 400}, {"nd_skinny_min": 1000, "nd_skinny2_min": 1000}):
        for k in ("nd_wave", "nd_skinny", "nd_staged"): s.set_option(k, opts.get(k, 1))
        for k in ("nd_skinny_min", "nd_skinny2_min"): s.set_option(k, opts.get(k, 0))
        best = 1e9
        for _ in range(6):
            s.evaluate(0.1, 0.1, maxiter=300)
            best = min(best, s.stats()["adjoint_ms"])
        print(O, opts, "adjoint_ms %.3f" % best, flush=True)
    s.close()
