import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bpldenoising_amd as B
ub, f, _ = bench.load_batch("faces_train_128_10", 10, 128, 128, 20211004)
s = B.TVSolver(128, 128, 10); s.set_data(ub, f)
kw = {}
for a_ in sys.argv[1:]:
    k_, v_ = a_.split('='); kw[k_] = int(v_)
print(kw)
for rep in range(2):
    d = []; e = []; a = []
    for _ in range(12):
        s.denoise(0.1, fetch=False, maxiter=5000, **kw); d.append(s.stats()["pdhg_ms"])
    for _ in range(12):
        s.evaluate(0.1, 0.1, fetch_u=False, maxiter=5000, **kw); st = s.stats(); e.append(st["pdhg_ms"]); a.append(st["adjoint_ms"])
    print("evaluate pdhg_ms:", " ".join("%.2f" % x for x in e))
    print("denoise pdhg_ms: median %.3f min %.3f | evaluate pdhg_ms: median %.3f min %.3f, adjoint %.3f, launches %d chains %d" % (np.median(d), min(d), np.median(e), min(e), np.median(a), st["launches"], st["launch_chains"]))
    d = []
    for _ in range(12):
        s.denoise(0.1, fetch=False, maxiter=5000, **kw); d.append(s.stats()["pdhg_ms"])
    print("denoise again: median %.3f" % np.median(d))
import time
for gap_ms in (0.0, 0.5, 1.0, 2.0, 5.0, 20.0):
    d = []
    for _ in range(12):
        if gap_ms: time.sleep(gap_ms * 1e-3)
        s.denoise(0.1, fetch=False, maxiter=5000, **kw); d.append(s.stats()["pdhg_ms"])
    print("denoise after %.1f ms of idle: median %.3f min %.3f" % (gap_ms, np.median(d), min(d)))
