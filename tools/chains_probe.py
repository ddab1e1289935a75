"""Launch chains 1..5 on the reference batch and on config 5's share (round 4: streams are now created back to back per device,
so three or four chains get a hardware queue each).  usage: python tools/chains_probe.py  (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bpldenoising_amd as B

def run(O, n, iters, alpha, chains_list, reps=3):
    ub, f, label = bench.load_batch("faces_train_128_10" if n == 128 else "synthetic", O, n, n, 20211004)
    s = B.TVSolver(n, n, O)
    s.set_data(ub, f)
    ref = None
    for ch in chains_list:
        best = 1e9
        for _ in range(reps):
            u = s.denoise(alpha, maxiter=iters, chains=ch)
            best = min(best, s.stats()["pdhg_ms"])
        st = s.stats()
        if ref is None: ref = u
        print("%dx%dx%d chains %d (ran %d): %.3f ms = %.4g it/s  variant %d T %d bitwise %s" % (O, n, n, ch, st["launch_chains"], best, iters / best * 1e3, st.get("variant", -1), st.get("tile_iters", -1), np.array_equal(u, ref)))
    s.close()

run(10, 128, 5000, 0.1, [1, 2, 3, 4, 5])
rng = np.random.default_rng(3)
amap = 0.05 + 0.1 * rng.random((1024, 1024))
run(8, 1024, 2000, amap, [1, 2, 3, 4])
