"""Does it matter WHICH hardware queues the two launch-chain streams get?  Creates k dummy HIP streams before the first handle
(shifting the creation-order mapping of streams onto hardware queues) and times the reference batch and config 5's share.
usage: python tools/queue_probe.py [kmax]  (GPU box)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bpldenoising_amd as B

hip = ctypes.CDLL("libamdhip64.so")
kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(3)
amap = 0.05 + 0.1 * rng.random((1024, 1024))
ub1, f1, _ = bench.load_batch("faces_train_128_10", 10, 128, 128, 20211004)
ub5, f5, _ = bench.load_batch("synthetic", 8, 1024, 1024, 20211004)
dummies = []
for k in range(kmax + 1):
    if k > 0:
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        dummies.append(st)
    out = []
    for (O, n, iters, alpha, ub, f) in ((10, 128, 5000, 0.1, ub1, f1), (8, 1024, 2000, amap, ub5, f5)):
        s = B.TVSolver(n, n, O)
        s.set_data(ub, f)
        best = 1e9
        for _ in range(4):
            s.denoise(alpha, maxiter=iters)
            best = min(best, s.stats()["pdhg_ms"])
        out.append(iters / best * 1e3)
        s.close()            # the last handle of the device: its streams go, the next handle creates new ones
    print("dummy streams created before: %d   reference batch %.4g it/s   config-5 share %.4g it/s" % (k, out[0], out[1]), flush=True)
