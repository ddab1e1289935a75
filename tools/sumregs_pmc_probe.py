#!/usr/bin/env python3
"""Counter probe of the two sum-of-regularisers PDHG kernels (rocprofv3 --pmc; eager launches so that every dispatch is
one kernel): sr_tile_kernel<32,32> on faces_train_128_10 and sr_strip_kernel<3,48,16> on 4 x 256^2, forward solves only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpldenoising_amd import TVSolver, testdataset
from conftest import synth_batch
A3 = np.array([0.03, 0.02, 0.05])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ub, f = testdataset("faces_train", npz=os.path.join(ROOT, "tests/golden/datasets.npz"))
s = TVSolver(128, 128, 10); s.set_data(ub[:10], f[:10])
s.sumregs_denoise(A3, fetch=False, maxiter=iters, variant=1, use_graph=0, chains=1)
st = s.stats()
print("tile : 10x128^2 region %d T %d tiles %d launches %d pdhg %.3f ms" % (st["region_i"], st["tile_iters"], st["tiles"], st["launches"], st["pdhg_ms"]), flush=True)
s.close()
ub, f = synth_batch(4, 256, 256, seed=3)
s = TVSolver(256, 256, 4); s.set_data(ub, f)
s.sumregs_denoise(A3, fetch=False, maxiter=iters, variant=2, use_graph=0, chains=1)
st = s.stats()
print("strip: 4x256^2 region %d T %d tiles %d launches %d pdhg %.3f ms" % (st["region_i"], st["tile_iters"], st["tiles"], st["launches"], st["pdhg_ms"]), flush=True)
s.close()
