"""Where the host time of one bilevel_learn run goes (round 4): the warm run of bench.py's outer_loop extra, alone, with
torch's HIP context alive, and with a second handle alive (what bench.py's process looks like).
usage: python tools/outer_profile.py [torch] [handle]  (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bpldenoising_amd as B

ub, f, label = bench.load_batch("faces_train_128_10", 10, 128, 128, 20211004)
if "torch" in sys.argv:
    import torch
    z = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
keep = None
if "handle" in sys.argv:
    keep = B.TVSolver(128, 128, 10)
    keep.set_data(ub, f)
    keep.denoise(0.1, maxiter=5000)
for rep in range(3):
    t0 = time.perf_counter()
    x, u, hist = B.trbox.bilevel_learn((ub, f), B.tv_op_learning_function, 0.1, 0.1, maxiter=20)
    dt = time.perf_counter() - t0
    st = B.learning_function._solver_for(ub, f).stats() if hasattr(B.learning_function, "_solver_for") else {}
    print("run %d: %.1f ms, %d evaluations, last evaluation: pdhg %.2f ms adjoint %.2f ms" % (rep, 1e3 * dt, len(hist) + 1, st.get("pdhg_ms", 0), st.get("adjoint_ms", 0)))
