#!/usr/bin/env python3
"""Generate tests/golden/golden_v2.npz: expected outputs of the TV learning-function path.

The reference (Julia + un-vendored VariationalImaging) cannot run here, so these vectors come from
this repo's own restatements -- PARITY UNPINNED BY THE REFERENCE:
  * u, cost, gap   : oracle/bpltv_oracle.c ("spec v2" PDHG), cross-checked against the independent
                     numpy restatement oracle/np_twin.py (max|du| recorded per case);
  * gradients      : the *literal* sparse saddle systems of
                     /root/reference/src/TVLearningFunctionVec.jl:98-135, :137-161, :192-215, :219-254
                     assembled by oracle/np_twin.py on that u and solved by sparse LU + extended
                     precision refinement (np_twin.solve_refined), i.e. the exact solution of the
                     reference's formulas to ~1e-8.
Inputs are the reference's own images (tests/golden/datasets.npz).  Run in the build container:
    python tests/golden/make_golden.py
"""
import os, sys, json, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import np_twin as T, c_oracle as co

NPZ = os.path.join(HERE, "datasets.npz")
P22 = np.array([[0.08, 0.12], [0.1, 0.05]])      # numpy (n, m) = Julia 2x2 column major
P21 = np.array([[0.06, 0.15]])                   # Julia 2x1 (generate_2d_cost style alpha = [a;b])

CASES = [
    # name, dataset, image slice, alpha, maxiter, store_u
    ("circle_scalar", "circle_128_10", (0, 1), 0.1, 5000, True),
    ("cameraman10_scalar", "cameraman_128_10", (0, 1), 0.1, 5000, True),
    ("cameraman5_scalar_tvdenoise", "cameraman_128_5", (0, 1), 0.05, 10000, False),
    ("cameraman10_patch22", "cameraman_128_10", (0, 1), P22, 5000, False),
    ("cameraman10_patch21", "cameraman_128_10", (0, 1), P21, 5000, False),
    ("faces_train_scalar", "faces_train_128_10", (0, 10), 0.07, 5000, False),
    ("faces_val_patch22", "faces_val_128_10", (0, 3), P22, 5000, False),
    ("cameraman10_short", "cameraman_128_10", (0, 1), 0.1, 300, True),
]


def main():
    out = {}
    meta = []
    for name, ds, (lo, hi), alpha, maxiter, store_u in CASES:
        ub, f = T.load_dataset(NPZ, ds)
        ub, f = ub[lo:hi], f[lo:hi]
        u, y1, y2 = co.pdhg(f, alpha, maxiter=maxiter, return_dual=True, nthreads=8)
        ut = T.pdhg_denoise(f[:1], alpha, maxiter=maxiter)
        twin_du = float(np.abs(ut - u[:1]).max())
        cost = co.cost(u, ub)
        gap = co.gap(u, y1, y2, f, alpha)
        g_lit = T.batch_gradient(alpha, u, ub, reg=False, refine=10)
        g_reg = T.batch_gradient(alpha, u, ub, reg=True)
        g_c = co.gradient(alpha, u, ub)
        out[name + "/cost"] = np.float64(cost)
        out[name + "/gap"] = gap
        out[name + "/grad"] = np.asarray(g_lit, dtype=np.float64)
        out[name + "/grad_reg"] = np.asarray(g_reg, dtype=np.float64)
        out[name + "/u_crc32"] = np.uint32(zlib.crc32(np.ascontiguousarray(u).tobytes()))
        out[name + "/u_sum"] = np.float64(u.sum())
        if store_u:
            out[name + "/u"] = u
        rel = np.max(np.abs(np.asarray(g_c) - np.asarray(g_lit)) / np.abs(np.asarray(g_lit)))
        meta.append(dict(name=name, dataset=ds, lo=lo, hi=hi, maxiter=maxiter,
                         alpha=np.asarray(alpha).tolist(), twin_max_du=twin_du,
                         c_oracle_vs_literal_grad_rel=float(rel)))
        print(name, "cost", cost, "gap", gap.max(), "grad", g_lit, "C-vs-literal", rel, "twin du", twin_du, flush=True)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **out)
    print("wrote golden_v2.npz", os.path.getsize(os.path.join(HERE, "golden_v2.npz")))


if __name__ == "__main__":
    main()
