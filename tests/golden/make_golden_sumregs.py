#!/usr/bin/env python3
"""Generate tests/golden/golden_sumregs.npz: expected outputs of the sum-of-regularisers learning function on the
reference's own images (tests/golden/datasets.npz).  PARITY UNPINNED BY THE REFERENCE (absent package, no fixtures):
  * u, cost, gap : oracle/sumregs_oracle.c, cross-checked against the numpy twin oracle/np_twin_sumregs.py;
  * gradients    : the LITERAL sparse systems of /root/reference/src/SumRegsLearningFunction.jl:264-327, :330-407
                   (7n^2 saddle systems, sparse LU + extended-precision refinement) and :112-167, :195-262
                   (gradient_reg, incl. the non-symmetric row-scaled patch system) assembled by np_twin_sumregs.py.
Run in the build container:  python tests/golden/make_golden_sumregs.py
"""
import json, os, sys, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import np_twin as T, np_twin_sumregs as S, c_oracle as co

NPZ = os.path.join(HERE, "datasets.npz")
A3 = np.array([0.03, 0.02, 0.05])
A0 = np.array([0.001, 0.001, 0.001])      # the drivers' start value, /root/reference/src/BPLDenoising.jl:428
P3 = np.stack([np.array([[0.03, 0.05], [0.02, 0.04]]), np.array([[0.02, 0.03], [0.05, 0.02]]),
               np.array([[0.04, 0.02], [0.03, 0.06]])])   # numpy (3, n, m) = Julia 2x2x3
CASES = [("cameraman10_vector", "cameraman_128_10", (0, 1), A3, 5000),
         ("cameraman10_start", "cameraman_128_10", (0, 1), A0, 5000),
         ("cameraman10_patch223", "cameraman_128_10", (0, 1), P3, 5000),
         ("faces_train_vector", "faces_train_128_10", (0, 3), A3, 5000)]


def main():
    out, meta = {}, []
    for name, ds, (lo, hi), alpha, maxiter in CASES:
        ub, f = T.load_dataset(NPZ, ds)
        ub, f = ub[lo:hi], f[lo:hi]
        u, y = co.sumregs_pdhg(f, alpha, maxiter=maxiter, return_dual=True, nthreads=8)
        twin_du = float(np.abs(S.pdhg(f[:1], alpha, maxiter=300) - co.sumregs_pdhg(f[:1], alpha, maxiter=300)).max())
        g_lit = S.batch_gradient(alpha, u, ub, reg=False, refine=10)
        g_reg = S.batch_gradient(alpha, u, ub, reg=True)
        g_c = co.sumregs_gradient(alpha, u, ub)
        g_creg = co.sumregs_gradient(alpha, u, ub, reg=True)
        out[name + "/cost"] = np.float64(co.cost(u, ub))
        out[name + "/gap"] = co.sumregs_gap(u, y, f, alpha)
        out[name + "/grad"] = np.asarray(g_lit)
        out[name + "/grad_reg"] = np.asarray(g_reg)
        out[name + "/u_crc32"] = np.uint32(zlib.crc32(np.ascontiguousarray(u).tobytes()))
        rel = float(np.abs(g_c - g_lit).max() / np.abs(g_lit).max())
        relr = float(np.abs(g_creg - g_reg).max() / np.abs(g_reg).max())
        meta.append(dict(name=name, dataset=ds, lo=lo, hi=hi, maxiter=maxiter, alpha=np.asarray(alpha).tolist(),
                         twin_max_du_300=twin_du, c_oracle_vs_literal_grad_rel=rel, c_oracle_vs_literal_grad_reg_rel=relr))
        print(name, "cost", out[name + "/cost"], "gap", out[name + "/gap"].max(), "grad", np.ravel(g_lit)[:4], "C-vs-literal", rel, relr,
              "twin du", twin_du, flush=True)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "golden_sumregs.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "golden_sumregs.npz")), "bytes")


if __name__ == "__main__":
    main()
