#!/usr/bin/env python3
"""Developer probe (not part of the product): parity + timing sweep of the PDHG kernel variants on
one GPU.  Usage: python tools/gpu_probe.py [--quick]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bpldenoising_amd import TVSolver
from oracle import c_oracle as co


def synth(O, N, M, seed=20211004):
    rng = np.random.default_rng(seed)
    jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
    ub = np.zeros((O, N, M))
    for k in range(O):
        img = 0.3 + 0.4 * (ii / M) * rng.random() + 0.2 * (jj / N) * rng.random()
        for _ in range(6):
            ci, cj, r = rng.random() * M, rng.random() * N, (0.05 + 0.2 * rng.random()) * min(M, N)
            img = np.where((ii - ci) ** 2 + (jj - cj) ** 2 < r * r, rng.random(), img)
        ub[k] = img
    f = np.round(255 * np.clip(ub + 0.1 * rng.standard_normal(ub.shape), 0, 1)) / 255
    return np.clip(ub, 0, 1), f


def main():
    quick = "--quick" in sys.argv
    out = {}
    # ---- parity, small
    O, N, M = 3, 70, 50
    ub, f = synth(O, N, M, 1)
    s = TVSolver(M, N, O)
    s.set_data(ub, f)
    amap = 0.05 + 0.1 * np.random.default_rng(2).random((N, M))
    for name, alpha in (("scalar", 0.1), ("patch", np.array([[0.05, 0.1], [0.2, 0.08]])), ("map", amap)):
        u0 = co.pdhg(f, alpha, maxiter=203)
        for var in ((1, 2, 6, 9) if quick else range(1, 11)):
            for T in (1, 3, 4, 8):
                try:
                    u = s.denoise(alpha, maxiter=203, variant=var, tile_iters=T)
                except Exception as e:
                    print("variant", var, "T", T, "ERR", e); continue
                d = np.abs(u - u0).max()
                if d != 0.0:
                    print("PARITY", name, "variant", var, "T", T, "max|du|", d, "bitexact", np.array_equal(u, u0))
        print("parity sweep done for", name, flush=True)
    u, cost, grad = s.evaluate(0.1, 0.1, maxiter=203)
    u0 = co.pdhg(f, 0.1, maxiter=203)
    print("evaluate: du", np.abs(u - u0).max(), "cost", cost, co.cost(u0, ub), "grad", grad, co.gradient(0.1, u0, ub), s.stats())
    s.close()
    # ---- timing on the headline batch
    O, N, M = 10, 128, 128
    ub, f = synth(O, N, M)
    s = TVSolver(M, N, O)
    s.set_data(ub, f)
    res = []
    u_ref = None
    for var in (1, 3, 4, 5, 7, 10):
        for T in (4, 5, 6, 7, 8, 10):
            for chains in (1, 2):
                try:
                    u = s.denoise(0.1, fetch=True, maxiter=5000, variant=var, tile_iters=T, chains=chains)
                    if u_ref is None:
                        u_ref = u
                    elif not np.array_equal(u, u_ref):
                        print("MISMATCH between configurations", var, T, chains, np.abs(u - u_ref).max())
                    t = []
                    for _ in range(3):
                        s.denoise(0.1, fetch=False, maxiter=5000, variant=var, tile_iters=T, chains=chains)
                        st = s.stats(); t.append(st["pdhg_ms"])
                    res.append((var, T, chains, min(t), st["tiles"], st["launches"], st["total_ms"]))
                    print("var %2d T %d chains %2d: pdhg %.3f ms  (%.0f it/s)  tiles %d launches %d wall %.3f" % (
                        var, st["tile_iters"], chains, min(t), 5000 / min(t) * 1e3, st["tiles"], st["launches"], st["total_ms"]), flush=True)
                except Exception as e:
                    print("var", var, "T", T, "ERR", e, flush=True)
    u0 = co.pdhg(f, 0.1, maxiter=5000, nthreads=8)
    print("10x128x128 5000 it vs oracle: bitexact", np.array_equal(u_ref, u0), "max|du|", np.abs(u_ref - u0).max())
    t0 = time.time(); u, cost, grad = s.evaluate(0.1, 0.1); t1 = time.time()
    print("evaluate 10x128x128:", t1 - t0, "s", s.stats())
    s.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"))


if __name__ == "__main__":
    main()
