#!/usr/bin/env python3
"""What can the UNPINNED choices of the PDHG restatement change?  (CPU only)

The reference's hot loop (`op_denoise_pdps`) lives in a package that is not on disk and has no pinned version;
its own tests hold no expected values (/root/reference/test/runtests.jl:1-6).  The oracle therefore *chooses*:
x0 = f, primal step first, omega from the current tau, L = sqrt(8), projection by a Newton rsqrt.  This script
runs the oracle's recurrence with each choice flipped (oracle/bpltv_oracle.c: bplo_pdhg_variant) for the
reference's 5000 iterations on the reference's own images and records

    max|du|, ||du||_2        against the oracle's u_5000
    dcost/cost, dgrad/grad   of the learning function (loss and adjoint gradient on that u)
    ||u_5000 - u*||_2 and its certificate sqrt(2*gap_5000)   (u* = 200 000-iteration solve, gap ~1e-12)
    the parameter learned by the full outer loop (trbox.bilevel_learn, maxiter 20) with that variant inside

Output: tests/golden/unpinned_study.json (committed; tests/test_unpinned.py re-checks a part of it) and the
markdown table for DESIGN.md section 2 on stdout.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle as co
from oracle import np_twin as T
from bpldenoising_amd import trbox

NPZ = os.path.join(ROOT, "tests", "golden", "datasets.npz")
L8 = float(np.sqrt(8.0))
VARIANTS = [
    ("oracle recurrence (unfused arithmetic)", 0, L8),
    ("x0 = 0 instead of f", 1, L8),
    ("dual step first", 2, L8),
    ("projection alpha/sqrt(n2) (IEEE)", 4, L8),
    ("projection y/max(1,|y|/alpha)", 8, L8),
    ("omega from the updated tau", 16, L8),
    ("L = 2*sqrt(2)*(1-1/n)", 0, L8 * (1 - 1.0 / 128)),
    ("x0 = 0, dual first, IEEE projection", 1 | 2 | 4, L8),
]
NTH = min(8, os.cpu_count() or 1)


def lf_variant(flags, L, maxiter):
    def lf(x, ds, delta, **kw):
        ubar, f = ds
        u = co.pdhg_variant(f, x, maxiter=maxiter, flags=flags, L=L, nthreads=NTH)
        return u, co.cost(u, ubar), co.gradient(x, u, ubar, reg=not (delta > 1e-6))
    return lf


def study(dataset, nimg, alpha, x0, delta0, maxiter=5000, outer=True, ustar_iters=200000):
    ub, f = T.load_dataset(NPZ, dataset)
    ub, f = ub[:nimg], f[:nimg]
    u0 = co.pdhg(f, alpha, maxiter=maxiter, nthreads=NTH)
    c0 = co.cost(u0, ub)
    g0 = np.asarray(co.gradient(alpha, u0, ub))
    us, y1s, y2s = co.pdhg_variant(f, alpha, maxiter=ustar_iters, flags=0, L=L8, nthreads=NTH, return_dual=True)
    gap_star = float(co.gap(us, y1s, y2s, f, alpha).max())
    rows = []
    for name, flags, L in VARIANTS:
        t = time.time()
        u, y1, y2 = co.pdhg_variant(f, alpha, maxiter=maxiter, flags=flags, L=L, nthreads=NTH, return_dual=True)
        gap = co.gap(u, y1, y2, f, alpha)
        c = co.cost(u, ub)
        g = np.asarray(co.gradient(alpha, u, ub))
        dist = np.sqrt(((u - us) ** 2).reshape(nimg, -1).sum(1))
        row = {
            "variant": name, "flags": flags, "L": L,
            "max_abs_du": float(np.abs(u - u0).max()),
            "l2_du": float(np.sqrt(((u - u0) ** 2).sum())),
            "dcost_rel": float(abs(c - c0) / abs(c0)),
            "dgrad_rel": float(np.abs(g - g0).max() / np.abs(g0).max()),
            "gap_max": float(gap.max()),
            "dist_to_ustar_max": float(dist.max()),
            "certificate_sqrt_2gap": float(np.sqrt(2 * gap.max())),
        }
        if outer:
            x, _, hist = trbox.bilevel_learn((ub, f), lf_variant(flags, L, maxiter), x0, delta0, maxiter=20, tol=1e-5)
            row["learned"] = np.asarray(x).tolist()
            row["outer_iterations"] = len(hist)
            row["final_cost"] = hist[-1]["function_value"]
        row["seconds"] = round(time.time() - t, 1)
        rows.append(row)
        print("  %-40s max|du| %.2e  dcost %.1e  dgrad %.1e  dist(u*) %.2e <= %.2e  learned %s" % (
            name, row["max_abs_du"], row["dcost_rel"], row["dgrad_rel"], row["dist_to_ustar_max"],
            row["certificate_sqrt_2gap"], row.get("learned")), file=sys.stderr, flush=True)
    if outer:   # the oracle itself through the same loop
        xo, _, ho = trbox.bilevel_learn((ub, f), lambda x, ds, d, **kw: co.tv_op_learning_function(x, ds, d, maxiter=maxiter, nthreads=NTH),
                                        x0, delta0, maxiter=20, tol=1e-5)
        learned0 = np.asarray(xo).tolist()
    else:
        learned0 = None
    return {"dataset": dataset, "images": nimg, "alpha": np.asarray(alpha).tolist(), "maxiter": maxiter,
            "oracle_cost": c0, "oracle_grad": g0.tolist(), "oracle_learned": learned0,
            "ustar_iterations": ustar_iters, "ustar_gap": gap_star, "rows": rows}


def table(st):
    out = ["| variant | max\\|Δu\\| | Δcost/cost | Δgrad/grad | ‖u₅₀₀₀−u*‖₂ | √(2·gap) | learned parameter |", "|---|---|---|---|---|---|---|"]
    for r in st["rows"]:
        lp = r.get("learned")
        lp = "—" if lp is None else (("%.6f" % lp) if np.ndim(lp) == 0 else np.array2string(np.asarray(lp), precision=5, separator=", ").replace("\n", ""))
        out.append("| %s | %.1e | %.1e | %.1e | %.2e | %.2e | %s |" % (r["variant"], r["max_abs_du"], r["dcost_rel"], r["dgrad_rel"],
                                                                     r["dist_to_ustar_max"], r["certificate_sqrt_2gap"], lp))
    return "\n".join(out)


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    res = {"note": __doc__.strip().split("\n")[0], "studies": []}
    cases = [("cameraman_128_10", 1, 0.1, 0.1, 0.1),                        # BASELINE config 2's image, scalar alpha
             ("faces_train_128_10", 10, 0.1, 0.1, 0.1),                     # config 3
             ("cameraman_128_10", 1, np.array([[0.08, 0.12], [0.1, 0.05]]), 1e-4 * np.ones((2, 2)), 1e-4)]   # config 4
    for ds, n, a, x0, d0 in cases[:1] if quick else cases:
        print("== %s x%d alpha %s" % (ds, n, np.asarray(a).tolist()), file=sys.stderr, flush=True)
        st = study(ds, n, a, x0, d0, outer=not quick, ustar_iters=20000 if quick else 200000)
        res["studies"].append(st)
        print("\n**%s, %d image(s), α = %s** (oracle: cost %.6f, learned %s; u*: %d iterations, gap %.1e)\n" % (
            ds, n, np.asarray(a).tolist(), st["oracle_cost"], st["oracle_learned"], st["ustar_iterations"], st["ustar_gap"]))
        print(table(st))
    if not quick:
        json.dump(res, open(os.path.join(ROOT, "tests", "golden", "unpinned_study.json"), "w"), indent=1)
