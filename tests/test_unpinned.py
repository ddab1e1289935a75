"""What the UNPINNED choices of the PDHG restatement can change (tools/unpinned_study.py).

The reference's loop lives in an absent, un-versioned package and its tests hold no expected values
(/root/reference/test/runtests.jl:1-6), so the oracle's x0 = f, primal-first ordering, L = sqrt(8) and Newton-rsqrt
projection are choices.  The committed study (tests/golden/unpinned_study.json) ran each choice flipped for the
reference's 5000 iterations on the reference's images; this test re-runs a part of it and checks the invariants the
DESIGN.md table states: every variant is within its own certificate sqrt(2*gap) of the minimiser, variants differ
by <= 1e-4 in u and the learned parameter moves by < 1 %."""
import json
import os
import numpy as np
from oracle import np_twin as T
from conftest import DATASETS_NPZ, GOLDEN


def test_committed_study_invariants():
    st = json.load(open(os.path.join(GOLDEN, "unpinned_study.json")))
    assert len(st["studies"]) == 3
    for s in st["studies"]:
        assert s["maxiter"] == 5000 and s["ustar_gap"] < 1e-8
        base = np.asarray(s["oracle_learned"], dtype=float)
        for r in s["rows"]:
            assert r["dist_to_ustar_max"] <= r["certificate_sqrt_2gap"] + 2 * np.sqrt(2 * s["ustar_gap"])
            assert r["max_abs_du"] <= 1e-4 and r["dcost_rel"] <= 5e-4
            assert np.abs(np.asarray(r["learned"], dtype=float) - base).max() <= 1e-2 * np.abs(base).max()
        # the arithmetic-only variants (projection form) do not move u at all beyond rounding
        for r in s["rows"]:
            if r["flags"] in (0, 4, 8) and abs(r["L"] - np.sqrt(8.0)) < 1e-12:
                assert r["max_abs_du"] < 1e-13


def test_variants_against_a_certified_minimiser(oracle):
    ub, f = T.load_dataset(DATASETS_NPZ, "cameraman_128_10")
    alpha = 0.1
    us, y1, y2 = oracle.pdhg_variant(f, alpha, maxiter=60000, flags=0, return_dual=True)
    gs = float(oracle.gap(us, y1, y2, f, alpha).max())
    assert 0 <= gs < 1e-7                                   # ||us - u*||_2 <= sqrt(2 gs) < 5e-4
    u0 = oracle.pdhg(f, alpha, maxiter=5000)
    rows = {r["flags"]: r for r in json.load(open(os.path.join(GOLDEN, "unpinned_study.json")))["studies"][0]["rows"]
            if abs(r["L"] - np.sqrt(8.0)) < 1e-12}
    for flags in (0, 1, 2, 1 | 2 | 4):
        u, y1, y2 = oracle.pdhg_variant(f, alpha, maxiter=5000, flags=flags, return_dual=True)
        gap = float(oracle.gap(u, y1, y2, f, alpha).max())
        dist = float(np.sqrt(((u - us) ** 2).sum()))
        assert dist <= np.sqrt(2 * gap) + np.sqrt(2 * gs)    # strong convexity: 0.5||u - u*||^2 <= gap
        assert np.isclose(np.abs(u - u0).max(), rows[flags]["max_abs_du"], rtol=1e-6, atol=1e-15)   # the committed numbers
    # the product's arithmetic (fused multiply-adds, Newton rsqrt) == the plain restatement to rounding
    assert np.abs(oracle.pdhg_variant(f, alpha, maxiter=5000, flags=0) - u0).max() < 1e-13
