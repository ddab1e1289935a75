#!/usr/bin/env python3
"""Developer probe (EXPERIMENTS build, tools/_bin/libbpltv_exp.so): where do the launches of the two chains of a solve lie
relative to each other in a fast step and in a slow one ("two kinds of steps")?  The first workgroup of every
launch stamps the 100 MHz clock (params.reserved[3] & 1024).  usage: python tools/chain_phase.py [steps]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BPLTV_LIB_PATH", os.path.join(ROOT, "tools", "_bin", "libbpltv_exp.so"))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from bpldenoising_amd import TVSolver
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
ub, f, _ = bench.load_batch("faces_train_128_10", 10, 128, 128, 20211004)
s = TVSolver(128, 128, 10); s.set_data(ub, f)
a = np.array([0.1])
log = (C.c_longlong * (2 * 4096))()


EVAL = "evaluate" in sys.argv      # every step a whole evaluate (PDHG + loss + adjoint gradient): the PDHG of the next one starts behind an adjoint
_params = s.params


def _params_dbg(*a_, **kw):
    p = _params(*a_, **kw)
    p.reserved[3] = 1024
    return p


def run(dbg):
    if EVAL:
        s.params = _params_dbg
        s.evaluate(0.1, 0.1, fetch_u=False, maxiter=5000)
        s.params = _params
        return s.stats()
    p = s.params(maxiter=5000)
    p.reserved[3] = dbg
    s._check(s._lib.bpltv_denoise(s._h, a.ctypes.data_as(C.POINTER(C.c_double)), 1, 1, C.byref(p), None))
    return s.stats()


for _ in range(3):
    run(1024)
res = []
for k in range(steps):
    st = run(1024)
    assert s._lib.bpltv_debug_tlog(log) == 0
    t = np.array(log[:], dtype=np.int64).reshape(2, 4096) * 0.01      # us
    nl = st["launches"] // 2
    res.append((st["pdhg_ms"], t[0, :nl].copy(), t[1, :nl + 1].copy()))
ms = np.array([r[0] for r in res])
print("steps %d: min %.3f median %.3f max %.3f ms; slow (> 1.1 x min): %d" % (steps, ms.min(), np.median(ms), ms.max(), (ms > 1.1 * ms.min()).sum()))


def describe(tag, r):
    ms_, A, B = r
    t0 = A[0]
    dA, dB = np.diff(A), np.diff(B[1:])
    print("%s step %.3f ms: chain 0 period mean %.2f us (min %.2f max %.2f), chain 1 %.2f us; chain 1 starts %.1f us after chain 0" % (tag, ms_, dA.mean(), dA.min(), dA.max(), dB.mean(), B[0] - t0))
    # phase of chain 1's launch k+1 (full launches) inside chain 0's period, sampled along the sequence
    for k0 in (0, 8, 16, 24, 32, 40, 48, 100, 150, 160, 170, 300, 500, 600):
        k = np.arange(k0, min(k0 + 8, len(A) - 1))
        ph = []
        for kk in k:
            j = np.searchsorted(A, B[kk + 1]) - 1
            if 0 <= j < len(A) - 1:
                ph.append((B[kk + 1] - A[j]) / (A[j + 1] - A[j]))
        print("   launches %3d..: chain 0 starts %s   phase of chain 1 in chain 0's period %s" % (k0, " ".join("%.1f" % (x - t0) for x in A[k0:k0 + 5]), " ".join("%.2f" % x for x in ph)))


fast = min(res, key=lambda r: r[0])
slow = max(res, key=lambda r: r[0])
describe("fast", fast)
if slow[0] > 1.1 * fast[0]:
    describe("slow", slow)
    for r in [r for r in res if r[0] > 1.03 * fast[0]][:6]:
        describe("slow-ish", r)
elif np.median(ms) > 1.03 * fast[0]:
    describe("median", sorted(res, key=lambda r: r[0])[len(res) // 2])
    sl = [r for r in res if r[0] > 1.1 * fast[0]]
    print("start offsets (chain 1 - chain 0, us) of the slow steps: %s" % " ".join("%.1f" % (r[2][0] - r[1][0]) for r in sl))
print("start offsets of the fast steps: %s" % " ".join("%.1f" % (r[2][0] - r[1][0]) for r in res if r[0] <= 1.1 * fast[0]))
s.close()
