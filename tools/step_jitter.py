"""Per-step PDHG event times of the headline workload (round 4): how much of the mean is jitter?
usage: python tools/step_jitter.py [steps]  (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import bpldenoising_amd as B
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
kw = {"maxiter": 5000}
size, images, sumregs = 128, 10, 0
for a in sys.argv[2:]:
    k, v = a.split("=")
    if k == "size": size = int(v)
    elif k == "sumregs": sumregs = int(v)
    elif k == "images": images = int(v)
    else: kw[k] = int(v)
ub, f, _ = bench.load_batch("faces_train_128_10" if size == 128 else "synthetic", images, size, size, 20211004)
s = B.TVSolver(size, size, images)
alpha = 0.1 if size == 128 else 0.05 + 0.1 * np.random.default_rng(3).random((size, size))
s.set_data(ub, f)
step = (lambda: s.sumregs_denoise(np.array([0.05, 0.03, 0.02]), fetch=False, **kw)) if sumregs else (lambda: s.denoise(alpha, fetch=False, **kw))
for _ in range(3):
    step()
ev, wall, l0, l1 = [], [], [], []
for _ in range(steps):
    t0 = time.perf_counter()
    step()
    wall.append(1e3 * (time.perf_counter() - t0))
    st = s.stats(); ev.append(st["pdhg_ms"]); l0.append(st["launch_host_ms"][0]); l1.append(st["launch_host_ms"][1])
ev, wall = np.array(ev), np.array(wall)
print(kw, size, images, "event ms: min %.3f median %.3f mean %.3f max %.3f" % (ev.min(), np.median(ev), ev.mean(), ev.max()))
print("wall  ms: min %.3f median %.3f mean %.3f max %.3f" % (wall.min(), np.median(wall), wall.mean(), wall.max()))
l0, l1 = np.array(l0), np.array(l1)
slow = ev > 1.1 * ev.min()
print("host ms of hipGraphLaunch, chain 0 / chain 1: fast steps %.3f / %.3f, slow steps %.3f / %.3f" % (l0[~slow].mean(), l1[~slow].mean(), l0[slow].mean() if slow.any() else 0, l1[slow].mean() if slow.any() else 0))
print("slow steps (> 1.1 x min): %d of %d; launches %d" % ((ev > 1.1 * ev.min()).sum(), steps, s.stats()["launches"]))
