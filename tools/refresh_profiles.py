#!/usr/bin/env python3
"""Copy the summaries of the last tools/prof1.sh run (gpurun_out/) into profiles/ (tracked)."""
import csv, glob, json, os, shutil, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out"); PRO = os.path.join(ROOT, "profiles")

def newest(pattern):
    fs = glob.glob(os.path.join(OUT, pattern))
    return max(fs, key=os.path.getmtime) if fs else None

def mean_counters(path, kernel_sub):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        if kernel_sub in r["Kernel_Name"]:
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}

tag = os.environ.get("PROFILE_TAG", "r01")
ks = newest("prof_kt/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(PRO, "%s_kernel_stats_bench.csv" % tag))
ev = newest("prof_eval/*/*kernel_stats.csv")
if ev:
    shutil.copy(ev, os.path.join(PRO, "%s_kernel_stats_evaluate_v4_bcr.csv" % tag))
p1 = newest("prof_pmc1/*/*counter_collection.csv")
if p1:
    m = mean_counters(p1, "pdhg_tile_kernel")
    with open(os.path.join(PRO, "%s_pmc_sq_pdhg_summary.csv" % tag), "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch\n")
        for (k, c), (n, v) in sorted(m.items()):
            f.write('"%s",%s,%d,%.1f\n' % (k, c, n, v))
p2, p3 = newest("prof_pmc2/*/*counter_collection.csv"), newest("prof_pmc3/*/*counter_collection.csv")
if p2 and p3:
    fe = [v for (k, c), v in mean_counters(p2, "pdhg_tile_kernel").items() if c == "FETCH_SIZE"][0][1]
    wr = [v for (k, c), v in mean_counters(p3, "pdhg_tile_kernel").items() if c == "WRITE_SIZE"][0][1]
    tf = os.path.join(PRO, "traffic.json")
    t = json.load(open(tf))
    t["FETCH_SIZE_KiB_per_10_image_launch"] = fe
    t["WRITE_SIZE_KiB_per_10_image_launch"] = wr
    t["hbm_bytes_per_10_image_launch"] = t["hbm_bytes_per_launch"] = (2 * fe + wr) * 1024
    json.dump(t, open(tf, "w"), indent=1)
bl = os.path.join(OUT, "bench1.log")
if os.path.exists(bl):
    line = [l for l in open(bl) if l.startswith("{")][-1]
    open(os.path.join(PRO, "%s_bench_line.json" % tag), "w").write(line)
    b = json.loads(line)
    print("bench:", b["value"], b["roofline"]["frac"], b["roofline"]["avg_launch_us"], b["cpu_baseline"])
if ks:
    for r in csv.DictReader(open(ks)):
        if "pdhg_tile_kernel" in r["Name"]:
            print("rocprof:", r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
