#!/usr/bin/env python3
"""profiles/ bookkeeping for tools/prof2.sh.

  refresh_profiles.py aggregate <dir>    ON THE GPU BOX, after the rocprofv3 passes: reduce every raw
        *counter_collection.csv to one row per (kernel, counter) -- dispatches, mean and sum per dispatch -- and
        every *kernel_trace.csv to its *kernel_stats.csv, then delete the raw per-dispatch files (a 40k-launch
        evaluate is ~100 MB of rows; gpurun brings back 64 MiB).
  refresh_profiles.py publish [tag]      IN THE BUILD CONTAINER: copy the summaries of gpurun_out/$PROF_DIR (default prof2) into
        profiles/<tag>_*.csv and rebuild profiles/traffic.json (HBM bytes and VALU wave-instructions per
        pdhg_tile_kernel launch, per workload) with the gfx950 FETCH_SIZE x2 correction of
        /opt/skills/guides/MI355X_MICROARCH.md (section HBM).
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRO = os.path.join(ROOT, "profiles")
SRC = os.path.join(ROOT, "gpurun_out", os.environ.get("PROF_DIR", "prof2"))


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).replace("bpltv::", "")


def aggregate(d):
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        grid = {}
        with open(path) as fh:
            for r in csv.DictReader(fh):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                a = agg[k]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                grid.setdefault(k[0], (r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", ""),
                                       r.get("VGPR_Count", ""), r.get("SGPR_Count", "")))
        out = os.path.join(os.path.dirname(path), "pmc_summary.csv")
        with open(out, "w") as fh:
            fh.write("kernel,counter,dispatches,mean_per_dispatch,sum,grid_size,workgroup_size,lds,vgpr,sgpr\n")
            for (k, c), (n, s) in sorted(agg.items()):
                g = grid[k]
                fh.write('"%s",%s,%d,%.6g,%.6g,%s\n' % (k, c, n, s / n, s, ",".join(str(x) for x in g)))
        os.remove(path)
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        # per-kernel busy time and the gaps in front of it (what a --stats table does not show)
        rows = []
        with open(path) as fh:
            for r in csv.DictReader(fh):
                nm = short(r["Kernel_Name"])
                if nm.startswith("hb2_update"):   # the parts of the trailing update differ by their grid only
                    try:
                        nm += "[%s wg]" % (int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))))
                    except (KeyError, ValueError):
                        pass
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
        rows.sort()
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        prev_end = None
        for st, en, nm in rows:
            a = agg[nm]
            a[0] += 1
            a[1] += (en - st) / 1e3
            if prev_end is not None and st > prev_end:
                a[2] += (st - prev_end) / 1e3
            prev_end = max(prev_end, en) if prev_end else en
        with open(os.path.join(os.path.dirname(path), "timeline_summary.csv"), "w") as fh:
            fh.write("kernel,calls,busy_us,avg_us,idle_gap_before_us\n")
            for nm, (c, b, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fh.write('"%s",%d,%.1f,%.3f,%.1f\n' % (nm, c, b, b / c, g))
            if rows:
                fh.write('"(span of the trace)",%d,%.1f,,\n' % (len(rows), (max(r[1] for r in rows) - rows[0][0]) / 1e3))
        os.remove(path)
    for pat in ("*agent_info.csv", "*_domain_stats.csv"):
        for path in glob.glob(os.path.join(d, "**", pat), recursive=True):
            os.remove(path)


def newest(pattern):
    fs = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def pmc_mean(tagdir, kernel_sub, counter):
    p = newest(os.path.join(tagdir, "**", "pmc_summary.csv"))
    if not p:
        return None
    best = None
    for r in csv.DictReader(open(p)):
        if kernel_sub in r["kernel"] and r["counter"] == counter:
            if best is None or int(r["dispatches"]) > best[0]:
                best = (int(r["dispatches"]), float(r["mean_per_dispatch"]))
    return best[1] if best else None


def publish(tag):
    os.makedirs(PRO, exist_ok=True)
    copied = []
    for sub, name in (("kt_bench", "kernel_stats_bench"), ("kt_cfg5", "kernel_stats_cfg5_pdhg"),
                      ("kt_eval128", "kernel_stats_evaluate_128"), ("kt_evalcfg5", "kernel_stats_cfg5_evaluate"),
                      ("kt_nd", "kernel_stats_nd_unit_1024x8"), ("kt_sumregs", "kernel_stats_sumregs"),
                      ("kt_single", "kernel_stats_single_image")):
        ks = newest(os.path.join(sub, "**", "*kernel_stats.csv"))
        if ks:
            shutil.copy(ks, os.path.join(PRO, "%s_%s.csv" % (tag, name))); copied.append(name)
        tl = newest(os.path.join(sub, "**", "timeline_summary.csv"))
        if tl:
            shutil.copy(tl, os.path.join(PRO, "%s_%s_timeline.csv" % (tag, name.replace("kernel_stats_", "")))); copied.append(name + "_timeline")
    for sub, name in (("pmc_bench_sq1", "pmc_sq_pdhg"), ("pmc_cfg5_sq1", "pmc_sq_cfg5_pdhg"), ("pmc_cfg5_sq2", "pmc_sq2_cfg5_pdhg"), ("pmc_eval128_sq1", "pmc_sq1_evaluate_128"),
                      ("pmc_eval128_sq2", "pmc_sq2_evaluate_128"), ("pmc_evalcfg5_sq1", "pmc_sq1_cfg5_evaluate"),
                      ("pmc_evalcfg5_sq2", "pmc_sq2_cfg5_evaluate"), ("pmc_evalcfg5_fetch", "pmc_fetch_cfg5_evaluate"),
                      ("pmc_evalcfg5_write", "pmc_write_cfg5_evaluate"), ("pmc_bench_fetch", "pmc_fetch_pdhg"),
                      ("pmc_bench_write", "pmc_write_pdhg"), ("pmc_cfg5_fetch", "pmc_fetch_cfg5_pdhg"),
                      ("pmc_cfg5_write", "pmc_write_cfg5_pdhg"), ("pmc_hb_sq", "pmc_hb_lu_unit"),
                      ("pmc_single_sq1", "pmc_sq_single_image"), ("pmc_single_fetch", "pmc_fetch_single_image"),
                      ("pmc_single_write", "pmc_write_single_image"), ("pmc_sumregs_sq1", "pmc_sq_sumregs"),
                      ("pmc_sumregs_sq2", "pmc_sq2_sumregs")):
        p = newest(os.path.join(sub, "**", "pmc_summary.csv"))
        if p:
            shutil.copy(p, os.path.join(PRO, "%s_%s_summary.csv" % (tag, name))); copied.append(name)
    for log, name in (("bench.log", "bench_line"), ("bench_cfg5.log", "bench_line_cfg5"), ("bench_single.log", "bench_line_single"), ("bench_f32.log", "bench_line_f32"),
                      ("bench_cfg5_f32.log", "bench_line_cfg5_f32")):
        p = os.path.join(SRC, log)
        if os.path.exists(p):
            lines = [l for l in open(p) if l.startswith("{")]
            if lines:
                open(os.path.join(PRO, "%s_%s.json" % (tag, name)), "w").write(lines[-1]); copied.append(name)
    for log in ("eval_cfg5.log", "sumregs_time.log", "sumregs_large.log", "eval_128.log", "nd_unit_time.log", "pmc_hb_sq.log"):
        p = os.path.join(SRC, log)
        if os.path.exists(p):
            with open(os.path.join(PRO, "%s_%s" % (tag, log)), "w") as fh:
                fh.writelines(l for l in open(p) if "amdgpu.ids" not in l)
            copied.append(log)
    # traffic.json: per workload, HBM-side bytes and VALU wave-instructions of one pdhg_tile_kernel launch
    tf = os.path.join(PRO, "traffic.json")
    tj = {"workloads": {}}
    if os.path.exists(tf):
        try:
            old = json.load(open(tf))
            if "workloads" in old:
                tj = old
        except Exception:
            pass
    tj["correction"] = ("gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, section HBM); "
                        "WRITE_SIZE exact; both in KiB per dispatch")
    for key, pre, line in (("10x128x128 scalar", "pmc_bench", "bench_line"), ("8x1024x1024 map", "pmc_cfg5", "bench_line_cfg5"),
                           ("1x128x128 scalar", "pmc_single", "bench_line_single")):
        fe = pmc_mean(pre + "_fetch", "pdhg_", "FETCH_SIZE")      # pdhg_tile_kernel or pdhg_rows_kernel: the one with most dispatches
        wr = pmc_mean(pre + "_write", "pdhg_", "WRITE_SIZE")
        vi = pmc_mean(pre + "_sq1", "pdhg_", "SQ_INSTS_VALU")
        bl = os.path.join(PRO, "%s_%s.json" % (tag, line))
        if fe is None or wr is None or not os.path.exists(bl):
            continue
        b = json.loads(open(bl).read())
        tj["workloads"][key] = {
            "round": tag, "tile_iters": b["config"]["tile_iters"], "tiles": b["config"]["tiles_per_launch"],
            "FETCH_SIZE_KiB_per_launch": fe, "WRITE_SIZE_KiB_per_launch": wr,
            "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
            "valu_wave_instructions_per_launch": vi,
            # the counter passes launch the whole grid eagerly; the bench line's dispatch holds 1 / launch_chains of it
            "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"] * b["roofline"].get("launch_chains", 1),
            "source": "tools/prof2.sh / prof3.sh / prof4.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, eager launches: --no-graph) of the bench command",
        }
    json.dump(tj, open(tf, "w"), indent=1)
    print("published:", ", ".join(copied))
    print(json.dumps(tj["workloads"], indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "aggregate":
        aggregate(sys.argv[2])
    elif len(sys.argv) >= 2 and sys.argv[1] == "publish":
        publish(sys.argv[2] if len(sys.argv) > 2 else "r02")
    else:
        print(__doc__)
