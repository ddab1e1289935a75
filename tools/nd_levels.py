"""Per-level breakdown of the last factorisation in a rocprofv3 kernel trace of `tools/_bin/nd_unit time M nimg`
(rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- tools/_bin/nd_unit time 1024 8).  usage: python tools/nd_levels.py DIR/t_kernel_trace.csv"""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        nm = r['Kernel_Name'].replace('bpltv::', '').split('(')[0].replace('void ', '')
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm, int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])))
rows.sort()
first = [i for i, r in enumerate(rows) if r[2].startswith('nd_front') and (i == 0 or rows[i - 1][2].startswith(('nd_bwd', '__amd')) or not rows[i - 1][2].startswith('nd_'))]
i0 = first[-1]
seq = []
for r in rows[i0:]:
    if r[2].startswith('nd_fwd'):
        break
    seq.append(r)
print("factorisation: %d kernels, %.2f ms" % (len(seq), (seq[-1][1] - seq[0][0]) / 1e6))
lv, cur = [], []
for r in seq:
    if r[2].startswith('nd_gather') or r[2].startswith('nd_front'):
        if cur: lv.append(cur)
        cur = []
    cur.append(r)
lv.append(cur)
tot = {}
for L in lv:
    by = {}
    for r in L:
        by[r[2]] = by.get(r[2], 0) + (r[1] - r[0]) / 1e3
        tot[r[2]] = tot.get(r[2], 0) + (r[1] - r[0]) / 1e3
    print("%8.1f us  %2d kernels  fronts %5d x %d  " % ((L[-1][1] - L[0][0]) / 1e3, len(L), L[0][3], L[0][4]) + "  ".join("%s %.0f" % (k.replace('_kernel', ''), v) for k, v in by.items()))
print("totals: " + "  ".join("%s %.0f" % (k.replace('_kernel', ''), v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
