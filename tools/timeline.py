"""Dev tool: per-launch timeline of the last adjoint gradient in a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
ends = [i for i, n in enumerate(names) if 'adj_gradpix' in n]
i1 = ends[-1]
i0 = max(i for i in range(i1) if 'adj_setup' in names[i])
agg = {}
prev_end = None
for r in rows[i0:i1 + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].split('(')[0].replace('bpltv::', '').replace('void ', '')[:30]
    gap = (st - prev_end) / 1e3 if prev_end else 0
    prev_end = en
    if len(sys.argv) > 2:
        print("%-30s grid %-16s dur %8.2f us gap %6.2f" % (nm, r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'] + 'x' + r['Grid_Size_Z'], (en - st) / 1e3, gap))
    a = agg.setdefault(nm, [0, 0.0, 0.0]); a[0] += 1; a[1] += (en - st) / 1e3; a[2] += gap
tot = (int(rows[i1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
for nm, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-30s calls %4d  busy %9.1f us  gaps %7.1f us" % (nm, c, d, g))
print("span %.1f us, busy %.1f us" % (tot, sum(v[1] for v in agg.values())))
