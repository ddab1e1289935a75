#!/usr/bin/env python3
"""Developer probe: reduce a rocprofv3 kernel-trace csv to (queue, stream, short kernel name, start, end, workgroups)
rows, gzip'ed -- small enough to bring back from the GPU box for an offline look at stream overlap."""
import csv, gzip, re, sys
src, dst = sys.argv[1], sys.argv[2]
with open(src) as fh, gzip.open(dst, "wt") as out:
    w = csv.writer(out)
    for r in csv.DictReader(fh):
        nm = re.sub(r"^void ", "", r["Kernel_Name"]).replace("bpltv::", "")
        nm = re.sub(r"\(.*$", "", nm)
        w.writerow([r["Queue_Id"], r.get("Stream_Id", ""), nm, r["Start_Timestamp"], r["End_Timestamp"],
                    int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))])
