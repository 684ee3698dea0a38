#!/usr/bin/env python3
"""Average duration of the launches of one kernel (name prefix) by grid size from a rocprofv3 kernel_trace CSV:
    python tools/trace_by_grid.py kernel_trace.csv gn_bwd_onepass"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sys.argv[2] not in n:
        continue
    key = (n.split("(")[0].replace("void ", "")[:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""),
           r.get("Workgroup_Size_X", ""))
    rows[key][0] += 1
    rows[key][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(v[1] for v in rows.values())
print("kernel, grid threads x/y/z, block: launches, avg us, share")
for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %8s %6s %6s %5s : %5d  %8.1f us  %5.1f %%" % (*k, c, t / c / 1e3, 100 * t / tot))
print("total %.2f ms" % (tot / 1e6))
