#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc counter_collection CSV (several counters in one pass) into per-kernel-family sums and print,
per family, each counter per launch and as a fraction of SQ_WAVE_CYCLES when present.
    python tools/pmc_multi.py <counter_collection.csv> <out.json> [name filter]"""
import csv, json, sys
from collections import defaultdict

path, out = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
res = {}
for k, cs in agg.items():
    n = max(cnt[k].values())
    res[k] = dict(launches=n, **{c: v / n for c, v in cs.items()})
json.dump(res, open(out, "w"), indent=1)
key = "SQ_WAVE_CYCLES" if any("SQ_WAVE_CYCLES" in v for v in res.values()) else None
for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get(key, 0) * kv[1]["launches"]) if key else -kv[1]["launches"])[:14]:
    if flt and flt not in k:
        continue
    print(k[:70], "launches", v["launches"])
    for c, x in sorted(v.items()):
        if c == "launches":
            continue
        frac = f"  ({x / v[key]:.3f} of WAVE_CYCLES)" if key and v.get(key) else ""
        print(f"    {c:28s} {x:16.1f}{frac}")
