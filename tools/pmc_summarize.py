#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc counter_collection CSV into per-kernel-family averages (per launch)."""
import csv
import json
import sys
from collections import defaultdict


def main(path, counter, out):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    res = {k: dict(launches=c, total=t, per_launch=t / c) for k, (c, t) in agg.items()}
    json.dump(dict(counter=counter, kernels=res), open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["total"])[:12]:
        print(counter, k[:50], v["launches"], round(v["per_launch"], 1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
