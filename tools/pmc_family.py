#!/usr/bin/env python3
"""HBM traffic and achieved bandwidth of one kernel family (name prefix) from two rocprofv3 PMC passes and a kernel-stats CSV of the
same bench command:
    python tools/pmc_family.py pmc_FETCH_SIZE.json pmc_WRITE_SIZE.json kernel_stats.csv <prefix> <sequences> out.json
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md), so
reads are doubled - and re-calibrated in the same passes on the two streaming kernels whose byte counts are exact (fastweight reads
2 parameter arenas and writes 1, adam reads 4 and writes 3).  Per kernel: launches, bytes per launch, average duration (kernel
stats of the same command), achieved GB/s = (2 x FETCH + WRITE) / duration."""
import csv
import json
import sys

ARENA_BYTES = 26_977_504 * 4


def main(fetch_json, write_json, stats_csv, prefix, seqs, out):
    seqs = int(seqs)
    F, W = json.load(open(fetch_json))["kernels"], json.load(open(write_json))["kernels"]
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        n = r["Name"].split("(")[0].replace("void ", "")
        dur[n] = float(r["TotalDurationNs"]) / max(int(r["Calls"]), 1)
    cal = {}
    for k, (nr, nw) in dict(fastweight_kernel=(2, 1), adam_kernel=(4, 3)).items():
        f = next((v for n, v in F.items() if n.startswith(k)), None)
        w = next((v for n, v in W.items() if n.startswith(k)), None)
        if f and w:
            # (replica groups issue an update as three launches over consecutive arena ranges: compare the per-launch average with a third)
            parts = 3 if (seqs > 1 and f["launches"] % 3 == 0 and f["per_launch"] * 1024 * 2 < 0.6 * nr * ARENA_BYTES * seqs) else 1
            cal[k] = dict(launches_per_update=parts, fetch_x2_over_true=2 * f["per_launch"] * 1024 * parts / (nr * ARENA_BYTES * seqs),
                          write_over_true=w["per_launch"] * 1024 * parts / (nw * ARENA_BYTES * seqs))
    per, tb, tt = {}, 0.0, 0.0
    for name, f in F.items():
        if not name.startswith(prefix):
            continue
        w = W.get(name, dict(per_launch=0.0))
        rd, wr = f["per_launch"] * 1024 * 2, w["per_launch"] * 1024
        d = dur.get(name)
        per[name] = dict(launches_in_pmc_pass=f["launches"], read_bytes_per_launch=rd, write_bytes_per_launch=wr, avg_duration_us=d / 1e3 if d else None,
                         achieved_GBps=(rd + wr) / d if d else None)
        if d:
            tb += (rd + wr) * f["launches"]
            tt += d * f["launches"]
    res = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) + --kernel-trace --stats of `bench.py --seqs %d`" % seqs,
               family_prefix=prefix, calibration=cal, family_achieved_GBps=tb / tt if tt else None, peak_GBps=8000, per_kernel=per)
    json.dump(res, open(out, "w"), indent=1)
    print("calibration", {k: {a: round(b, 3) for a, b in v.items()} for k, v in cal.items()})
    for n, v in sorted(per.items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches_in_pmc_pass"]):
        print("%-44s launches %5d  read %8.2f MB  write %8.2f MB  %7.1f us  %6.0f GB/s" % (n[:44], v["launches_in_pmc_pass"], v["read_bytes_per_launch"] / 1e6,
              v["write_bytes_per_launch"] / 1e6, v["avg_duration_us"] or 0, v["achieved_GBps"] or 0))
    print("family %s: %.0f GB/s of 8000" % (prefix, res["family_achieved_GBps"] or 0))


if __name__ == "__main__":
    main(*sys.argv[1:7])
