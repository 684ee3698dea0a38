#!/usr/bin/env python3
"""Merge the two per-kernel PMC summaries (tools/pmc_summarize.py on a `--pmc FETCH_SIZE` pass and a `--pmc WRITE_SIZE`
pass of the same bench command) into the conv-family traffic record bench.py's roofline.traffic reads.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json <commit> <out.json>

Units and corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads (x2 below; re-calibrated on every run on the two streaming kernels whose byte counts
are exact: fastweight reads 2 arenas / writes 1, adam reads 4 / writes 3); WRITE_SIZE calibrates exact."""
import json
import sys

ARENA_BYTES = 26_977_504 * 4          # SURVEY 8a row 1, padded spans included (the arena the streaming kernels walk)


def fam(name):
    return name.startswith("igemm_mfma_kernel") or name.startswith("igemm_k4_") or name.startswith("igemm_tp_kernel")


def main(fetch_json, write_json, commit, out, seqs=1, csrc_sha16=None):
    seqs = int(seqs)                      # sequences per launch of the profiled command: the streaming kernels move seqs arenas
    F, W = json.load(open(fetch_json))["kernels"], json.load(open(write_json))["kernels"]
    cal = {}
    for k, (nr, nw) in dict(fastweight_kernel=(2, 1), adam_kernel=(4, 3)).items():
        f = next((v for n, v in F.items() if n.startswith(k)), None)
        w = next((v for n, v in W.items() if n.startswith(k)), None)
        if f and w:
            # replica groups issue an update as THREE launches over consecutive arena ranges (adapt_step.hip weight_update): the byte count that
            # is exact is the one of a whole update, so the average per launch is compared with a third of it
            parts = 3 if (seqs > 1 and f["launches"] % 3 == 0 and f["per_launch"] * 1024 * 2 < 0.6 * nr * ARENA_BYTES * seqs) else 1
            cal[k] = dict(launches_per_update=parts, read_bytes_true=nr * ARENA_BYTES * seqs / parts, fetch_kib=f["per_launch"],
                          fetch_ratio=f["per_launch"] * 1024 * parts / (nr * ARENA_BYTES * seqs),
                          write_bytes_true=nw * ARENA_BYTES * seqs / parts, write_kib=w["per_launch"],
                          write_ratio=w["per_launch"] * 1024 * parts / (nw * ARENA_BYTES * seqs))
    per, n, rd, wr = {}, 0, 0.0, 0.0
    for name, f in F.items():
        if not fam(name):
            continue
        w = W.get(name, dict(per_launch=0.0, launches=f["launches"]))
        per[name] = dict(launches=f["launches"], read_bytes=f["per_launch"] * 1024 * 2, write_bytes=w["per_launch"] * 1024)
        n += f["launches"]
        rd += f["total"] * 1024 * 2
        wr += w.get("total", w["per_launch"] * f["launches"]) * 1024
    res = dict(source="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no tracing) on "
                      "`bench.py --seqs %d --steps 2 --warmup 1 --no_roofline --no_sub_records --no_cpu_baseline --percentile_frames 0`" % seqs,
               commit=commit, csrc_sha16=csrc_sha16, calibration=cal,
               kernel="every conv instantiation on the path (igemm_tp_kernel<...>, igemm_mfma_kernel<...>, igemm_k4_fwd_kernel, igemm_k4_dgrad_kernel)",
               launches=n, hbm_read_bytes_per_launch=rd / max(n, 1), hbm_write_bytes_per_launch=wr / max(n, 1),
               hbm_bytes_per_launch=(rd + wr) / max(n, 1), per_variant=per)
    json.dump(res, open(out, "w"), indent=1)
    print("conv family: %d launches, read %.2f MB + write %.2f MB per launch; calibration %s" %
          (n, rd / max(n, 1) / 1e6, wr / max(n, 1) / 1e6, {k: (round(v["fetch_ratio"], 3), round(v["write_ratio"], 3)) for k, v in cal.items()}))


if __name__ == "__main__":
    main(*sys.argv[1:7])
