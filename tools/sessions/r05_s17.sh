#!/bin/bash
# r05 s17: batch > 1 (one sequence): latency schedule vs throughput schedule by batch size (tp_batch_min)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s17; O=gpurun_out/s17; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
for B in 4 6 8 12 16; do for tb in 99 $B; do
  DYB_TP_BATCH_MIN=$tb timeout 200 python bench.py --seqs 1 --batch $B --steps 16 --warmup 4 $Q > $O/b_B${B}_$tb.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/b_B${B}_$tb.json").read().strip().splitlines()[-1]); print("B=$B tp_batch_min=$tb", round(d["value"], 1), round(d["ms_per_step"], 2), flush=True)
PY
done; done
