#!/bin/bash
# r05 s15: where is the crossover between the latency schedule and the throughput schedule in sequences per launch?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s15; O=gpurun_out/s15; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
for S in 2 3 4 5 6 7; do for tm in 99 $S; do
  DYB_TP_MIN=$tm timeout 200 python bench.py --seqs $S --steps 20 --warmup 4 $Q > $O/b_S${S}_$tm.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/b_S${S}_$tm.json").read().strip().splitlines()[-1]); print("S=$S tp_min=$tm", round(d["value"], 1), round(d["ms_per_step"], 2), flush=True)
PY
done; done
