#!/bin/bash
# r05 s9: in-kernel fold of the throughput kernel per mode (bit 0 forward incl. statistics, bit 1 data gradient, bit 2 weight gradient)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s9; O=gpurun_out/s9; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
run() { # tag env seqs steps warm extra
  env $2 timeout 300 python bench.py --seqs $3 --steps $4 --warmup $5 $Q $6 > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1])
    print("$1 [$2] S=$3 $6:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step", flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-1500:])
PY
}
for rep in a b; do
  run f0$rep DYB_TP_FOLD=0 32 12 3
  run f1$rep DYB_TP_FOLD=1 32 12 3
  run f4$rep DYB_TP_FOLD=4 32 12 3
  run f5$rep DYB_TP_FOLD=5 32 12 3
done
run s16_f0 DYB_TP_FOLD=0 16 16 4
run s16_f1 DYB_TP_FOLD=1 16 16 4
