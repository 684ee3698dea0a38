#!/bin/bash
# r05 s21: one sequence - grid cap / minimum K-steps of the latency form's split policy
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s21; O=gpurun_out/s21; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
run() { env $3 timeout 200 python bench.py --seqs $2 --steps $4 --warmup 8 $Q > $O/b_$1.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1]); print("$1 S=$2 [$3]", round(d["value"], 1), round(d["ms_per_step"], 3), flush=True)
PY
}
run base 1 X=1 60
run c256 1 DYB_GRID_CAP=256 60
run c384 1 DYB_GRID_CAP=384 60
run c512 1 DYB_GRID_CAP=512 60
run c640 1 DYB_GRID_CAP=640 60
run c768 1 DYB_GRID_CAP=768 60
run c512m4 1 "DYB_GRID_CAP=512 DYB_MIN_STEPS=4" 60
run c384m4 1 "DYB_GRID_CAP=384 DYB_MIN_STEPS=4" 60
run base_b 1 X=1 60
run s4_base 4 X=1 24
run s4_c512 4 DYB_GRID_CAP=512 24
run s4_c2048 4 DYB_GRID_CAP=2048 24

