#!/bin/bash
# r05 s23: parallel passes (two streams) with 4 vs 8 hardware queues; final A/B of the shipped form
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s23; O=gpurun_out/s23; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
run() { env $2 timeout 300 python bench.py --seqs $5 --steps $3 --warmup 4 $Q $4 > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1]); print("$1 [$2] S=$5 $4:", round(d["value"], 1), round(d["ms_per_step"], 3), flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-1200:])
PY
}
F="--full_losses 1 --inner_step 1"
run full_off DYB_PAR_PASSES=0 24 "$F" 1
run full_on DYB_PAR_PASSES=1 24 "$F" 1
run full_on_q8 "DYB_PAR_PASSES=1 GPU_MAX_HW_QUEUES=8" 24 "$F" 1
run full_off_q8 "DYB_PAR_PASSES=0 GPU_MAX_HW_QUEUES=8" 24 "$F" 1
run s1_q4 X=1 60 "" 1
run s1_q8 GPU_MAX_HW_QUEUES=8 60 "" 1
run s32_q4 X=1 10 "" 32
run s32_q8 GPU_MAX_HW_QUEUES=8 10 "" 32
