#!/bin/bash
# r05 s32: default term set with the level's passes side by side: 4 vs 8 hardware queues for replica groups
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s32; O=gpurun_out/s32; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
F="--full_losses 1 --inner_step 1 --seqs_full 1"
run() { env $2 timeout 300 python bench.py --seqs $3 --steps $4 --warmup 3 $Q $F $5 > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1]); print("$1 [$2] S=$3 $5:", round(d["value"], 1), round(d["ms_per_step"], 2), flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-800:])
PY
}
run s32_q4 GPU_MAX_HW_QUEUES=4 32 8
run s32_q8 GPU_MAX_HW_QUEUES=8 32 8
run s5_q4 GPU_MAX_HW_QUEUES=4 5 12
run s5_q8 GPU_MAX_HW_QUEUES=8 5 12
run dyn32_q4 GPU_MAX_HW_QUEUES=4 32 6 "--cos_sim_threshold 9.724e-05"
run dyn32_q8 GPU_MAX_HW_QUEUES=8 32 6 "--cos_sim_threshold 9.724e-05"
