#!/bin/bash
# r05 s24: GPU_MAX_HW_QUEUES 4 (runtime default) vs 8 across the configurations
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s24; O=gpurun_out/s24; export TMPDIR=/tmp
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
for rep in a b; do for q in 4 8; do
run s1_q$q$rep GPU_MAX_HW_QUEUES=$q 60 "" 1
run so_q$q$rep GPU_MAX_HW_QUEUES=$q 10 "--second_order 1" 1
run s5_q$q$rep GPU_MAX_HW_QUEUES=$q 20 "" 5
run dyn1_q$q$rep GPU_MAX_HW_QUEUES=$q 16 "$F --cos_sim_threshold 9.724e-05" 1
done; done
run full32_q4 GPU_MAX_HW_QUEUES=4 8 "$F --seqs_full 1" 32
run full32_q8 GPU_MAX_HW_QUEUES=8 8 "$F --seqs_full 1" 32
run b8_q4 GPU_MAX_HW_QUEUES=4 12 "--batch 8" 1
run b8_q8 GPU_MAX_HW_QUEUES=8 12 "--batch 8" 1
