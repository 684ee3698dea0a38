#!/bin/bash
# r05 s30: parallel passes: up to how many replicas per launch?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s30; O=gpurun_out/s30; export TMPDIR=/tmp
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
run s16_m8 DYB_PAR_MAX_REPLICAS=8 16 10
run s16_m16 DYB_PAR_MAX_REPLICAS=16 16 10
run s32_m8 DYB_PAR_MAX_REPLICAS=8 32 8
run s32_m64 DYB_PAR_MAX_REPLICAS=64 32 8
run dyn32_m8 DYB_PAR_MAX_REPLICAS=8 32 6 "--cos_sim_threshold 9.724e-05"
run dyn32_m16 DYB_PAR_MAX_REPLICAS=16 32 6 "--cos_sim_threshold 9.724e-05"
run dyn32_m64 DYB_PAR_MAX_REPLICAS=64 32 6 "--cos_sim_threshold 9.724e-05"
