#!/bin/bash
# r05 s10: dynamic-BOA loop with the extra steps' forward shared (the previous step's final inference): parity tests, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s10; O=gpurun_out/s10; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_replica_full_gpu.py tests/test_adaptation_gpu.py -q -x > $O/pytest_dyn.txt 2>&1; tail -3 $O/pytest_dyn.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
run() { # tag env seqs steps warm extra
  env $2 timeout 300 python bench.py --seqs $3 --steps $4 --warmup $5 $Q $6 > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1])
    print("$1 [$2] S=$3 $6:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step", d.get("dynamic_loop_extra_steps_mean"), flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-1500:])
PY
}
F="--full_losses 1 --inner_step 1 --seqs_full 1 --cos_sim_threshold 9.724e-05"
run dyn32_off DYB_SHARE_DYN_FWD=0 32 6 2 "$F"
run dyn32_on DYB_SHARE_DYN_FWD=1 32 6 2 "$F"
run dyn1_off DYB_SHARE_DYN_FWD=0 1 16 4 "--full_losses 1 --inner_step 1 --cos_sim_threshold 9.724e-05"
run dyn1_on DYB_SHARE_DYN_FWD=1 1 16 4 "--full_losses 1 --inner_step 1 --cos_sim_threshold 9.724e-05"
