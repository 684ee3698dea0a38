#!/bin/bash
# r05 s22: one sequence, default term set: history / exemplar passes on the stepper's own streams (par_passes): parity, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s22; O=gpurun_out/s22; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_replica_full_gpu.py tests/test_adaptation_gpu.py -q -x -k "parallel or full or default or dynamic or forced" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
run() { env $2 timeout 300 python bench.py --seqs 1 --steps $3 --warmup 4 $Q $4 > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1]); print("$1 [$2] $4:", round(d["value"], 1), round(d["ms_per_step"], 3), flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-1200:])
PY
}
F="--full_losses 1 --inner_step 1"
for rep in a b; do
run full_off$rep DYB_PAR_PASSES=0 24 "$F"
run full_on$rep DYB_PAR_PASSES=1 24 "$F"
done
run dyn_off DYB_PAR_PASSES=0 16 "$F --cos_sim_threshold 9.724e-05"
run dyn_on DYB_PAR_PASSES=1 16 "$F --cos_sim_threshold 9.724e-05"
