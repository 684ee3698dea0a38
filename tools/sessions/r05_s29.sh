#!/bin/bash
# r05 s29: parallel passes for launches that cover a few replicas (small groups, the tail of the dynamic loop): parity, sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s29; O=gpurun_out/s29; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_replica_full_gpu.py -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
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
for S in 2 3 4; do
  run s${S}_m1 DYB_PAR_MAX_REPLICAS=1 $S 12
  run s${S}_m4 DYB_PAR_MAX_REPLICAS=4 $S 12
done
run s5_m4 DYB_PAR_MAX_REPLICAS=4 5 12
run s5_m8 DYB_PAR_MAX_REPLICAS=8 5 12
run s8_m8 DYB_PAR_MAX_REPLICAS=8 8 12
run s8_m4 DYB_PAR_MAX_REPLICAS=4 8 12
run dyn32_m1 DYB_PAR_MAX_REPLICAS=1 32 6 "--cos_sim_threshold 9.724e-05"
run dyn32_m4 DYB_PAR_MAX_REPLICAS=4 32 6 "--cos_sim_threshold 9.724e-05"
