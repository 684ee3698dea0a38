#!/bin/bash
# r05 s25: parallel passes with 8 hardware queues: two streams (history, exemplar) vs three (+ teacher forward)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s25; O=gpurun_out/s25; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
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
for rep in a b; do for p in 0 1 2; do
run full_p$p$rep DYB_PAR_PASSES=$p 24 "$F"
done; done
for p in 1 2; do run dyn_p$p DYB_PAR_PASSES=$p 16 "$F --cos_sim_threshold 9.724e-05"; done
DYB_PAR_PASSES=2 timeout 600 python -m pytest tests/test_replica_full_gpu.py -q -x -k "parallel" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
