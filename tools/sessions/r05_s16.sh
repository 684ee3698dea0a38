#!/bin/bash
# r05 s16: throughput schedule from 5 sequences per launch (drivers' policy): parity at S = 5, sweep points
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s16; O=gpurun_out/s16; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_replica_full_gpu.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
for S in 4 5 6 7 8 32; do
  timeout 200 python bench.py --seqs $S --steps 20 --warmup 4 $Q > $O/b_S$S.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/b_S$S.json").read().strip().splitlines()[-1]); print("S=$S", round(d["value"], 1), round(d["ms_per_step"], 2), flush=True)
PY
done
