#!/bin/bash
# r05 s28: the reference's default term set by sequences per GPU (what a sharded run of the real stream uses)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s28; O=gpurun_out/s28; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
F="--full_losses 1 --inner_step 1 --seqs_full 1"
for S in 1 2 3 4 5 8 16; do
  E="X=1"; [ $S = 1 ] && E="GPU_MAX_HW_QUEUES=8"
  env $E timeout 300 python bench.py --seqs $S --steps 12 --warmup 3 $Q $F > $O/b_S$S.json 2> $O/b_S$S.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_S$S.json").read().strip().splitlines()[-1]); print("default term set S=$S", round(d["value"], 1), round(d["ms_per_step"], 2), flush=True)
except Exception as e:
    print("S=$S failed", e, open("$O/b_S$S.err").read()[-800:])
PY
done
