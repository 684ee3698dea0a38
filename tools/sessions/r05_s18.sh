#!/bin/bash
# r05 s18: default term set at 32 sequences - does the round's last policy change (tp_batch_min 8) matter here?  same-session pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s18; O=gpurun_out/s18; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
F="--full_losses 1 --inner_step 1 --seqs_full 1"
for rep in a b; do for tb in 8 16; do
  DYB_TP_BATCH_MIN=$tb timeout 300 python bench.py --seqs 32 --steps 10 --warmup 3 $Q $F > $O/b_$tb$rep.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/b_$tb$rep.json").read().strip().splitlines()[-1]); print("full32 tp_batch_min=$tb", round(d["value"], 1), round(d["ms_per_step"], 2), flush=True)
PY
done; done
