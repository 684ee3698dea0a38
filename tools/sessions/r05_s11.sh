#!/bin/bash
# r05 s11: write-through stores in the one-pass GroupNorm backward (tp_gn_wt)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s11; O=gpurun_out/s11; export TMPDIR=/tmp
DYB_TP_GN_WT=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "onepass or hmr_engine" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for rep in a b c; do for wt in 0 1; do
  DYB_TP_GN_WT=$wt timeout 300 python bench.py --seqs 32 --steps 12 --warmup 3 $Q > $O/b_$wt$rep.json 2> $O/b_$wt$rep.err
  python - <<PY
import json
d = json.loads(open("$O/b_$wt$rep.json").read().strip().splitlines()[-1])
print("tp_gn_wt=$wt", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step", flush=True)
PY
done; done
