#!/bin/bash
# r05 s20: one sequence - the split-K cost model's constants now that folds happen in-kernel (env-read once per process)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s20; O=gpurun_out/s20; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
run() { env $2 timeout 200 python bench.py --seqs 1 --steps 60 --warmup 10 $Q > $O/b_$1.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1]); print("$1 [$2]", round(d["value"], 1), round(d["ms_per_step"], 3), flush=True)
PY
}
run base X=1
run fold2 DYB_BWD_FOLD_US=2.0
run fold1 DYB_BWD_FOLD_US=0.8
run cap2k DYB_GRID_CAP=2048
run cap512 DYB_GRID_CAP=512
run kstep7 DYB_KSTEP_US=0.7
run kstep3 DYB_KSTEP_US=0.35
run fwdslab DYB_FWD_SLAB_US=0.05
run fwdslab4 DYB_FWD_SLAB_US=0.4
run min1 DYB_MIN_STEPS=1
run min4 DYB_MIN_STEPS=4
run base2 X=1
