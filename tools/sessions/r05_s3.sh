#!/bin/bash
# r05 s3: in-kernel split-K fold (+ statistics with the tiles): kernel tests, A/B at 32 / 1 / 5 sequences, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s3; O=gpurun_out/s3; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "inkernel_fold or gnstats or hmr_engine" > $O/pytest_fold.txt 2>&1; tail -3 $O/pytest_fold.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
run() { # tag env seqs steps warm
  env $2 timeout 300 python bench.py --seqs $3 --steps $4 --warmup $5 $Q > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1])
    print("$1 [$2] S=$3:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-1500:])
PY
}
run s32_off DYB_TP_FOLD=0 32 10 3
run s32_on DYB_TP_FOLD=1 32 10 3
run s32_off2 DYB_TP_FOLD=0 32 10 3
run s32_on2 DYB_TP_FOLD=1 32 10 3
run s1_off DYB_LAT_FOLD=0 1 60 10
run s1_on DYB_LAT_FOLD=1 1 60 10
run s1_off2 DYB_LAT_FOLD=0 1 60 10
run s1_on2 DYB_LAT_FOLD=1 1 60 10
run s5_off DYB_LAT_FOLD=0 5 30 6
run s5_on DYB_LAT_FOLD=1 5 30 6
run s16_off DYB_TP_FOLD=0 16 16 4
run s16_on DYB_TP_FOLD=1 16 16 4
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
