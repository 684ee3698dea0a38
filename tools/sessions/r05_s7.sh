#!/bin/bash
# r05 s7: one sequence - where does the chain wait for the auxiliary queue (window around the largest main-queue gap), side-stream variants;
# the whole GPU suite on the current tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s7; O=gpurun_out/s7; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
run() { # tag env seqs steps warm extra
  env $2 timeout 300 python bench.py --seqs $3 --steps $4 --warmup $5 $Q $6 > $O/b_$1.json 2> $O/b_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1])
    print("$1 [$2] S=$3 $6:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), flush=True)
except Exception as e:
    print("$1 failed", e, open("$O/b_$1.err").read()[-1500:])
PY
}
run s1_ov2 X=1 1 60 10 "--overlap 2"
run s1_ov1 X=1 1 60 10 "--overlap 1"
run s1_ov0 X=1 1 60 10 "--overlap 0"
run s1_noaux DYB_NO_AUX=1 1 60 10 "--overlap 2"
run s1_ov2b X=1 1 60 10 "--overlap 2"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o trace -- python $R/bench.py --seqs 1 --steps 8 --warmup 2 $Q) > $O/trace.log 2>&1
t=$(find $O/tr -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/gap_window.py $t 1600 150 0 > $O/gap_window0.txt && python tools/gap_window.py $t 600 100 1 > $O/gap_window1.txt && head -3 $O/gap_window0.txt
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_s1.txt
rm -rf $O/tr
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
