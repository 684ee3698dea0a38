#!/bin/bash
# r05 s4: latency-form in-kernel fold with raw slabs kept where the consumer folds them; launch census at one sequence; is the default
# term set at 32 sequences host-bound or device-bound (queue timelines)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s4; O=gpurun_out/s4; export TMPDIR=/tmp
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
run s1_off DYB_LAT_FOLD=0 1 60 10
run s1_on DYB_LAT_FOLD=1 1 60 10
run s1_off2 DYB_LAT_FOLD=0 1 60 10
run s1_on2 DYB_LAT_FOLD=1 1 60 10
run s5_off DYB_LAT_FOLD=0 5 30 6
run s5_on DYB_LAT_FOLD=1 5 30 6
trace() { # tag env seqs steps warm extra
  (cd /tmp && env $2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$1 -o trace -- python $R/bench.py --seqs $3 --steps $4 --warmup $5 $Q $6) > $O/trace_$1.log 2>&1
  f=$(find $O/tr_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$1.csv
  t=$(find $O/tr_$1 -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_$1.txt && head -5 $O/frame_timeline_$1.txt
  rm -rf $O/tr_$1
}
trace s1_on DYB_LAT_FOLD=1 1 8 2
trace s1_off DYB_LAT_FOLD=0 1 8 2
trace full32 X=1 32 3 1 "--full_losses 1"
