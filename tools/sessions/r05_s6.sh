#!/bin/bash
# r05 s6: default term set at 32 sequences after the host-side fixes (exemplars resident + shared, gate views cut once): rate, queue timeline,
# host profile; the same with the dynamic loop entered
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s6; O=gpurun_out/s6; export TMPDIR=/tmp
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
F="--full_losses 1 --inner_step 1 --seqs_full 1"
run full32 X=1 32 8 2 "$F"
run full32b X=1 32 8 2 "$F"
run dyn32 X=1 32 6 2 "$F --cos_sim_threshold 9.724e-05"
run full1 X=1 1 20 4 "--full_losses 1 --inner_step 1"
timeout 300 python -m cProfile -o /tmp/prof.out bench.py --seqs 32 --steps 6 --warmup 2 $Q $F > /dev/null 2> $O/cprofile.err
python - <<'PY' > gpurun_out/s6/cprofile_full32.txt 2>&1
import pstats
p = pstats.Stats("/tmp/prof.out")
p.sort_stats("cumulative").print_stats(45)
p.sort_stats("tottime").print_stats(30)
PY
head -70 $O/cprofile_full32.txt | tail -50
trace() { # tag env seqs steps warm extra
  (cd /tmp && env $2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$1 -o trace -- python $R/bench.py --seqs $3 --steps $4 --warmup $5 $Q $6) > $O/trace_$1.log 2>&1
  t=$(find $O/tr_$1 -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_$1.txt && head -12 $O/frame_timeline_$1.txt
  rm -rf $O/tr_$1
}
trace full32 X=1 32 4 1 "$F"
