#!/bin/bash
# Round 4, GPU session 11: do lockstep groups on different HIP streams share the chip?  (s5: two groups of S/2 ran no faster than one group of
# S/2 alone.)  Kernel trace of 2 groups x 2 sequences by hardware queue, with the runtime's default number of hardware queues and with 8.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
tr() {  # tag env args
  (cd /tmp && env $2 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$1 -o trace -- python $R/bench.py $3 $Q) > $O/trace_$1.log 2>&1
  t=$(find $O/tr_$1 -name "*kernel_trace.csv" | head -1)
  echo "== $1"; [ -n "$t" ] && python tools/queue_overlap.py $t | tee $O/queues_$1.txt; tail -1 $O/trace_$1.log | cut -c1-150
  rm -rf $O/tr_$1
}
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), "groups", d["config"].get("lockstep_groups"), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
tr s4g2_q4 "" "--seqs 4 --groups 2 --steps 10 --warmup 3"
tr s4g2_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 4 --groups 2 --steps 10 --warmup 3"
one s4g1 "" "--seqs 4 --steps 20 --warmup 5"
one s4g2_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 4 --groups 2 --steps 20 --warmup 5"
one s4g4_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 4 --groups 4 --steps 20 --warmup 5"
one s4g4_q16 "GPU_MAX_HW_QUEUES=16" "--seqs 4 --groups 4 --steps 20 --warmup 5"
one s2g2_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 2 --groups 2 --steps 30 --warmup 6"
one s8g2_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 8 --groups 2 --steps 16 --warmup 4"
one s32g2_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 32 --groups 2 --steps 12 --warmup 3"
one s32g1_q8 "GPU_MAX_HW_QUEUES=8" "--seqs 32 --steps 12 --warmup 3"
