#!/bin/bash
# Round 4, GPU session 13 (no kernel-source change since the closing session): autograd nodes without zero-filled gradients of unused outputs
# (second-order frame rate, parity), and fresh traces of the two launch-bound configurations - one sequence first order, one sequence second order.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 2), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one so "" "--second_order 1 --seqs 1 --steps 12 --warmup 3"
one so_b "" "--second_order 1 --seqs 1 --steps 12 --warmup 3"
one so_full "" "--second_order 1 --full_losses 1 --inner_step 1 --seqs 1 --steps 6 --warmup 2"
timeout 900 python -m pytest tests/test_adaptation_gpu.py -q -m gpu -k "fused_level or second_order or first_frame or autograd" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
tr() {  # tag args
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$1 -o trace -- python $R/bench.py $2 $Q) > $O/trace_$1.log 2>&1
  t=$(find $O/tr_$1 -name "*kernel_trace.csv" | head -1); f=$(find $O/tr_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1"; [ -n "$f" ] && cp $f $O/kernel_stats_$1.csv
  [ -n "$t" ] && python tools/queue_overlap.py $t | tee $O/queues_$1.txt
  [ -n "$t" ] && [ "$1" = "fo_S1" ] && python tools/frame_timeline.py $t $O/frame_timeline_$1.txt > /dev/null && head -4 $O/frame_timeline_$1.txt
  rm -rf $O/tr_$1
}
tr fo_S1 "--seqs 1 --steps 40 --warmup 8"
tr so_S1 "--second_order 1 --seqs 1 --steps 10 --warmup 2"
