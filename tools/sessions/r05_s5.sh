#!/bin/bash
# r05 s5: in-kernel fold in the tangent passes (second order), default term set at 32 sequences: queue timelines; 4096^3 calibration
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s5; O=gpurun_out/s5; export TMPDIR=/tmp
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
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_adaptation_gpu.py -q -x -k "hessian or second_order or tangent" > $O/pytest_so.txt 2>&1; tail -3 $O/pytest_so.txt
run so_off DYB_LAT_FOLD=0 1 12 3 "--second_order 1"
run so_on DYB_LAT_FOLD=1 1 12 3 "--second_order 1"
run so_off2 DYB_LAT_FOLD=0 1 12 3 "--second_order 1"
run so_on2 DYB_LAT_FOLD=1 1 12 3 "--second_order 1"
run full1_off DYB_LAT_FOLD=0 1 20 4 "--full_losses 1 --inner_step 1"
run full1_on DYB_LAT_FOLD=1 1 20 4 "--full_losses 1 --inner_step 1"
run full32 X=1 32 6 2 "--full_losses 1 --inner_step 1 --seqs_full 1"
trace() { # tag env seqs steps warm extra
  (cd /tmp && env $2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$1 -o trace -- python $R/bench.py --seqs $3 --steps $4 --warmup $5 $Q $6) > $O/trace_$1.log 2>&1
  f=$(find $O/tr_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$1.csv
  t=$(find $O/tr_$1 -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_$1.txt && head -5 $O/frame_timeline_$1.txt
  rm -rf $O/tr_$1
}
trace full32 X=1 32 4 1 "--full_losses 1 --inner_step 1 --seqs_full 1"
timeout 200 python tools/tp_lab.py 16 16 4096 4096 1 1 > $O/gemm4096.txt 2>&1; cat $O/gemm4096.txt
