#!/bin/bash
# Round 4, GPU session 2: the one-pass GroupNorm backward without cache-wide fences (device-scope stores / loads for the exchanged
# words), first-frame outer-gradient gates, A/B of the three GroupNorm-backward modes, kernel durations by grid, a two-rank smoke of
# bench.py's N > 1 control flow on this one GPU.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_adaptation_gpu.py tests/test_headline_gpu.py -q -m gpu -s -k "onepass or one_image_throughput or first_frame_first_order or headline_32" > $O/pytest.txt 2>&1; grep -E "passed|failed|first-frame outer|worst_rel" $O/pytest.txt | tail -12
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one onepass2 DYB_TP_GN_ONEPASS=2 "--seqs 32 --steps 12 --warmup 3"
one onepass1 DYB_TP_GN_ONEPASS=1 "--seqs 32 --steps 12 --warmup 3"
one onepass0 DYB_TP_GN_ONEPASS=0 "--seqs 32 --steps 12 --warmup 3"
one onepass2b DYB_TP_GN_ONEPASS=2 "--seqs 32 --steps 12 --warmup 3"
echo "--- kernel trace, one-pass (mode 2)"
(cd /tmp && DYB_TP_GN_ONEPASS=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 6 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32_onepass.csv && python tools/step_breakdown.py $O/kernel_stats_S32_onepass.csv 8 | tee $O/step_breakdown_onepass.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_by_grid.py $t gn_bwd | tee $O/gn_bwd_by_grid.txt | head -30
[ -n "$t" ] && python tools/trace_by_grid.py $t "gn_" > $O/gn_all_by_grid.txt
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt
rm -rf $O/trf
echo "--- two ranks on one GPU (control flow of --gpus 2 only)"
DYB_BENCH_SMOKE_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --seqs 8 --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --percentile_frames 0 > $O/bench_2rank.json 2> $O/bench_2rank.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_2rank.json").read().strip().splitlines()[-1])
    print("2 ranks:", round(d["value"], 1), "frames/s; pw3d point:", d.get("pw3d_operating_point"), "workload:", d["config"]["workload"])
except Exception as e:
    print("2-rank smoke failed:", e, open("$O/bench_2rank.err").read()[-800:])
PY
