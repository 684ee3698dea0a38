#!/bin/bash
# Round 4, GPU session 1: the one-pass GroupNorm backward (norm_pool.hip) on hardware - parity, A/B against the two-launch form in the
# headline configuration, kernel stats + FETCH/WRITE counters of the GroupNorm family - plus what round 3 left unmeasured (sequences
# per GPU beyond 32, two free-running groups of 32) and the 4096^3 fp32 GEMM calibration of igemm_tp_kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "onepass or one_image_throughput or test_groupnorm" > $O/pytest_gn.txt 2>&1; tail -3 $O/pytest_gn.txt
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
one onepass0 DYB_TP_GN_ONEPASS=0 "--seqs 32 --steps 12 --warmup 3"
one onepass1 DYB_TP_GN_ONEPASS=1 "--seqs 32 --steps 12 --warmup 3"
one onepass2b DYB_TP_GN_ONEPASS=2 "--seqs 32 --steps 12 --warmup 3"
one s48 DYB_TP_GN_ONEPASS=2 "--seqs 48 --steps 10 --warmup 3"
one s64 DYB_TP_GN_ONEPASS=2 "--seqs 64 --steps 8 --warmup 2"
one g2x32 DYB_TP_GN_ONEPASS=2 "--seqs 64 --groups 2 --steps 8 --warmup 2"
echo "--- 4096^3 fp32 GEMM on igemm_tp_kernel (16 x 16x16 pixels, 1x1, 4096 -> 4096)"
timeout 200 python tools/tp_lab.py 16 16 4096 4096 1 1 "tp_grid=512" "tp_grid=1024,tp_kernel=3" 2>&1 | tail -3 | tee $O/gemm4096.txt
echo "--- kernel stats, one-pass"
(cd /tmp && DYB_TP_GN_ONEPASS=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32_onepass.csv && python tools/step_breakdown.py $O/kernel_stats_S32_onepass.csv 10 | tee $O/step_breakdown_onepass.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt
rm -rf $O/trf
grep -i "gn_" $O/kernel_stats_S32_onepass.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && DYB_TP_GN_ONEPASS=2 timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > $O/pmc_$C.log 2>&1
  f=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C $O/pmc_$C.json > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
python tools/pmc_family.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/kernel_stats_S32_onepass.csv gn_ 32 $O/pmc_gn_traffic.json 2>&1 | tail -12
