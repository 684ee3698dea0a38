#!/bin/bash
# Round 2, GPU session 1: parity suite at HEAD (new: batch-8 + exemplars, second order at inner_step 3, batched single-launch
# 1x1 kernels in both dispatches), smoke, default bench (sub-records, percentiles, CPU baseline), kernel-trace stats and the two
# PMC passes at this commit.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
COMMIT=${1:-unknown}
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -x 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0) > gpurun_out/prof.log 2>&1
python tools/trace_analyze.py $(find gpurun_out/prof -name "*kernel_trace.csv" | head -1) gpurun_out/trace_summary.json > gpurun_out/trace_summary.txt 2>&1
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats.csv 2>/dev/null
rm -f $(find gpurun_out/prof -name "*kernel_trace.csv")
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --overlap 0 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0) > gpurun_out/pmc_$C.log 2>&1
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C gpurun_out/pmc_$C.json > gpurun_out/pmc_$C.txt 2>&1
  rm -rf gpurun_out/pmc_$C
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json $COMMIT gpurun_out/pmc_igemm_traffic.json > gpurun_out/pmc_traffic.txt 2>&1
timeout 200 python tools/enginebench.py > gpurun_out/enginebench.log 2>&1
tail -6 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log
cut -c1-1500 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/bench_default.json'))
    for k in ('roofline', 'frame_time_ms', 'second_order', 'batch8_exemplars', 'full_default_losses', 'cpu_baseline'):
        print(k, json.dumps(d.get(k))[:400])
except Exception as e:
    print('bench json unreadable', e)
PY
cat gpurun_out/pmc_traffic.txt; tail -1 gpurun_out/enginebench.log | cut -c1-300
