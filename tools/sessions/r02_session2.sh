#!/bin/bash
# Round 2 measurement session at HEAD: parity suite, smoke, default bench (16 sequences per GPU; sweep, sub-records, CPU baseline),
# kernel-trace stats at S = 16 and S = 1, the two PMC passes (S = 16), engine probe.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
COMMIT=${1:-unknown}
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for S in 16 1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_S$S -o trace -- python $R/bench.py --seqs $S --steps 8 --warmup 2 $Q) > gpurun_out/prof_S$S.log 2>&1
  cp $(find gpurun_out/prof_S$S -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_S$S.csv 2>/dev/null
  [ $S = 1 ] && python tools/frame_timeline.py $(find gpurun_out/prof_S$S -name "*kernel_trace.csv" | head -1) gpurun_out/frame_timeline_S1.txt > /dev/null 2>&1
  rm -rf gpurun_out/prof_S$S
done
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --seqs 16 --steps 2 --warmup 1 $Q) > gpurun_out/pmc_$C.log 2>&1
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C gpurun_out/pmc_$C.json > gpurun_out/pmc_$C.txt 2>&1
  rm -rf gpurun_out/pmc_$C
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json $COMMIT gpurun_out/pmc_igemm_traffic.json > gpurun_out/pmc_traffic.txt 2>&1
timeout 200 python tools/enginebench.py > gpurun_out/enginebench.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/bench_default.json'))
    print({k: d[k] for k in ('value', 'ms_per_step', 'host_issue_ms_per_step', 'n_gpus')})
    for k in ('roofline', 'frame_time_ms', 'sequences_per_gpu_sweep', 'second_order', 'batch8_exemplars', 'batch16_fp32_vs_bf16', 'full_default_losses', 'cpu_baseline'):
        print(k, json.dumps(d.get(k))[:1200])
except Exception as e:
    print('bench json unreadable', e)
PY
tail -3 gpurun_out/bench_default.err; cat gpurun_out/pmc_traffic.txt; tail -1 gpurun_out/enginebench.log | cut -c1-300
