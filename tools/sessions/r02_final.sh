#!/bin/bash
# Round 2 closing measurement at HEAD: full parity suite, smoke, default bench (32 sequences per GPU; sweep, sub-records, CPU
# baseline, per-shape conv table), kernel-trace stats at S = 32, the two PMC traffic passes (S = 32).  All into gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
COMMIT=${1:-unknown}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --conv_table gpurun_out/conv_table_default.csv > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/bench_default.json'))
    print({k: d[k] for k in ('value', 'ms_per_step', 'host_issue_ms_per_step', 'n_gpus')})
    for k in ('roofline', 'frame_time_ms', 'sequences_per_gpu_sweep', 'second_order', 'batch8_exemplars', 'batch16_fp32_vs_bf16', 'full_default_losses', 'cpu_baseline'):
        print(k, json.dumps(d.get(k))[:700])
except Exception as e:
    print('bench json unreadable', e)
PY
tail -3 gpurun_out/bench_default.err
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_S32 -o trace -- python $R/bench.py --seqs 32 --steps 6 --warmup 2 $Q) > gpurun_out/prof_S32.log 2>&1
cp $(find gpurun_out/prof_S32 -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_S32.csv 2>/dev/null; rm -rf gpurun_out/prof_S32
head -8 gpurun_out/kernel_stats_S32.csv | cut -c1-160
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > gpurun_out/pmc_$C.log 2>&1
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C gpurun_out/pmc_$C.json > gpurun_out/pmc_$C.txt 2>&1
  rm -rf gpurun_out/pmc_$C
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json $COMMIT gpurun_out/pmc_igemm_traffic.json 32 > gpurun_out/pmc_traffic.txt 2>&1
cat gpurun_out/pmc_traffic.txt
