#!/bin/bash
# Round-end measurement session on the GPU box: default bench (with CPU baseline), variants, kernel-trace stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --steps 40 --warmup 8 --full_losses 1 --inner_step 1 --no_cpu_baseline --no_roofline > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
timeout 300 python bench.py --steps 40 --warmup 8 --schedule minimal --no_cpu_baseline --no_roofline > gpurun_out/bench_min.json 2> gpurun_out/bench_min.err
timeout 300 python bench.py --steps 30 --warmup 6 --batch 8 --no_cpu_baseline --no_roofline > gpurun_out/bench_b8.json 2> gpurun_out/bench_b8.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline) > gpurun_out/prof.log 2>&1
python tools/trace_analyze.py $(find gpurun_out/prof -name "*kernel_trace.csv" | head -1) gpurun_out/trace_summary.json > gpurun_out/trace_summary.txt 2>&1
rm -f $(find gpurun_out/prof -name "*kernel_trace.csv")
timeout 600 python -m pytest tests/test_adaptation_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k second_order -s 2>&1 | grep -E "second-order grad|passed|failed" > gpurun_out/pytest_so.log
for f in bench_default bench_full bench_min bench_b8; do echo "== $f"; cut -c1-220 gpurun_out/$f.json; done
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(json.dumps(d['cpu_baseline'])); print(json.dumps(d['roofline'])[:400])"
cat gpurun_out/pytest_so.log
