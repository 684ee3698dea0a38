#!/bin/bash
# Round-end measurement session on the GPU box: parity tests, smoke, default bench (with CPU baseline), variants,
# kernel-trace stats, PMC traffic passes, engine probe.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --steps 40 --warmup 8 --full_losses 1 --inner_step 1 --no_cpu_baseline --no_roofline > gpurun_out/bench_full.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 4 --second_order 1 --no_cpu_baseline --no_roofline > gpurun_out/bench_so.json 2> /dev/null
timeout 300 python bench.py --steps 30 --warmup 6 --batch 8 --no_cpu_baseline --no_roofline > gpurun_out/bench_b8.json 2> /dev/null
timeout 200 python tools/enginebench.py > gpurun_out/enginebench.log 2>&1
timeout 200 python tools/levelprof.py 2 > gpurun_out/levelprof.log 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline) > gpurun_out/prof.log 2>&1
python tools/trace_analyze.py $(find gpurun_out/prof -name "*kernel_trace.csv" | head -1) gpurun_out/trace_summary.json > gpurun_out/trace_summary.txt 2>&1
rm -f $(find gpurun_out/prof -name "*kernel_trace.csv")
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 240 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --overlap 0 --no_cpu_baseline --no_roofline) > gpurun_out/pmc_$C.log 2>&1
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C gpurun_out/pmc_$C.json > gpurun_out/pmc_$C.txt 2>&1
  rm -rf gpurun_out/pmc_$C
done
cat gpurun_out/pytest_gpu.log | tail -2; tail -1 gpurun_out/smoke.log
for f in bench_default bench_full bench_so bench_b8; do echo "== $f"; cut -c1-200 gpurun_out/$f.json; done
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(json.dumps(d['cpu_baseline'])[:300]); print(json.dumps(d['roofline'])[:500])"
tail -1 gpurun_out/enginebench.log | cut -c1-300; head -3 gpurun_out/pmc_FETCH_SIZE.txt; head -3 gpurun_out/pmc_WRITE_SIZE.txt
