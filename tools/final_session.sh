R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline) > gpurun_out/prof.log 2>&1
python tools/trace_analyze.py $(find gpurun_out/prof -name "*kernel_trace.csv" | head -1) gpurun_out/trace_summary.json > gpurun_out/trace_summary.txt 2>&1
rm -f $(find gpurun_out/prof -name "*kernel_trace.csv")
timeout 100 python tools/enginebench.py > gpurun_out/enginebench.log 2>&1
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step']); print(json.dumps(d['cpu_baseline'])[:200]); print(json.dumps(d['roofline'])[:420])"
tail -1 gpurun_out/enginebench.log | cut -c1-240
