#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel trace.  Everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [stages...]   stages: info tests smoke bench prof
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
STAGES=${@:-info tests smoke bench prof}
for s in $STAGES; do
  case $s in
    info)  (rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|Socket|Core") > $OUT/info.txt 2>&1 ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -rA 2>&1 | tail -120 > $OUT/pytest_gpu.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ;;
    bench) timeout 900 python bench.py --steps 40 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err ;;
    benchmin) timeout 600 python bench.py --steps 40 --warmup 8 --schedule minimal --no_cpu_baseline > $OUT/bench_min.json 2> $OUT/bench_min.err ;;
    prof)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline) > $OUT/prof.log 2>&1
           find $OUT/prof -name "*stats*" | head -5 >> $OUT/prof.log
           python tools/trace_analyze.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $OUT/trace_summary.json > $OUT/trace_summary.txt 2>&1
           rm -f $(find $OUT/prof -name "*kernel_trace.csv") ;;
    pmc)   for C in FETCH_SIZE WRITE_SIZE; do
             (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --overlap 0 --no_cpu_baseline --no_roofline) > $OUT/pmc_$C.log 2>&1
             f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
             python tools/pmc_summarize.py $f $C $OUT/pmc_$C.json > $OUT/pmc_$C.txt 2>&1
             rm -rf $OUT/pmc_$C
           done ;;
    micro) timeout 300 python tools/microbench.py 1 > $OUT/microbench.log 2>&1 ;;
  esac
  echo "stage $s done rc=$?" >> $OUT/stages.log
done
tail -5 $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/smoke.log 2>/dev/null | tail -3; cat $OUT/bench.json 2>/dev/null
