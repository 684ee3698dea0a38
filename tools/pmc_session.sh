mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 40 --warmup 8 --no_cpu_baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; cut -c1-200 gpurun_out/bench.json; python -c "import json; d=json.load(open('gpurun_out/bench.json')); print(json.dumps(d.get('roofline'))[:900])"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --overlap 0 --no_cpu_baseline --no_roofline) > gpurun_out/pmc_$C.log 2>&1
  echo "pmc $C rc=$?"
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C gpurun_out/pmc_$C.json > gpurun_out/pmc_$C.txt 2>&1
  rm -rf gpurun_out/pmc_$C
  head -14 gpurun_out/pmc_$C.txt
done
