#!/bin/bash
# r05 s1: cache policy of the throughput kernel's result stores (tp_wt): lab per shape, then the headline loop A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s1; O=gpurun_out/s1; export TMPDIR=/tmp
DYB_TP_WT=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "throughput_kernel" > $O/pytest_wt1.txt 2>&1; tail -2 $O/pytest_wt1.txt
SH="14,256,256,3,1;14,256,1024,1,1;14,1024,256,1,1;28,128,128,3,1;28,128,512,1,1;56,64,64,3,1;56,64,256,1,1;7,512,2048,1,1;7,2048,512,1,1;7,512,512,3,1;28,256,256,3,2"
timeout 300 python tools/tp_lab_multi.py 32 "$SH" "tp_wt=0" "tp_wt=1" "tp_wt=2" "tp_wt=3" > $O/lab_wt.txt 2> $O/lab_wt.err
tail -5 $O/lab_wt.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for wt in 0 1 2 0 1; do
  DYB_TP_WT=$wt timeout 200 python bench.py --seqs 32 --steps 10 --warmup 3 $Q > $O/bench_wt$wt.json 2> $O/bench_wt$wt.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_wt$wt.json").read().strip().splitlines()[-1])
    print("tp_wt=$wt headline", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step", flush=True)
except Exception as e:
    print("bench wt=$wt failed", e, open("$O/bench_wt$wt.err").read()[-800:])
PY
done
