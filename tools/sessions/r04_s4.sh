#!/bin/bash
# Round 4, GPU session 4 (short): lab of the one-pass GroupNorm backward (poll spacing, workgroup size, streaming part alone) on five
# layer shapes at 32 replicas; A/B of the forward statistics leaving with the conv tiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
V="tp_gn_threads=256,tp_gn_poll=1 tp_gn_threads=256,tp_gn_poll=8 tp_gn_threads=256,tp_gn_poll=32 tp_gn_threads=256,tp_gn_poll=-1 tp_gn_threads=1024,tp_gn_poll=1 tp_gn_threads=1024,tp_gn_poll=8 tp_gn_threads=1024,tp_gn_poll=-1 tp_gn_threads=512,tp_gn_poll=8"
for SH in "32 12544 64 1 1" "32 3136 256 1 3" "32 3136 64 1 0" "32 784 512 1 3" "32 196 256 1 0" "32 196 1024 1 3" "32 49 2048 1 3"; do
  timeout 120 python tools/gn_lab.py $SH $V 2>&1 | grep cfg | tee -a $O/gn_lab.txt
done
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "one_image_throughput or layer_gnstats" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step", flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one fuse1 "DYB_TP_GN_FUSE_STATS=1" "--seqs 32 --steps 12 --warmup 3"
one fuse0 "DYB_TP_GN_FUSE_STATS=0" "--seqs 32 --steps 12 --warmup 3"
one fuse1_poll8 "DYB_TP_GN_FUSE_STATS=1 DYB_TP_GN_POLL=8" "--seqs 32 --steps 12 --warmup 3"
