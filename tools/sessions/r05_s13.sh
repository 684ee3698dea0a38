#!/bin/bash
# r05 s13: the N > 1 control flow of bench.py on a one-GPU box (two ranks on cuda:0, gloo) after the time-slicing fix
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s13; O=gpurun_out/s13; export TMPDIR=/tmp
DYB_BENCH_SMOKE_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no_cpu_baseline > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_two_ranks_one_gpu.json").read().strip().splitlines()[-1])
    print("two ranks on one GPU (control flow only):", d["n_gpus"], round(d["value"], 1), d["scaling"], (d.get("pw3d_operating_point") or {}).get("value"))
except Exception as e:
    print("two-rank smoke failed:", e, open("$O/bench_two_ranks_one_gpu.err").read()[-1200:])
PY
