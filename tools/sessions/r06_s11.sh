#!/bin/bash
# r06 s11: GroupNorm backward beside the weight gradients, re-measured on the round-6 schedule: one-pass workgroup size, conv occupancy cap, chain priority
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s11; O=gpurun_out/s11; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
run A=0
run DYB_TP_GN_THREADS=512
run DYB_TP_GN_THREADS=256
run DYB_TP_OCC=2
run DYB_CHAIN_PRIORITY=-1
run DYB_TP_GN_ONEPASS=0
run A=0
run DYB_TP_GN_THREADS=512 DYB_CHAIN_PRIORITY=-1
run DYB_TP_GRID=768
run DYB_TP_GRID=384
