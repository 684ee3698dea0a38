#!/bin/bash
# r06 s16: split-K targets per mode: fewer, longer weight-gradient workgroups (no slabs, the update in the epilogue, fewer slots taken from the chain) /
# data gradients with fewer slabs for the one-pass GroupNorm backward to fold
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
run A=0
run DYB_TP_GRID_WGRAD=256
run DYB_TP_GRID_WGRAD=128
run DYB_TP_GRID_WGRAD=32
run DYB_TP_GRID_DGRAD=256
run DYB_TP_GRID_DGRAD=128
run A=0
run DYB_TP_GRID_WGRAD=1024
run DYB_TP_GRID_DGRAD=1024
