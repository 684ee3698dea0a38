#!/bin/bash
# r06 s14: weight-gradient launches capped at 2 / 1 workgroups per CU (they run beside the chain) x one-pass GroupNorm workgroup size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s14; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
run A=0
run DYB_TP_OCC_WGRAD=2
run DYB_TP_OCC_WGRAD=2 DYB_TP_GN_THREADS=512
run DYB_TP_OCC_WGRAD=2 DYB_TP_GN_THREADS=256
run DYB_TP_OCC_WGRAD=1
run DYB_TP_OCC_WGRAD=1 DYB_TP_GN_THREADS=512
run A=0
