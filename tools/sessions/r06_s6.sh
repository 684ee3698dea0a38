#!/bin/bash
# r06 s6: the auxiliary (weight-gradient) stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask): headline A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s6; O=gpurun_out/s6; export TMPDIR=/tmp
for m in "" alt:256 half:256 q3:256 q1:256 "" hi:256; do
DYB_AUX_CU_MASK=$m timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records 2>$O/err_$m.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mask=[$m]', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(d['roofline']['achieved'],1), 'union', round(d['roofline']['achieved_while_convs_run'],1))" || tail -3 $O/err_$m.txt
done
