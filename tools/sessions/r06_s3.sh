#!/bin/bash
# r06 s3: gate tests after the cosine kernel went to double accumulation (+ fp64 reference cosines in the goldens), train-mode teacher on
# the stepper, the stem's forward on the throughput kernel (C4 form): parity on the GPU, then A/B of the headline with tp_stem 0 / 1
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s3; O=gpurun_out/s3; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s -k "gate or gated or train_mode_teacher or throughput_kernel or forward_shared or headline or hmr_engine" 2>&1 | grep -v "^$" > $O/pytest_sel.log
grep -E "end-of-stream|^gate |train-mode|FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_sel.log | cut -c1-500 | tail -60
for rep in 1 2; do
for stem in 0 1; do
DYB_TP_STEM=$stem timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records --conv_table $O/conv_table_stem$stem.txt > $O/bench_stem$stem.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_stem$stem.json')); print('tp_stem=$stem', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4))"
done; done
grep "^f\|^kind" $O/conv_table_stem0.txt | head -4; grep "^f\|^kind\|224" $O/conv_table_stem1.txt | head -8
