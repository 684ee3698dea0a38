#!/bin/bash
# r06 s7: Adam from the outer level's weight-gradient epilogue ("fuse_adam"): parity (kernel cases vs torch.optim.Adam, stepper bit-identity
# of the four fuse_fast x fuse_adam combinations, headline goldens), then A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s7; O=gpurun_out/s7; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s -k "writes_fast_weights or fast_weights_from or headline or ranged or replica" 2>&1 | grep -v "^$" > $O/pytest_sel.log
grep -E "FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_sel.log | cut -c1-400 | tail -30
for rep in 1 2 3; do
for f in 0 1; do
DYB_FUSE_ADAM=$f timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records > $O/bench_adam$f.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_adam$f.json')); print('fuse_adam=$f', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4), 'union', round(d['roofline']['achieved_while_convs_run'],1))"
done; done
for s in 5 8 16 64; do for f in 0 1; do
DYB_FUSE_ADAM=$f timeout 300 python bench.py --steps 20 --warmup 5 --seqs $s --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('S=$s fuse_adam=$f', round(d['value'],1))"
done; done
