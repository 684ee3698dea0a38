#!/bin/bash
# r06 s12: fuse_fast on the latency schedule (one sequence) + fold launches: parity, then A/B at 1 / 2 / 4 / 5 / 32 sequences
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s12; O=gpurun_out/s12; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s -k "writes_fast_weights or fast_weights_from or native_stepper_is_bit or stream_matches or first_frame" 2>&1 | grep -v "^$" > $O/pytest_sel.log
grep -E "FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_sel.log | cut -c1-400 | tail -20
for rep in 1 2; do
for s in 1 2 4 5 32; do for f in 0 1; do
steps=20; [ $s = 1 ] && steps=100
DYB_FUSE_FAST=$f timeout 300 python bench.py --steps $steps --warmup 5 --seqs $s --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('S=$s fuse_fast=$f', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
