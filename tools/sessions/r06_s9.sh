#!/bin/bash
# r06 s9: the regressor's matrices updated in linear_outer_kernel's epilogue (fast weights / Adam): bit-identity, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s9; O=gpurun_out/s9; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s -k "fast_weights_from or ranged_weight or linear" 2>&1 | grep -v "^$" > $O/pytest_sel.log
grep -E "FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_sel.log | cut -c1-400 | tail -20
for rep in 1 2 3; do
for f in 0 1; do
DYB_FUSE_LINEAR=$f timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fuse_linear=$f', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
