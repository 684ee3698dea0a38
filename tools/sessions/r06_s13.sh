#!/bin/bash
# r06 s13: teacher EMA inside the Adam pass: parity of the default-term-set paths, A/B at 32 sequences / one sequence
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s13; O=gpurun_out/s13; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_replica_full_gpu.py tests/test_adaptation_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -s -k "ema or full or gate or gated or parallel or forward_shared or teacher" 2>&1 | grep -v "^$" > $O/pytest_sel.log
grep -E "FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_sel.log | cut -c1-400 | tail -20
Q8="GPU_MAX_HW_QUEUES=8"
run() { env $3 $4 timeout 300 python bench.py --sub_record "$2" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $4', d.get('value') and round(d['value'],1), d.get('ms_per_step') and round(d['ms_per_step'],2), d.get('error'))"; }
for rep in 1 2; do for fe in 0 1; do
run full_S32 '{"name": "full_default_losses_S32", "steps": 10, "warmup": 3, "batch": 1, "inner_step": 1, "note": "", "roofline_peak": null, "seqs": 32, "kw": {"full_losses": 1}}' $Q8 DYB_FUSE_EMA=$fe
run full_S1 '{"name": "full_default_losses", "steps": 24, "warmup": 6, "batch": 1, "inner_step": 1, "note": "", "roofline_peak": null, "seqs": 1, "kw": {"full_losses": 1}}' $Q8 DYB_FUSE_EMA=$fe
done; done
