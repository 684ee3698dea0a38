#!/bin/bash
# r06 s2: the whole parity suite without -x (every noise-derived bound at once; the reports of the stream tests), then the three
# side configurations whose exemplars come through retrieval() (pinned host cache, per-sequence upload inside the clock)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s2; O=gpurun_out/s2; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s 2>&1 | grep -v "^$" > $O/pytest_gpu_full.log
grep -E "end-of-stream|^gate |first-frame|FAILED|ERROR|passed|failed|AssertionError" $O/pytest_gpu_full.log | cut -c1-600 | tail -80
Q8="GPU_MAX_HW_QUEUES=8"
run() { env $3 timeout 300 python bench.py --sub_record "$2" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d.get('value') and round(d['value'],1), d.get('ms_per_step') and round(d['ms_per_step'],2), d.get('error'))"; }
run full_default_losses '{"name": "full_default_losses", "steps": 24, "warmup": 6, "batch": 1, "inner_step": 1, "note": "", "roofline_peak": null, "seqs": 1, "kw": {"full_losses": 1}}' $Q8
run full_default_losses_resident '{"name": "full_default_losses", "steps": 24, "warmup": 6, "batch": 1, "inner_step": 1, "note": "", "roofline_peak": null, "seqs": 1, "kw": {"full_losses": 1, "resident_exemplars": 1}}' $Q8
run batch8_exemplars '{"name": "batch8_exemplars", "steps": 10, "warmup": 3, "batch": 8, "inner_step": 3, "note": "", "roofline_peak": null, "seqs": 1, "kw": {"retrieval": 1, "lower_level_mixtrain": 1, "upper_level_mixtrain": 1, "sample_num": 8}}' A=1
run batch8_exemplars_resident '{"name": "batch8_exemplars", "steps": 10, "warmup": 3, "batch": 8, "inner_step": 3, "note": "", "roofline_peak": null, "seqs": 1, "kw": {"retrieval": 1, "lower_level_mixtrain": 1, "upper_level_mixtrain": 1, "sample_num": 8, "resident_exemplars": 1}}' A=1
run full_S32 '{"name": "full_default_losses_S32", "steps": 10, "warmup": 3, "batch": 1, "inner_step": 1, "note": "", "roofline_peak": null, "seqs": 32, "kw": {"full_losses": 1}}' $Q8
