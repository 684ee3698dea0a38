#!/bin/bash
# r05 s35: configs[2] (batch 8 + exemplars; the exemplar pass on its own stream): 4 vs 8 hardware queues
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s35; O=gpurun_out/s35; export TMPDIR=/tmp
for q in 4 8 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --sub_record '{"name": "batch8_exemplars", "steps": 10, "warmup": 3, "batch": 8, "inner_step": 3, "note": "configs[2]", "roofline_peak": null, "seqs": 1, "kw": {"retrieval": 1, "lower_level_mixtrain": 1, "upper_level_mixtrain": 1, "sample_num": 8}}' 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch8_exemplars q=$q', round(d['value'],1), round(d['ms_per_step'],2))"
done
