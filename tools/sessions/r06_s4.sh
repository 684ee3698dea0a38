#!/bin/bash
# r06 s4: the whole parity suite (three fp32 draws in the noise files, refined gate assertions)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s4; O=gpurun_out/s4; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s 2>&1 | grep -v "^$" > $O/pytest_gpu_full.log
grep -E "end-of-stream|^gate |train-mode|first-frame|FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_gpu_full.log | cut -c1-520 | tail -70
