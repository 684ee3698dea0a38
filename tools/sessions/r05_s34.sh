#!/bin/bash
# r05 s34: last sanity check of the committed tree: smoke + a short slice of the GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_replica_full_gpu.py tests/test_headline_gpu.py -q -x -k "parallel or S5 or 5-" 2>&1 | tail -2
