#!/bin/bash
# r06 s8: fuse_adam after dyb_adam_one's roundings were spelled out: bit-identity tests + the stream goldens
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s8; O=gpurun_out/s8; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s -k "writes_fast_weights or fast_weights_from or headline or ranged or replica or stream_matches or adam" 2>&1 | grep -v "^$" > $O/pytest_sel.log
grep -E "FAILED|ERROR|passed|failed|AssertionError|^E  " $O/pytest_sel.log | cut -c1-400 | tail -30
