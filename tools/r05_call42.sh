#!/bin/bash
# GPU box, round 5 call 42: phase stamps of the recompute backward kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c42; mkdir -p $O
BGK_LIB=$PWD/gpurun_variants/lib_ts.so timeout 300 python tools/r05_rc_ts.py 2>&1 | tail -10 | tee $O/rc_ts.txt
