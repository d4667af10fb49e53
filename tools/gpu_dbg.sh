#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3dbg
timeout 600 python tools/dbg_r3.py $DBGARGS 2>&1 | grep -v Warning | tail -80 > gpurun_out/r3dbg/out.txt
cat gpurun_out/r3dbg/out.txt
