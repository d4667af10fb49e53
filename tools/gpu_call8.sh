#!/bin/bash
# GPU box: the default bench line under a hard timeout, then build() / smoke()
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 420 python bench.py ) > gpurun_out/bench8.json 2> gpurun_out/bench8.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench8.json; tail -5 gpurun_out/bench8.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
