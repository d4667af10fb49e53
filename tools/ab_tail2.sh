#!/bin/bash
# GPU box: time of the sampling-tail kernel (tools/dbg_tail.py --time-only) with every gpurun_variants/lib_*.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in gpurun_variants/lib_*.so; do
  echo "$(basename $f .so): $(BGK_LIB=$PWD/$f python tools/dbg_tail.py --time-only 2>/dev/null | grep 'register=True')"
done
