#!/bin/bash
# GPU box: per-phase wave cycles (s_memtime) of every gpurun_variants/lib_ts*.so -- cycle-based ablations (wall-time ablations are confounded by DVFS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04_ts}
mkdir -p $OUT
for f in gpurun_variants/lib_ts*.so; do
  n=$(basename $f .so)
  BGK_LIB=$PWD/$f timeout 200 python tools/r04_phase_ts.py > $OUT/$n.txt 2>&1
  echo "== $n"; grep -A12 "B|A inv=0" $OUT/$n.txt | grep -v "^   \[" ; grep "T|F inv=0\|F|T inv=0" $OUT/$n.txt
done
