#!/bin/bash
# GPU box: kernel-trace timing (tools/dev_v2.py --time-v2) of every gpurun_variants/lib_*.so; usage: tools/ab_variants.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1
mkdir -p $OUT
for f in gpurun_variants/lib_*.so; do
  n=$(basename $f .so)
  BGK_LIB=$PWD/$f rocprofv3 --kernel-trace --output-format csv -d $OUT/$n -o t -- python tools/dev_v2.py --time-v2 > $OUT/$n.txt 2>&1
  echo "$n: $(python tools/trace_split.py $OUT/$n | grep h2v2 | sed 's/.*h2v2_kernel<1, //')"
done
