#!/bin/bash
# GPU box, round 4 experiment 1: MFMA event-order micro-benchmark + A/B of the kernel variants in gpurun_variants/ + accuracy check of each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_exp1
mkdir -p $OUT
timeout 120 tools/ubench/mfma_order > $OUT/mfma_order.txt 2>&1
cat $OUT/mfma_order.txt
for f in gpurun_variants/lib_*.so; do
  n=$(basename $f .so)
  BGK_LIB=$PWD/$f timeout 300 python tools/dev_v2.py --check > $OUT/check_$n.txt 2>&1
  echo "== $n check"; tail -18 $OUT/check_$n.txt | cut -c1-230 | tail -6
done
timeout 900 bash tools/ab_variants.sh r04_exp1/ab
