#!/bin/bash
# GPU box: accuracy check + same-box A/B of the kernel variants in gpurun_variants/ (+ the phase timestamps of lib_ts.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04_exp2}
mkdir -p $OUT
for f in gpurun_variants/lib_*.so; do
  n=$(basename $f .so)
  [ "$n" = lib_ts ] && continue
  [ "$n" = lib_base ] && continue
  BGK_LIB=$PWD/$f timeout 300 python tools/dev_v2.py --check > $OUT/check_$n.txt 2>&1
  echo "== $n check"; grep -E "inv=" $OUT/check_$n.txt | cut -c1-200
done
timeout 900 bash tools/ab_variants.sh ${1:-r04_exp2}/ab
[ -f gpurun_variants/lib_ts.so ] && BGK_LIB=$PWD/gpurun_variants/lib_ts.so timeout 200 python tools/r04_phase_ts.py > $OUT/phase_ts.txt 2>&1; cat $OUT/phase_ts.txt
