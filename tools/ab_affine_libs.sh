#!/bin/bash
# GPU box: cfg 2 flow time with each library variant (tools/build_variants.sh), same box, same process layout
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="gpurun_variants/lib_$v.so"; fi
  echo "== $v"; BGK_LIB=$lib python tools/ab_affine.py 2>&1 | grep "variant 2" | tail -1
done
