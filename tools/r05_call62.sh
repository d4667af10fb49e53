#!/bin/bash
# GPU box, round 5 call 62: A-ring depth of the one-wave-per-SIMD kernels (bgk_dense_layer S = 12 / 16: 3 vs 6; width-256 coupling kernel: 3 vs 4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c62; mkdir -p $O
for v in base ring base ring; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  echo "== $v"; BGK_LIB=$lib python tools/r05_w256.py 1048576 10 2>&1 | grep "H=256\|H=192"
done | tee $O/ab.txt
