#!/bin/bash
# Timing ablations of the split-f16 fused kernel.  Build step (here, no GPU): tools/ablate_h2.sh build
# Run step (on the GPU box): tools/ablate_h2.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p gpurun_variants
  for v in 0 1 2 3 4 5 7 8 15; do
    BGK_EXTRA_FLAGS="-DBGK_ABL=$v" python -m bgflow_amd.build --force > /dev/null 2>&1
    cp bgflow_amd/libbgflow_amd.so gpurun_variants/lib_abl$v.so
  done
  python -m bgflow_amd.build --force > /dev/null 2>&1
else
  cp bgflow_amd/libbgflow_amd.so /tmp/lib_orig.so
  for f in gpurun_variants/lib_abl*.so; do
    cp $f bgflow_amd/libbgflow_amd.so
    echo -n "$(basename $f): "; BGK_GEMM=f16x2 python tools/prof_layer.py fused-BA 1048576 20 2>/dev/null | tail -1
  done
  cp /tmp/lib_orig.so bgflow_amd/libbgflow_amd.so
fi
