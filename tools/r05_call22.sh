#!/bin/bash
# GPU box, round 5 call 22: timing experiment -- the gradient stream of bgk_dense_backward_dx requested as 8 rows x 128 contiguous bytes per
# instruction (same bytes, wrong placement) against one 32-byte piece of 32 rows per instruction
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c22; mkdir -p $O
for v in base lines base lines; do
  lib=""; [ $v != base ] && lib=$PWD/gpurun_variants/lib_$v.so
  BGK_LIB=$lib bash tools/prof_kl.sh 2>&1 | grep -E "dense_bwd_dx" | cut -c1-140 | sed "s/^/$v  /" | tee -a $O/dx.txt
done
