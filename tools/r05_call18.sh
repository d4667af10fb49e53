#!/bin/bash
# GPU box, round 5 call 18: what the parameter write-out of the training forward costs as 64 dword stores per chunk, as 16 sixteen-byte stores
# (timing experiment, wrong layout) and not at all
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c18; mkdir -p $O
for v in base widestore nopsave; do
  lib=""; [ $v != base ] && lib=$PWD/gpurun_variants/lib_$v.so
  BGK_LIB=$lib bash tools/prof_kl.sh 2>&1 | grep -E "train_kernel" | cut -c1-140 | sed "s/^/$v  /" | tee -a $O/fwd.txt
done
