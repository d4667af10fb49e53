#!/bin/bash
# GPU box, round 5 call 23: split-K slabs of the weight-gradient kernel (workgroups per GEMM: 256 / 384 / 512 = shipped / 768): KL step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c23; mkdir -p $O
kl() { BGK_LIB=$2 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   '"$1"' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/kl_ab.txt; }
for rep in 1 2; do
  kl wg512 ""
  kl wg256 $PWD/gpurun_variants/lib_wg256.so
  kl wg384 $PWD/gpurun_variants/lib_wg384.so
  kl wg768 $PWD/gpurun_variants/lib_wg768.so
done
