#!/bin/bash
# GPU box, round 5 call 26: at most 128 / 192 split-K slabs for the one-block weight-gradient GEMMs (layers 1 and 0: 256 shipped): KL step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c26; mkdir -p $O
kl() { BGK_LIB=$2 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   '"$1"' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/kl_ab.txt; }
for rep in 1 2; do
  kl slabs256 ""
  kl cap128 $PWD/gpurun_variants/lib_cap128.so
  kl cap192 $PWD/gpurun_variants/lib_cap192.so
done
