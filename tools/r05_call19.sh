#!/bin/bash
# GPU box, round 5 call 19: element-major saved parameters: tests, KL step A/B, kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c19; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "element_major or train or kl_gradient or chain or narrow or spline_backward or rqs" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -8 | tee $O/tests.txt
kl() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   '"$1"' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/kl_ab.txt; }
for rep in 1 2 3; do
  kl element_major
  BGK_PACKED_PARAMS=0 kl reference_columns
done
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|dense_bwd|total GPU|train_kernel|wgrad_kernel|rqs_bwd" | cut -c1-160 | tee $O/kl_kernels.txt
