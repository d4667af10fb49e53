#!/bin/bash
# GPU box, round 5 call 16: bgk_dense_backward_dx with the first GEMM's operands shared through LDS: tests, KL step A/B, phase stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c16; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "dense_backward or backward_dx or dx or train or kl_gradient or chain or narrow or weight_grad" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -8 | tee $O/tests.txt
kl() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   '"$1"' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/kl_ab.txt; }
for rep in 1 2 3; do
  kl shared
  BGK_LIB=$PWD/gpurun_variants/lib_dxstream.so kl stream
done
for n_in in 17 9; do BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py $n_in 2>&1 | tail -15 | tee -a $O/dx_ts.txt; done
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|dense_bwd|total GPU|train_kernel|wgrad_kernel|rqs_bwd" | cut -c1-160 | tee $O/kl_kernels.txt
