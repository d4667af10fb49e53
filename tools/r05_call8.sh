#!/bin/bash
# GPU box, round 5 call 8: IC backward (DMA staging, clamp-aware adjoint) tests + timing; KL step A/B: padding rows between the halves
# of the [2, B, 128] allocations, spline VJP with the knots on the hardware forms, gradient batches of 2 k-steps in bgk_dense_backward_dx
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q -s -k "ic_backward or ic2xyz or global_ic or tail_as_one or generation_tail" 2>&1 | grep -E "passed|failed|^FAILED|IC backward|^E " | cut -c1-600 | tee $O/ic_tests.txt
kl() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   '"$1"' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/kl_ab.txt; }
for rep in 1 2; do
  kl base
  BGK_IC_BWD_NODMA=1 kl ic_nodma
  BGK_HALF_PAD_ROWS=16 kl pad16
  BGK_HALF_PAD_ROWS=48 kl pad48
  BGK_LIB=$PWD/gpurun_variants/lib_vjpfast.so kl vjpfast
  BGK_LIB=$PWD/gpurun_variants/lib_dg2.so kl dg2
done
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU|train_kernel|dense_bwd|wgrad_kernel|rqs_bwd" | cut -c1-160 | tee $O/kl_kernels.txt
