#!/bin/bash
# GPU box, round 5 call 3: failing tests with tracebacks, stage-wise gradient attribution, phase stamps of bgk_dense_backward_dx
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -s -k "narrow or any_bin or aten or chain" 2>&1 | grep -vE "^\s*$" | grep -E "passed|failed|FAILED|Error|assert|device kernels|^E " | cut -c1-400 | tail -40 | tee $O/tests.txt
timeout 600 python tools/r05_grad_stage_diag.py 2>&1 | grep -v Warning | tail -60 | tee $O/stage_diag.txt
BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -20 | tee $O/dx_ts.txt
# the spline VJP with the knots on the hardware forms (-DBGK_VJP_FAST=1): its tests, then the KL step base / fast / base / fast
BGK_LIB=$PWD/gpurun_variants/lib_vjpfast.so timeout 900 python -m pytest tests -m gpu -q -k "rqs or spline or grad or train or kl or chain" --deselect tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch 2>&1 | grep -E "passed|failed|^FAILED" | tail -12 | tee $O/vjpfast_tests.txt
for v in base vjpfast base vjpfast; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  BGK_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   '$v' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/vjpfast.txt
done
BGK_LIB=$PWD/gpurun_variants/lib_vjpfast.so bash tools/prof_kl.sh 2>&1 | grep -E "rqs_bwd|dense_bwd|wgrad_kernel|train_kernel" | tee $O/vjpfast_kernels.txt
