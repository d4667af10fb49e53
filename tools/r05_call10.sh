#!/bin/bash
# GPU box, round 5 call 10: IC backward (lean sweeps + fix-up launch) tests + kernel time, the KL-gradient test, KL step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -s -k "ic_backward or ic2xyz or global_ic or tail_as_one or generation_tail or kl_gradient or training" 2>&1 | grep -E "passed|failed|^FAILED|IC backward|KL gradient|^E " | cut -c1-330 | tee $O/ic_tests.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
for rep in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3))' | tee -a $O/kl.txt; done
