#!/bin/bash
# GPU box, round 5 call 14: IC backward after the table change (tests, stamps, kernel times); the all-samples f64 form of the KL-gradient test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c14; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q -k "ic_backward or ic2xyz or large_molecules" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -6 | tee $O/tests.txt
BGK_LIB=$PWD/gpurun_variants/lib_icbts.so timeout 300 python tools/r05_icb_ts.py 2>&1 | tail -9 | tee $O/icb_ts.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_slow.py -m gpu_slow -q -s 2>&1 | grep -E "passed|failed|flat KL gradient|^E " | cut -c1-300 | tee $O/kl_full.txt
