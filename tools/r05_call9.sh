#!/bin/bash
# GPU box, round 5 call 9: IC backward tests + kernel time; phase stamps of bgk_dense_backward_dx
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q -s -k "ic_backward or ic2xyz or global_ic or tail_as_one or generation_tail" 2>&1 | grep -E "passed|failed|^FAILED|IC backward|^E " | cut -c1-330 | tee $O/ic_tests.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
for n_in in 17 9; do BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py $n_in 2>&1 | tail -16 | tee -a $O/dx_ts.txt; done
