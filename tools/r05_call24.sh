#!/bin/bash
# GPU box, round 5 call 24: IC backward sweep with the adjoint rows of a placement read in one round: tests, stamps, kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q -k "ic_backward or ic2xyz or global_ic or tail_as_one or generation_tail or large_molecules" 2>&1 | grep -E "passed|failed|^FAILED|^E " | cut -c1-300 | tail -5 | tee $O/tests.txt
BGK_LIB=$PWD/gpurun_variants/lib_icbts.so timeout 300 python tools/r05_icb_ts.py 2>&1 | tail -9 | tee $O/icb_ts.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
