#!/bin/bash
# GPU box, round 5 call 11: IC backward tests + kernel times; per-sample attribution of chunk 31 of the KL-gradient test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q -s -k "ic_backward or ic2xyz or global_ic or tail_as_one or generation_tail" 2>&1 | grep -E "passed|failed|^FAILED|IC backward|^E " | cut -c1-330 | tee $O/ic_tests.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
timeout 600 python tools/r05_grad_persample_diag.py 8192 31 2>&1 | grep -v Warning | grep -E "forward|after block (20|19|15| 0)|prior z|g_z" | cut -c1-420 > $O/persample31.txt; head -30 $O/persample31.txt | cut -c1-330
