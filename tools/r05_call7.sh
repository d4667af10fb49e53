#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c7; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q -s -k "ic_backward or ic2xyz or global_ic or tail_as_one" 2>&1 | grep -E "passed|failed|^FAILED|IC backward|^E " | cut -c1-250 | tee $O/ic_tests.txt
timeout 600 python tools/r05_grad_persample_diag.py 8192 2>&1 | grep -v Warning | grep -E "forward|after block (20|19|15| 0)|prior z|g_z" | cut -c1-420 > $O/persample.txt; head -8 $O/persample.txt
timeout 900 python tools/r05_klgrad_diag.py 8192 2>&1 | grep -v Warning | head -4 | tee $O/klgrad_diag.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-200 | tee $O/kl_ic_bwd.txt
