#!/bin/bash
# GPU box, round 5 call 12: the whole -m gpu suite, IC backward kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c12; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -20 | tee $O/suite.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
