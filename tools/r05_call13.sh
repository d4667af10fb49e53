#!/bin/bash
# GPU box, round 5 call 13: IC tests incl. large molecules, fix-up / sweep kernel times, phase stamps of the DMA sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c13; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "ic or IC or tail or kl_gradient" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -12 | tee $O/tests.txt
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd|total GPU" | cut -c1-160 | tee $O/kl_kernels.txt
BGK_LIB=$PWD/gpurun_variants/lib_icbts.so timeout 300 python tools/r05_icb_ts.py 2>&1 | tail -9 | tee $O/icb_ts.txt
