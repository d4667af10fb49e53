#!/bin/bash
# GPU box, round 5 call 29: bgk_dense_backward_dx with the gradient column mask moved to the point of use and asm LDS fragment reads
# (no compiler-inserted vmcnt(0) between the next group's requests and the current group's matrix work): parity, stamps, KL step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c29; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dx or train or kl_gradient or chain or narrow or backward" 2>&1 | tail -5 | tee $O/pytest.txt
BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -18 | tee $O/dx_ts.txt
bash tools/ab_kl.sh dxold base dxcpp dxold base 2>&1 | tee $O/ab.txt
