#!/bin/bash
# GPU box, round 5 call 28: inside the first GEMM of bgk_dense_backward_dx: waiting for the wave's own requests | at the barrier | computing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c28; mkdir -p $O
BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -18 | tee $O/dx_ts.txt
