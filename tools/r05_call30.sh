#!/bin/bash
# GPU box, round 5 call 30: first GEMM of bgk_dense_backward_dx, the computing part split into mask + split | issue | k-steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c30; mkdir -p $O
BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -20 | tee $O/dx_ts.txt
