#!/bin/bash
# GPU box, round 5 call 35: one-hot gradients through bgk_dense_backward_dx, LDS-gradient form against the committed one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c35; mkdir -p $O
timeout 300 python tools/r05_dx_onehot.py save /tmp/a.pt 2>&1 | tail -3
BGK_LIB=$PWD/gpurun_variants/lib_dxold.so timeout 300 python tools/r05_dx_onehot.py save /tmp/b.pt 2>&1 | tail -3
python tools/r05_dx_onehot.py cmp /tmp/a.pt /tmp/b.pt | head -80 | tee $O/cmp.txt
