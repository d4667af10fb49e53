#!/bin/bash
# GPU box, round 5 call 33: bgk_dense_backward_dx, LDS-gradient form against the register form on aligned / unaligned gradient rows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c33; mkdir -p $O
timeout 300 python tools/r05_dx_align.py save /tmp/a.pt 2>&1 | tail -3
BGK_LIB=$PWD/gpurun_variants/lib_dxreg.so timeout 300 python tools/r05_dx_align.py save /tmp/b.pt 2>&1 | tail -3
python tools/r05_dx_align.py cmp /tmp/a.pt /tmp/b.pt | tee $O/cmp.txt
