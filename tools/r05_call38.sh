#!/bin/bash
# GPU box, round 5 call 38: bgk_dense_backward_dx first GEMM, MFMAs of group g - 1 interleaved with the split and the requests of group g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c38; mkdir -p $O
timeout 300 python tools/r05_dx_align.py save /tmp/a.pt 2>&1 | tail -3
BGK_LIB=$PWD/gpurun_variants/lib_nopipe.so timeout 300 python tools/r05_dx_align.py save /tmp/b.pt 2>&1 | tail -3
python tools/r05_dx_align.py cmp /tmp/a.pt /tmp/b.pt | sort -g -k6 | tail -2 | tee $O/cmp.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "dx or train or kl_gradient or chain or narrow or backward" 2>&1 | tail -3 | tee $O/pytest.txt
BGK_LIB=$PWD/gpurun_variants/lib_ts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -20 | tee $O/dx_ts.txt
bash tools/ab_kl.sh nopipe base nopipe base 2>&1 | grep -v "h2v2\|wgrad\|rqs_bwd" | tee $O/ab.txt
