#!/bin/bash
# GPU box, round 5 call 34: bgk_dense_backward_dx, LDS-gradient form (asm LDS reads ordered by a wait that names their registers):
# outputs against the committed form on aligned / unaligned rows, parity tests, stamps, KL step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c34; mkdir -p $O
timeout 300 python tools/r05_dx_align.py save /tmp/a.pt 2>&1 | tail -3
BGK_LIB=$PWD/gpurun_variants/lib_dxold.so timeout 300 python tools/r05_dx_align.py save /tmp/b.pt 2>&1 | tail -3
BGK_LIB=$PWD/gpurun_variants/lib_dxreg.so timeout 300 python tools/r05_dx_align.py save /tmp/c.pt 2>&1 | tail -3
BGK_LIB=$PWD/gpurun_variants/lib_dxring2.so timeout 300 python tools/r05_dx_align.py save /tmp/d.pt 2>&1 | tail -3
python tools/r05_dx_align.py cmp /tmp/a.pt /tmp/b.pt | sort -g -k6 | tail -4 | tee $O/cmp.txt
python tools/r05_dx_align.py cmp /tmp/c.pt /tmp/b.pt | sort -g -k6 | tail -2 | tee -a $O/cmp.txt
python tools/r05_dx_align.py cmp /tmp/d.pt /tmp/b.pt | sort -g -k6 | tail -2 | tee -a $O/cmp.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "dx or train or kl_gradient or chain or narrow or backward" 2>&1 | tail -5 | tee $O/pytest.txt
BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -20 | tee $O/dx_ts.txt
bash tools/ab_kl.sh dxold base dxring2 dxreg dxold base 2>&1 | grep -v "h2v2\|wgrad\|rqs_bwd" | tee $O/ab.txt
