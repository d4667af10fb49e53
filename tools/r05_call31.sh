#!/bin/bash
# GPU box, round 5 call 31: first GEMM of bgk_dense_backward_dx with the gradient tile copied through LDS by DMA (ring of 3 | 2 groups)
# against the register form: parity, stamps, KL step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c31; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dx or train or kl_gradient or chain or narrow or backward" 2>&1 | tail -5 | tee $O/pytest.txt
BGK_LIB=$PWD/gpurun_variants/lib_dxts.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -20 | tee $O/dx_ts.txt
bash tools/ab_kl.sh dxreg base dxring2 dxreg base 2>&1 | grep -v "h2v2\|wgrad\|rqs_bwd" | tee $O/ab.txt
