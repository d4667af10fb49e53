#!/bin/bash
# GPU box, round 5 call 32: the failing parity case of call 31, in full
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_kl_gradient_spline_couplings" 2>&1 | tail -40 | tee $O/pytest.txt
