#!/bin/bash
# GPU box, round 5 call 71: affine couplings with 1 / 4 / 5 / 8 hidden layers as one launch (bgk_coupling_affine_dense_deep; the README flow): tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c71; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "affine or readme or goldens or other_depths" 2>&1 | tail -15 | tee $O/pytest_new.txt
