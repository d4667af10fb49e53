#!/bin/bash
# GPU box, round 5 call 69: spline conditioners with 1 / 3 / 4 / 8 hidden layers as one launch (bgk_coupling_rqs_dense_deep): tests, times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c69; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "other_depths or deep_conditioner or envelope or 256 or narrow_hidden" 2>&1 | tail -12 | tee $O/pytest_new.txt
timeout 300 python tools/r05_deep.py 2>&1 | grep hidden | tee $O/deep.txt
