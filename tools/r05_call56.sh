#!/bin/bash
# GPU box, round 5 call 56: the width-256 kernel (hidden layers of 129 .. 256 units as one launch): parity tests, time against the
# layer-by-layer path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c56; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "256 or envelope or narrow_hidden or many_transformed" 2>&1 | tail -25 | tee $O/pytest.txt
timeout 300 python tools/r05_w256.py 2>&1 | tee $O/w256.txt
