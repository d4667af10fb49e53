#!/bin/bash
# GPU box, round 5 call 50: the recompute backward under a mixed circular mask
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c50; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -k "recomputed" 2>&1 | tail -12 | cut -c1-300 | tee $O/pytest.txt
