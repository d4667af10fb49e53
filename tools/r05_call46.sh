#!/bin/bash
# GPU box, round 5 call 46: the recompute parity test with the tolerance form for the hardware-form VJP
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c46; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "recomputed" 2>&1 | tail -12 | cut -c1-600 | tee $O/pytest1.txt
