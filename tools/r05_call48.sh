#!/bin/bash
# GPU box, round 5 call 48: recompute backward on every chunk shape; the all-samples f64 KL gradient check with the shipped path
# (parameters recomputed, hardware-form VJP)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c48; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -k "recomputed" 2>&1 | tail -4 | tee $O/pytest1.txt
timeout 1200 python -m pytest tests/test_gpu_slow.py -m gpu_slow -q -s 2>&1 | grep -E "flat KL|passed|failed|Error" | tee $O/slow.txt
