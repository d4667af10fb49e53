#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q "$@" 2>&1 | grep -v Warning | tail -60 > gpurun_out/r3t/pytest.txt
cat gpurun_out/r3t/pytest.txt
