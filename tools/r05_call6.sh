#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c6; mkdir -p $O
timeout 600 python tools/r05_grad_persample_diag.py 8192 2>&1 | grep -v Warning | tail -60 > $O/persample.txt
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "any_bin" 2>&1 | tail -3 > $O/anybin.txt
cat $O/anybin.txt
