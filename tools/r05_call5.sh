#!/bin/bash
# GPU box, round 5 call 5: attribution of the KL gradient's distance from f64 (variants of the backward; stage by stage)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c5; mkdir -p $O
timeout 900 python tools/r05_klgrad_diag.py 8192 2>&1 | grep -v Warning | tail -24 | tee $O/klgrad_diag.txt
timeout 600 python tools/r05_grad_stage_diag.py 2>&1 | grep -v Warning | tail -70 | tee $O/stage_diag.txt
