#!/bin/bash
# GPU box, round 5 call 51: counters of the KL step's kernels with the recompute backward in (HBM traffic, SQ)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_kl.sh > gpurun_out/r05_kl_pmc.txt 2>&1; tail -60 gpurun_out/r05_kl_pmc.txt
