#!/bin/bash
# GPU box: the -m gpu suite with the default build, then the kernel-trace A/B of gpurun_variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 tools/ab_variants.sh ab7
