#!/bin/bash
# GPU box, round 5 call 57: bgk_dense_layer (Linear layers of conditioners outside the one-launch envelope on a HIP kernel): its tests,
# the whole GPU suite (the layer-by-layer path changed under every test that uses it), the layer-by-layer timings again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c57; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "dense_layer or densenet_layers or readme_flow" 2>&1 | tail -25 | tee $O/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/pytest_all.txt
timeout 300 python tools/r05_w256.py 2>&1 | tee $O/w256.txt
