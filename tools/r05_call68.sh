#!/bin/bash
# GPU box, round 5 call 68: bgk_dense_layer's tile scale from the finite entries only; its tests, then the whole GPU suite and the smoke entry
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c68; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "dense_layer or densenet_layers or readme_flow or packer" 2>&1 | tail -4 | tee $O/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
