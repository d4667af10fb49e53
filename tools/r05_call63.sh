#!/bin/bash
# GPU box, round 5 call 63: device-side packer of bgk_dense_layer; whole GPU suite; default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c63; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "dense_layer or densenet_layers or readme_flow or packer" 2>&1 | tail -8 | tee $O/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_all.txt
python bench.py > $O/bench_plain.json 2>$O/bench_plain.err; head -c 300 $O/bench_plain.json; echo
