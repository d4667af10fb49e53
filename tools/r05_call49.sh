#!/bin/bash
# GPU box, round 5 call 49: bgk_dense_backward_dx after the removal of the unshipped first-GEMM variants: parity suite of the training path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c49; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dx or train or kl or chain or narrow or backward or trainer or recomputed" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("kl", d["kl"]["steps_per_s"])' | tee -a $O/pytest.txt
