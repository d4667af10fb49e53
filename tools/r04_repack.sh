#!/bin/bash
# GPU box: batched operand re-pack + deferred weight-gradient reduction -- parity tests, then the KL step with and without them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "batched_repack or spline_vjp or narrow" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "kltrainer or flat_adam or nll_training or fused_training or dense_backward or weight_grad or densenet_backward or training_gradients" 2>&1 | tail -4
for v in 1 0; do
BGK_BATCHED=$v timeout 200 python - <<PY
import os, json, subprocess, sys
from bgflow_amd import dense
dense.BATCHED_REPACK = dense.DEFERRED_WGRAD_REDUCE = os.environ["BGK_BATCHED"] == "1"
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "1", "--warmup", "1", "--kl-steps", "20"]
import runpy
runpy.run_path("bench.py", run_name="__main__")
PY
done 2>/dev/null | grep '"metric"' | python -c '
import sys, json
for l in sys.stdin:
    k = json.loads(l)["kl"]; print("kl", k["steps_per_s"], k["ms_per_step"], k.get("single_call"))'
