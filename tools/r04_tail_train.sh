#!/bin/bash
# GPU box: the one-launch training tail + hardware forms in the IC backward -- parity tests, then the KL step with and without the fused tail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "generation_tail_as_one or kl_gradient or batched_repack" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "ic_backward or kl_step_full or kltrainer or kl_gradient or nll_training or global_ic" 2>&1 | tail -4
for v in 1 0; do
BGK_FUSED_TAIL=$v timeout 200 python - <<PY
import os, sys
from bgflow_amd.flow import SequentialFlow
SequentialFlow.FUSE_TRAINING_TAIL = os.environ["BGK_FUSED_TAIL"] == "1"
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "1", "--warmup", "1", "--kl-steps", "20"]
import runpy
runpy.run_path("bench.py", run_name="__main__")
PY
done 2>/dev/null | grep '"metric"' | python -c '
import sys, json
for l in sys.stdin:
    k = json.loads(l)["kl"]; print("kl", k["steps_per_s"], k["ms_per_step"], k["loss"])'
