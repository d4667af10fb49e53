#!/bin/bash
# GPU box, round 5 call 1: the whole -m gpu suite (new tests of the round included, KL-gradient errors printed), then the KL step with
# 0 / 16 / 48 rows of padding between the halves of the [2, B, 128] allocations, then a kernel trace of the KL step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05c1
O=gpurun_out/r05c1
timeout 1500 python -m pytest tests -m gpu -q -s -x --deselect tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch 2>&1 | grep -E "passed|failed|error|Error|assert|KL gradient|^E " | tail -40 > $O/suite.txt
tail -15 $O/suite.txt
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -s -k "kl_gradient" 2>&1 | grep -E "passed|failed|KL gradient|^E " | tail -12 | tee $O/klgrad.txt
for pad in 0 16 48 0 16; do
  BGK_HALF_PAD_ROWS=$pad timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   pad '$pad' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3), k.get("single_call",{}).get("steps_per_s"))' | tee -a $O/pad.txt
done
bash tools/prof_kl.sh > $O/kl_stats.txt 2>&1; head -24 $O/kl_stats.txt
