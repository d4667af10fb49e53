#!/bin/bash
# GPU box, round 5 call 2: the -m gpu suite (no -x), the KL-gradient attribution, KL step with the training chain on / off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|device kernels of one KL" | tail -30 > $O/suite.txt
cat $O/suite.txt
timeout 900 python tools/r05_klgrad_diag.py 8192 2>&1 | grep -v Warning | tail -20 | tee $O/klgrad_diag.txt
for ch in 1 0 1 0; do
  BGK_TRAIN_CHAIN=$ch timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   chain '$ch' kl", round(k["steps_per_s"],2), round(k["ms_per_step"],3), k.get("single_call",{}).get("steps_per_s"))' | tee -a $O/chain.txt
done
BGK_TRAIN_CHAIN=1 bash tools/prof_kl.sh > $O/kl_stats.txt 2>&1; head -30 $O/kl_stats.txt
