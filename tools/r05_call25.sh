#!/bin/bash
# GPU box, round 5 call 25: single-pass kldiv (target energy + loss sums inside the generation tail's launch): tests, bench KL legs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c25; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "kl or KL or tail or train" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -8 | tee $O/tests.txt
for rep in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   kl", round(k["steps_per_s"],2), "single_call", k["single_call"].get("steps_per_s"), "single_pass", k.get("single_pass",{}).get("steps_per_s"), k.get("single_pass",{}).get("error"), k["single_call"].get("loss"), k.get("single_pass",{}).get("loss"))' | tee -a $O/kl.txt; done
