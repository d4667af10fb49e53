#!/bin/bash
# GPU box, round 5 call 15: cfg 2 with the two-network pipeline: tests, A/B against the sequential networks and two interleave ratios
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c15; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "affine" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -8 | tee $O/tests.txt
c2() { timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extras --kl-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("   '"$1"'", round(d["ms_per_step"],4), "ms", round(d["value"]/1e6,1), "M samples/s")' | tee -a $O/cfg2_ab.txt; }
for rep in 1 2 3; do
  c2 pipeline_vpm7
  BGK_AFFINE_NO_PIPE2=1 c2 sequential
  BGK_LIB=$PWD/gpurun_variants/lib_vpm4.so c2 pipeline_vpm4
  BGK_LIB=$PWD/gpurun_variants/lib_vpm10.so c2 pipeline_vpm10
done
