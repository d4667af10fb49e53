#!/bin/bash
# GPU box, round 5 call 27: cfg 5 with the dead rows of the affine kernel's last output tile skipped (vs the previous build); the fix-up launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c27; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "affine or cfg5 or augment or ic_backward" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -5 | tee $O/tests.txt
c5() { BGK_LIB=$2 timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --no-extras --kl-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("   '"$1"'", round(d["ms_per_step"],4), "ms")' | tee -a $O/cfg5_ab.txt; }
for rep in 1 2 3; do
  c5 skip_dead_rows ""
  c5 previous $PWD/gpurun_variants/lib_prev.so
done
bash tools/prof_kl.sh 2>&1 | grep -E "steps_per_s|ic_ic2xyz_bwd" | cut -c1-160 | tee $O/kl_kernels.txt
