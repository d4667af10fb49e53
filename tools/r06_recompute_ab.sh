#!/bin/bash
# GPU box: the three backward forms of cfg 2's couplings -- bgk_affine_backward + 2 x bgk_affine_net_backward64 (BGK_TAIL_FUSED64=0),
# ONE bgk_affine_coupling_backward64 on saved pre-activations (default), the same with nothing saved (BGK_RECOMPUTE64=1) --
# parity tests, then the KL step's kernels (5 steps at 2^20 under rocprofv3) and the plain step time for each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -5
for v in "separate" "onecall" "recompute"; do
  unset BGK_TAIL_FUSED64 BGK_RECOMPUTE64
  [ $v = separate ] && export BGK_TAIL_FUSED64=0
  [ $v = recompute ] && export BGK_RECOMPUTE64=1
  OUT=gpurun_out/rc64_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python tools/r06_kl_legs.py cfg2 5 > $OUT/log.txt 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("   [$v] GPU ms per step:", round(sum(float(r["TotalDurationNs"]) for r in rows)/1e6/7,3))
for r in rows[:5]: print("   [$v]", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  timeout 300 python tools/r06_kl_legs.py cfg2 10 2>/dev/null | grep "^{" | cut -c100-330
done
