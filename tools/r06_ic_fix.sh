#!/bin/bash
# GPU box: the IC backward's fix-up launch, lane-per-sample (round 5; BGK_IC_FIX_LANES=1) against wave-per-sample (round 6) -- parity
# tests, then the kernels' averages over 5 KL steps at 2^18 under rocprofv3 and the KL leg itself
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -m gpu -k "ic_backward or kl_gradient or kl_step" 2>&1 | tail -3
for v in "" 1; do
  export BGK_IC_FIX_LANES=$v; [ -z "$v" ] && unset BGK_IC_FIX_LANES
  OUT=gpurun_out/icfix_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ic2xyz" in r["Name"] or "cdf_bwd" in r["Name"]: print("   [lanes_only=$v]", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   kl", k["steps_per_s"], k["ms_per_step"])'
done
