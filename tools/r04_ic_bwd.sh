#!/bin/bash
# GPU box: IC backward kernel variants -- parity tests, kernel stats and KL step with the register kernel and with the LDS-row kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu -k "ic_backward or kl_step_full or global_ic or generation_tail_as_one or kl_gradient" 2>&1 | tail -3
for v in "" 1; do
  export BGK_IC_BWD_LDS=$v; [ -z "$v" ] && unset BGK_IC_BWD_LDS
  OUT=gpurun_out/icb_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ic2xyz" in r["Name"] or "cdf" in r["Name"]: print("   [$v]", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   kl", k["steps_per_s"], k["ms_per_step"])'
done
