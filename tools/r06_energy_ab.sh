#!/bin/bash
# GPU box: energy kernels, float4 rows (round 6) vs staging kernels (BGK_ENERGY_STAGED=1): parity test, then cfg 2's KL step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "energy" 2>&1 | tail -3
for v in 1 ""; do
  export BGK_ENERGY_STAGED=$v; [ -z "$v" ] && unset BGK_ENERGY_STAGED
  OUT=gpurun_out/energy_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python tools/r06_kl_legs.py cfg2 5 > $OUT/log.txt 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "energy" in r["Name"]: print("   [staged=$v]", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  timeout 300 python tools/r06_kl_legs.py cfg2 10 2>/dev/null | grep "^{" | cut -c100-260
done
