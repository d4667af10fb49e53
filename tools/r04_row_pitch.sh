#!/bin/bash
# GPU box: row pitch (floats) of the saved spline parameters (BGK_PARAM_PITCH) and of their gradients (BGK_ROW_PITCH); 4:4 = round-3 layout -- KL step rate
# and the averages of the kernels that read those rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 4:4 4:32 4:16 32:32 4:4 4:32; do
  export BGK_PARAM_PITCH=${v%%:*} BGK_ROW_PITCH=${v##*:}
  OUT=gpurun_out/pitch_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    for k in ("train_kernel","dense_bwd_dx_kernel<1>","rqs_bwd","wgrad_kernel"):
        if k in r["Name"]: out.append(f"{k} {float(r['AverageNs'])/1e3:.1f}")
print("   [pitch $v]", " | ".join(out))
PY
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; k=json.loads(sys.stdin.read())["kl"]; print("   kl", k["steps_per_s"], k["ms_per_step"])'
done
