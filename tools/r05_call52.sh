#!/bin/bash
# GPU box, round 5 call 52: recompute backward with the slot's gradient maximum formed inside the slot (no registers parked in scratch):
# ring depth 4 (base) | 6 | 8, ring start in front of the VJP (depth 4 | 6); parity, stamps, KL step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c52; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "recomputed or kl_gradient or training_forward_gradients" 2>&1 | tail -3 | tee $O/pytest1.txt
BGK_LIB=$PWD/gpurun_variants/lib_ts.so timeout 300 python tools/r05_rc_ts.py 2>&1 | tail -10 | tee $O/rc_ts.txt
for v in base rd6 rd8 rd4e rd6e base rd6 rd8 rd4e rd6e; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  OUT=gpurun_out/ab_rc_${v}; rm -rf $OUT; mkdir -p $OUT
  BGK_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  echo "== $v $(grep '"metric"' $OUT/log.txt | python -c 'import sys,json; print(json.loads(sys.stdin.read())["kl"]["steps_per_s"])')"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if any(k in r["Name"] for k in ("recompute","dx_kernel<1>")): print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done 2>&1 | tee $O/ab.txt
