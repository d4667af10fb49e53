#!/bin/bash
# GPU box, round 5 call 54: training forward instance without the parameter write-out code (1 spilled register instead of 18): parity of
# the training path, KL step A/B against the write-out instance skipping its stores at run time (lib_savep)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c54; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "train or kl or chain or narrow or trainer or recomputed or element_major" 2>&1 | tail -3 | tee $O/pytest1.txt
for v in savep base savep base; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  OUT=gpurun_out/ab_rc_${v}; rm -rf $OUT; mkdir -p $OUT
  BGK_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  echo "== $v $(grep '"metric"' $OUT/log.txt | python -c 'import sys,json; print(json.loads(sys.stdin.read())["kl"]["steps_per_s"])')"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if any(k in r["Name"] for k in ("recompute","train_kernel")): print("   ", r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done 2>&1 | tee $O/ab.txt
