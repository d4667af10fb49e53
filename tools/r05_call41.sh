#!/bin/bash
# GPU box, round 5 call 41: recompute backward with its inputs requested a slot ahead; spline VJP with the knots on the hardware forms
# (lib_fast: -DBGK_VJP_FAST=1 for the recompute kernel and bgk_rqs_backward); KL gradient accuracy with it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c41; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "recomputed" 2>&1 | tail -3 | tee $O/pytest1.txt
BGK_LIB=$PWD/gpurun_variants/lib_fast.so timeout 900 python -m pytest tests -m gpu -q -k "recomputed or kl_gradient or training_forward_gradients or spline_backward" 2>&1 | tail -12 | tee $O/pytest_fast.txt
for v in base fast; do
 for rc in 1 0; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  OUT=gpurun_out/ab_rc_${v}_$rc; rm -rf $OUT; mkdir -p $OUT
  BGK_LIB=$lib BGK_RECOMPUTE_PARAMS=$rc rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  echo "== $v recompute=$rc $(grep '"metric"' $OUT/log.txt | python -c 'import sys,json; print(json.loads(sys.stdin.read())["kl"]["steps_per_s"])')"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if any(k in r["Name"] for k in ("recompute","rqs_bwd","train_kernel","dx_kernel<1>","wgrad_kernel")): print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
 done
done 2>&1 | tee $O/ab.txt
