#!/bin/bash
# GPU box, round 5 call 47: recompute backward with the next chunk's GEMM threaded through the VJP's hook points (default) against
# the GEMM behind the VJP (lib_nothread) and the saved-parameter path: parity, stamps, KL step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c47; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "recomputed or kl_gradient or training_forward_gradients" 2>&1 | tail -4 | tee $O/pytest1.txt
BGK_LIB=$PWD/gpurun_variants/lib_ts.so timeout 300 python tools/r05_rc_ts.py 2>&1 | tail -10 | tee $O/rc_ts.txt
for v in base:0 base:1 nothread:1 base:0 base:1 nothread:1; do
  rc=${v#*:}; v=${v%:*}
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  OUT=gpurun_out/ab_rc_${v}_$rc; rm -rf $OUT; mkdir -p $OUT
  BGK_LIB=$lib BGK_RECOMPUTE_PARAMS=$rc rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  echo "== $v recompute=$rc $(grep '"metric"' $OUT/log.txt | python -c 'import sys,json; print(json.loads(sys.stdin.read())["kl"]["steps_per_s"])')"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if any(k in r["Name"] for k in ("recompute","rqs_bwd","train_kernel","dx_kernel<1>")): print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done 2>&1 | tee $O/ab.txt
