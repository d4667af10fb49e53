#!/bin/bash
# GPU box: the one-launch spline backward (bgk_spline_backward_dx) -- its parity tests, then the KL step with and without it
# (same box, rocprofv3 kernel trace: step rate + per-kernel averages)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "spline_vjp or kl_gradient or narrow" 2>&1 | tail -5 || exit 1
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rqs_backward or fused_training or kl_gradient or nll_training or kltrainer" 2>&1 | tail -5 || exit 1
for v in 1 0; do
  OUT=gpurun_out/spline_bwd_$v; rm -rf $OUT; mkdir -p $OUT
  BGK_SPLINE_BACKWARD_FUSED=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  echo "== fused=$v $(grep '"metric"' $OUT/log.txt | python -c 'import sys,json; print(json.loads(sys.stdin.read())["kl"]["steps_per_s"])')"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]: print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  BGK_SPLINE_BACKWARD_FUSED=$v timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 20 2>/dev/null | grep '"metric"' | python -c 'import sys,json; print("   unprofiled", json.loads(sys.stdin.read())["kl"])'
done
