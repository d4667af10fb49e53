#!/bin/bash
# GPU box, round 5 call 39: training forward without the parameter write-out + bgk_coupling_rqs_dense_h2_backward (parameters recomputed
# from z1): parity with the saved-parameter path, training / KL tests, KL step A/B (BGK_RECOMPUTE_PARAMS=0 = the saved-parameter path)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c39; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "recomputed or element_major" 2>&1 | tail -8 | tee $O/pytest1.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "train or kl or chain or narrow or backward or trainer" 2>&1 | tail -5 | tee $O/pytest2.txt
for v in 0 1 0 1; do
  OUT=gpurun_out/ab_rc_$v; rm -rf $OUT; mkdir -p $OUT
  BGK_RECOMPUTE_PARAMS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
  echo "== recompute=$v $(grep '"metric"' $OUT/log.txt | python -c 'import sys,json; print(json.loads(sys.stdin.read())["kl"]["steps_per_s"])')"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done 2>&1 | tee $O/ab.txt
