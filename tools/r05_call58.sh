#!/bin/bash
# GPU box, round 5 call 58: bgk_dense_layer tests (fixed), kernel trace of the layer-by-layer path at hidden width 256
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c58; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "dense_layer or densenet_layers or readme_flow" 2>&1 | tail -25 | tee $O/pytest_new.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o w -- python tools/r05_w256.py 1048576 5 > $O/w256_under_rocprof.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:100], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
