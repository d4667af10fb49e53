#!/bin/bash
# GPU box: per-kernel average durations of the coupling kernels (both generations, 3 layer kinds, both directions) under
# rocprofv3 --kernel-trace --stats.  usage: tools/prof_v2.sh <outdir-tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- python tools/dev_v2.py --time > $OUT/time.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
