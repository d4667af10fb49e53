#!/bin/bash
# rocprofv3 kernel stats of the KL training step (bench.py --kl-steps): run ON THE GPU BOX
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_kl
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 5 > $OUT/log.txt 2>&1
grep "\"metric\"" $OUT/log.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"kl\"])"
python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot/1e6)
for r in rows[:28]: print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms {r["Percentage"]:>6s}%')
PY
