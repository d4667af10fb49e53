#!/bin/bash
# rocprofv3 kernel stats of the cfg 2 (8 x affine coupling, dim 64) pass: run ON THE GPU BOX
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_cfg2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python bench.py --workload cfg2 --no-cpu-baseline --no-extras --steps 10 --warmup 2 --kl-steps 0 > $OUT/log.txt 2>&1
grep "\"metric\"" $OUT/log.txt | tail -1
python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot/1e6)
for r in rows[:14]: print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us {r["Percentage"]:>6s}%')
PY
