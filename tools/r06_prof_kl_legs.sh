#!/bin/bash
# rocprofv3 kernel stats of the KL training step of cfg 2 / cfg 5 (tools/r06_kl_legs.py): run ON THE GPU BOX
#   tools/r06_prof_kl_legs.sh [steps]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=${1:-5}
for CFG in cfg2 cfg5; do
OUT=gpurun_out/prof_kl_$CFG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o kl -- python tools/r06_kl_legs.py $CFG $STEPS > $OUT/log.txt 2>&1
grep "^{" $OUT/log.txt > $OUT/line.json
F=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kl_${CFG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kl_${CFG}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=$STEPS+2
print("$CFG: total GPU ms", round(tot/1e6,2), "per step (", n, "steps incl. warm-ups)", round(tot/1e6/n,3))
for r in rows[:24]: print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["TotalDurationNs"])/1e3/int(r["Calls"]):9.1f} us {r["Percentage"]:>6s}%')
PY
done
