#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of the default bench command + HBM traffic counters
# of the dominant kernel.  Outputs land in gpurun_out/prof_$1; copy the summaries to profiles/ afterwards.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep '"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json
# PMC passes (separate runs, counters only): FETCH_SIZE / WRITE_SIZE (KiB units; gfx950: FETCH_SIZE reads 1/2 of wide coalesced streams)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_rqs -o p -- python tools/prof_layer.py rqs-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_rqs -o p -- python tools/prof_layer.py rqs-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
for d in pmc_fetch pmc_write pmc_fetch_rqs pmc_write_rqs pmc_sq pmc_grbm; do echo "== $d"; python tools/pmc_summary.py $OUT/$d | grep -A12 "coupling_rqs\|rqs_kernel" ; done > $OUT/pmc_summary.txt
python bench.py > $OUT/bench_plain.json 2>/dev/null
head -c 600 $OUT/bench_plain.json; echo; cat $OUT/pmc_summary.txt | head -60
python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
