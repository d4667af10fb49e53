#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of the default bench command + HBM traffic / SQ counters
# of the dominant kernel.  Outputs land in gpurun_out/prof_$1; copy the summaries to profiles/ afterwards (tools/profile_copy.sh).
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
SHORT="--no-cpu-baseline --kl-steps 0"
# 1. kernel trace + stats of the bench command (default flags except the CPU leg / KL extra, which launch no hot-path kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py $SHORT > $OUT/bench_under_rocprof.log 2>&1
grep '"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json
# 1b. the headline workload alone (no cfg 2 / cfg 5 / exact-f32 legs): the dominant kernel's average here is the one bench.py's
#     roofline.avg_launch_ms must agree with (the full command's average mixes in cfg 5's wider layers)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -o cfg3 -- python bench.py $SHORT --no-extras > $OUT/cfg3_under_rocprof.log 2>&1
grep '"metric"' $OUT/cfg3_under_rocprof.log > $OUT/cfg3_line_under_rocprof.json
cp $(find $OUT/stats_cfg3 -name "*kernel_stats.csv" | head -1) $OUT/cfg3_kernel_stats.csv
# 2. PMC passes (separate runs, counters only) over the headline workload, 2 steps: HBM traffic of every kernel
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python bench.py $SHORT --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python bench.py $SHORT --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
python tools/traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/traffic.json > $OUT/traffic.txt
# 3. SQ / GRBM counters of one fused layer (B|A, d = 17), shipped mode and exact-f32 mode
for MODE in f16x2 f32; do
  export BGK_GEMM=$MODE
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq_$MODE -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/pmc_sq2_$MODE -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm_$MODE -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
done
unset BGK_GEMM
for d in pmc_sq_f16x2 pmc_sq2_f16x2 pmc_grbm_f16x2 pmc_sq_f32 pmc_sq2_f32 pmc_grbm_f32; do echo "== $d"; python tools/pmc_summary.py $OUT/$d coupling; done > $OUT/pmc_summary.txt
# 3b. cfg 2 (fused affine kernel): kernel stats + counters; KL training step: kernel stats + counters; fused generation tail counters
bash tools/prof_cfg2.sh > $OUT/cfg2_stats.txt 2>&1
cp gpurun_out/prof_cfg2/stats/*/c2_kernel_stats.csv $OUT/cfg2_kernel_stats.csv 2>/dev/null || cp $(find gpurun_out/prof_cfg2/stats -name "*kernel_stats.csv" | head -1) $OUT/cfg2_kernel_stats.csv
bash tools/pmc_cfg2.sh > $OUT/cfg2_pmc.txt 2>&1
C2="python bench.py --workload cfg2 --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cfg2 -o p -- $C2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_cfg2 -o p -- $C2 > /dev/null 2>&1
python tools/traffic_json.py $OUT/pmc_fetch_cfg2 $OUT/pmc_write_cfg2 $OUT/cfg2_traffic.json > $OUT/cfg2_traffic.txt
bash tools/prof_kl.sh > $OUT/kl_stats.txt 2>&1
cp $(find gpurun_out/prof_kl/stats -name "*kernel_stats.csv" | head -1) $OUT/kl_step_kernel_stats.csv
bash tools/pmc_kl.sh > $OUT/kl_pmc.txt 2>&1
bash tools/pmc_ic.sh > $OUT/ic_tail_pmc.txt 2>&1
# 4. un-profiled bench line (full default command incl. cpu_baseline + KL extra) and the multi-rank self-test of bench.py
python bench.py > $OUT/bench_plain.json 2>$OUT/bench_plain.err
BGK_BENCH_TEST_SHARED_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 262144 --kl-steps 2 --kl-batch 65536 2>$OUT/bench_2rank_selftest.err | grep "\"metric\"" > $OUT/bench_2rank_selftest.json
# 5. roofline.traffic measured inside the bench run (bench.py --pmc: its own rocprofv3 --pmc passes), cfg 5's two coupling kernels side by side
python bench.py --pmc --no-cpu-baseline --no-extras --kl-steps 0 > $OUT/bench_pmc_line.json 2>$OUT/bench_pmc_line.err
bash tools/pmc_cfg5.sh > $OUT/cfg5_pmc.txt 2>&1
head -c 600 $OUT/bench_plain.json; echo
cat $OUT/traffic.txt
python - <<PY
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print("  ", r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
