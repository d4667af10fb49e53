#!/bin/bash
# copy the judged summaries of a profile round from gpurun_out/prof_<tag> into profiles/ (tracked)
TAG=${1:-r04}
cd "$(dirname "$0")/.."
S=gpurun_out/prof_$TAG
cp $S/stats/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $S/traffic.json profiles/${TAG}_traffic.json
cp $S/traffic.txt profiles/${TAG}_traffic.txt
cp $S/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $S/bench_plain.json profiles/${TAG}_bench_line.json
cp $S/bench_line_under_rocprof.json profiles/${TAG}_bench_line_under_rocprof.json
[ -s $S/bench_2rank_selftest.json ] && cp $S/bench_2rank_selftest.json profiles/${TAG}_bench_2rank_selftest_shared_gpu.json
for f in cfg2_kernel_stats.csv kl_step_kernel_stats.csv cfg2_traffic.json cfg3_kernel_stats.csv cfg3_line_under_rocprof.json; do [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f; done
for f in cfg2_stats.txt cfg2_pmc.txt kl_stats.txt kl_pmc.txt ic_tail_pmc.txt cfg5_pmc.txt bench_pmc_line.json; do [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f; done
ls -la profiles/ | grep $TAG
