#!/bin/bash
# GPU box: instruction-cache counters of a kernel in the cfg 2 / cfg 3 bench command.  usage: pmc_icache.sh <workload> <kernel substring>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_icache_$1
mkdir -p $OUT
CMD="python bench.py --workload $1 --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/ic -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH_LEVEL SQ_BUSY_CYCLES --output-format csv -d $OUT/w -o p -- $CMD > /dev/null 2>&1
for d in ic w; do python tools/pmc_summary.py $OUT/$d $2; done
