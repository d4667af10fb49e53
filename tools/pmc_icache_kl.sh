#!/bin/bash
# GPU box: instruction-cache counters of the KL-step kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_icache_kl
mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 2"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/ic -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/vm -o p -- $CMD > /dev/null 2>&1
for k in dense_bwd_dx_kernel rqs_dense_h2v2_train rqs_bwd_kernel; do echo "#### $k"; for d in ic vm; do python tools/pmc_summary.py $OUT/$d "$k" | grep -v "^void\|^(anon"; done; done
