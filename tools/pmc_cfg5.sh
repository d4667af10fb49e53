#!/bin/bash
# GPU box: SQ counters of the affine (and, for comparison, the spline) coupling kernels in the cfg 5 bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_cfg5
mkdir -p $OUT
CMD="python bench.py --workload cfg5 --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/sq2 -o p -- $CMD > /dev/null 2>&1
for k in affine_dense_v2 h2v2; do for d in sq1 sq2; do python tools/pmc_summary.py $OUT/$d $k; done; done
