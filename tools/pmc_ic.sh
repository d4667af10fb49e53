#!/bin/bash
# GPU box: SQ counters of the fused generation tail (icdf + IC -> xyz) in the bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ic
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -o p -- python bench.py --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/sq2 -o p -- python bench.py --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python bench.py --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1 > /dev/null 2>&1
for d in sq1 sq2 grbm; do python tools/pmc_summary.py $OUT/$d ic2xyz; done
