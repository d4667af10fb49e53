#!/bin/bash
# GPU box: SQ counters + HBM traffic of the sampling-tail kernels (register-resident and LDS-table generations) -- tools/dbg_tail.py --time-only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_tail
mkdir -p $OUT
CMD="python tools/dbg_tail.py --time-only"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/sq2 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_WRITE_sum --output-format csv -d $OUT/tcc -o p -- $CMD > /dev/null 2>&1
for d in sq1 sq2 grbm fetch write tcc; do python tools/pmc_summary.py $OUT/$d ic2xyz; done
