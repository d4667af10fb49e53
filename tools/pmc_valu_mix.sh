#!/bin/bash
# GPU box: dynamic VALU instruction mix of a kernel in the default bench command.  usage: pmc_valu_mix.sh <workload> <kernel substring>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mix_$1
mkdir -p $OUT
CMD="python bench.py --workload $1 --no-cpu-baseline --no-extras --kl-steps 0 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_FMA_F16 --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
for d in a b c; do python tools/pmc_summary.py $OUT/$d $2 || tail -5 $OUT/$d.log; done
