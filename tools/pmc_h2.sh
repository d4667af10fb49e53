cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BGK_GEMM=f16x2
OUT=gpurun_out/pmc_h2
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/sq1 -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/sq2 -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python tools/prof_layer.py fused-BA 1048576 3 > /dev/null 2>&1
for d in sq1 sq2 grbm; do python tools/pmc_summary.py $OUT/$d coupling; done
