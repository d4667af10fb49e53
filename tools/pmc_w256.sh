#!/bin/bash
# GPU box: HBM traffic and SQ counters of the width-256 coupling kernel and of bgk_dense_layer (one B|A layer, hidden width 256, 2^20 samples)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_w256
mkdir -p $OUT
for path in fused layer; do
  CMD="python tools/r05_w256_one.py 256 $path"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${path}_fetch -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${path}_write -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/${path}_sq1 -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${path}_sq2 -o p -- $CMD > /dev/null 2>&1
done
for k in w256 dense_layer_kernel; do
  p=fused; [ $k = dense_layer_kernel ] && p=layer
  echo "==== $k"
  for d in fetch write sq1 sq2; do python tools/pmc_summary.py $OUT/${p}_$d $k; done
done
