#!/bin/bash
# GPU box: HBM traffic + SQ counters of every kernel in the KL training step (bench.py --kl-steps), one counter pass each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_kl
mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --kl-steps 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- $CMD > /dev/null 2>&1
for k in dense_bwd_dx_kernel wgrad_kernel "coupling_rqs_dense_h2v2_train_kernel" coupling_rqs_bwd_recompute_kernel rqs_bwd_kernel; do
  echo "#### $k"
  for d in fetch write sq grbm; do python tools/pmc_summary.py $OUT/$d "$k" | grep -v "^void\|^(anon"; done
done
