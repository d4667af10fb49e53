#!/bin/bash
# GPU box: L2 request counters of one fused spline coupling layer (B|A, d = 17, B = 2^20): how many bytes the operand stream pulls
# through L2 -> L1 per launch (every wave streams all packed weights for its own 32-sample tile)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_l2; rm -rf $OUT; mkdir -p $OUT
for C in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  n=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$n -o p -- python tools/prof_layer.py fused-BA 1048576 3 > $OUT/$n.log 2>&1
  echo "== $C"; python tools/pmc_summary.py $OUT/$n coupling 2>&1 | tail -6
done
