#!/bin/bash
# GPU box, round 5 call 37: bgk_dense_backward_dx first GEMM, ring depths and workgroup width:
#   base = 4 waves, operand ring 2, gradient ring 3 | A = 4 waves, operand ring 3, gradient ring 2 | B = 8 waves, rings 3 / 3 | C = 8 waves, rings 2 / 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c37; mkdir -p $O
timeout 300 python tools/r05_dx_align.py save /tmp/a.pt 2>&1 | tail -3
for v in A B C; do
  BGK_LIB=$PWD/gpurun_variants/lib_$v.so timeout 300 python tools/r05_dx_align.py save /tmp/$v.pt 2>&1 | tail -3
  echo "== $v against base"; python tools/r05_dx_align.py cmp /tmp/$v.pt /tmp/a.pt | sort -g -k6 | tail -1 | tee -a $O/cmp.txt
done
for v in Ats Bts; do echo "== $v"; BGK_LIB=$PWD/gpurun_variants/lib_$v.so timeout 300 python tools/r05_dx_ts.py 17 2>&1 | tail -20 | tee $O/dx_ts_$v.txt; done
bash tools/ab_kl.sh base A B C base A B C 2>&1 | grep -v "h2v2\|wgrad\|rqs_bwd" | tee $O/ab.txt
