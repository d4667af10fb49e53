#!/bin/bash
# GPU box: cfg 2 pass time of every gpurun_variants/lib_*.so (+ the in-tree library's round-3 kernel: BGK_AFFINE_NO_DMA=1), interleaved, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg2 or affine or stack or readme or running_logdet" --deselect tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch 2>&1 | tail -2
for rep in 1 2 3; do
for f in gpurun_variants/lib_*.so R03; do
  unset BGK_AFFINE_NO_DMA
  if [ "$f" = R03 ]; then export BGK_AFFINE_NO_DMA=1; lib=""; n=r03kernel; else lib=$PWD/$f; n=$(basename $f .so); fi
  BGK_LIB=$lib python bench.py --workload cfg2 --no-cpu-baseline --no-extras --steps 30 --warmup 5 --kl-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg2 $n', round(d['ms_per_step'],3), 'ms/step; kernel', round(r['avg_launch_ms'],4), 'ms; frac', round(r['step_view']['frac'],3))"
done; done
