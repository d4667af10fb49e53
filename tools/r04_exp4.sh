#!/bin/bash
# GPU box: affine / round-3 suites + cfg 5 and cfg 3 pass times, in-tree library against gpurun_variants/lib_r03.so (the round-3 kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -x -q -k "affine or cfg5 or augment or several_conditioning or full_size or envelope" 2>&1 | tail -3
for w in cfg5 cfg3; do
for v in r03 base r03 base; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  BGK_LIB=$lib python bench.py --workload $w --no-cpu-baseline --no-extras --steps 10 --warmup 3 --kl-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w $v', round(d['ms_per_step'],3), 'ms/step', [round(x,3) for x in r['block_ms']])"
done; done
