#!/bin/bash
# GPU box: sampling-tail block time (bench.py roofline.block_ms[-1]) with each library variant, same box
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="gpurun_variants/lib_$v.so"; fi
  BGK_LIB=$lib python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --kl-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'tail', r['block_ms'][-1], 'avg coupling', round(r['avg_launch_ms'],4), 'ms/step', round(d['ms_per_step'],3))"
done
