#!/bin/bash
# GPU box: cfg 5 pass time with each library variant (tools/build_variants.sh), same box
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="gpurun_variants/lib_$v.so"; fi
  BGK_LIB=$lib python bench.py --workload cfg5 --no-cpu-baseline --no-extras --steps 10 --warmup 3 --kl-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', round(d['ms_per_step'],3), 'ms/step', [round(x,3) for x in r['block_ms']])"
done
