#!/bin/bash
# GPU box, round 5 call 66: scheduler options of the compiler on bgk_fused2.hip (the dominant kernel's unpinned regions): headline step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c66; mkdir -p $O
for v in base maxilp trk base maxilp trk; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/gpurun_variants/lib_$v.so"; fi
  BGK_LIB=$lib python bench.py --no-cpu-baseline --no-extras --kl-steps 0 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('   $v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
done | tee $O/ab.txt
