#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3c5
timeout 900 python -m pytest tests -m gpu -x -q -k "tail or round3 or flow16 or cfg5 or goldens or smoke" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r3c5/pytest.txt
cat gpurun_out/r3c5/pytest.txt
python tools/dbg_tail.py > gpurun_out/r3c5/dbg_tail.txt 2>&1; tail -25 gpurun_out/r3c5/dbg_tail.txt
python bench.py --no-cpu-baseline --no-extras --kl-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('tail', r['block_ms'][-1], 'avg coupling', round(r['avg_launch_ms'],4), 'ms/step', round(d['ms_per_step'],3))"
