#!/bin/bash
# GPU box, round 5 call 70+: the whole GPU suite, the smoke entry and the default bench line with the round's last build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c70; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench_plain.json 2>$O/bench_plain.err; python -c "
import json; d=json.loads(open('$O/bench_plain.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['kl']['steps_per_s'], d['cfg2']['hbm_view']['frac'])"
