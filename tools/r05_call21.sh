#!/bin/bash
# GPU box, round 5 call 21: the whole -m gpu suite (-x, as the driver runs it), smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c21; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E " | cut -c1-300 | tail -8 | tee $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_plain.json 2>$O/bench_plain.err; tail -2 $O/bench_plain.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05c21/bench_plain.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_launch_ms", d["roofline"]["avg_launch_ms"], "mfma", d["roofline"]["mfma_util"]["frac"])
print("inverse", d["inverse"]["value"], "cfg2", d["cfg2"]["ms_per_step"], d["cfg2"]["hbm_view"]["frac"], "cfg5", d["cfg5"]["f32"]["ms_per_step"], d["cfg5"]["f32"]["hbm_view_frac"], d["cfg5"]["bf16"]["ms_per_step"])
print("kl", d["kl"]["steps_per_s"], d["kl"]["ms_per_step"], "single", d["kl"]["single_call"]["steps_per_s"], "exact", d["exact_f32_mode"]["ms_per_step"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
