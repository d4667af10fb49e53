#!/bin/bash
# GPU box: the round-end sequence the driver runs -- the -m gpu suite, smoke(), the default bench line (all under hard timeouts)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 420 python bench.py ) > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; tail -4 gpurun_out/bench_final.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().split("\n")[0])
print("value", d["value"], "ms", d["ms_per_step"], "avg", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "inv", d["inverse"]["value"], "kl", d["kl"]["steps_per_s"], "cfg2", d["cfg2"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
