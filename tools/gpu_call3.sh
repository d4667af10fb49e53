#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3c3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3c3/pytest.txt
cat gpurun_out/r3c3/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3c3/bench.json 2> gpurun_out/r3c3/bench.err
tail -5 gpurun_out/r3c3/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c3/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"avg launch",r["avg_launch_ms"],"frac",r["frac"])
print("block_ms",r["block_ms"], "sum", sum(r["block_ms"]))
for k in ("exact_f32_mode","cfg2","cfg5","kl"):
    v=d.get(k); print(k, {kk:vv for kk,vv in (v or {}).items() if kk in ("value","ms_per_step","steps_per_s","error","f32","bf16")})
PY
