#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3c6
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -8 > gpurun_out/r3c6/pytest.txt
cat gpurun_out/r3c6/pytest.txt
( time timeout 900 python bench.py > gpurun_out/r3c6/bench.json 2> gpurun_out/r3c6/bench.err ) 2>&1 | grep real
tail -3 gpurun_out/r3c6/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c6/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"avg launch",r["avg_launch_ms"],"frac",r["frac"], d["dtype"])
print("block_ms",r["block_ms"], "sum", sum(r["block_ms"]))
print("inverse", d.get("inverse"))
c=d.get("cpu_baseline",{})
print("cpu value", c.get("value"), c.get("cores"), {k:(v.get("value"), v.get("slowest_pass_s")) for k,v in c.get("torch_cpu_configurations",{}).items() if v})
print("cpu inverse", c.get("inverse",{}).get("value"), "kl", {k:v for k,v in c.get("kl",{}).items() if k!='sample'})
print("parity", json.dumps(c.get("parity_sample",{}).get("dlogp_rel_vs_f64_oracle")), c.get("parity_sample",{}).get("bin_index_differences"))
for k in ("exact_f32_mode","cfg2","cfg5","kl"):
    v=d.get(k); print(k, {kk:vv for kk,vv in (v or {}).items() if kk in ("value","ms_per_step","steps_per_s","error","f32","bf16")})
PY
BGK_BENCH_TEST_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 262144 --kl-steps 2 --kl-batch 65536 > gpurun_out/r3c6/bench_2rank.json 2>gpurun_out/r3c6/bench_2rank.err; tail -2 gpurun_out/r3c6/bench_2rank.err; python -c "
import json; d=json.loads(open('gpurun_out/r3c6/bench_2rank.json').read().strip().splitlines()[-1]); print('2-rank selftest', d['n_gpus'], d['value'], json.dumps(d.get('rccl'))[:600])"
