#!/bin/bash
# GPU box, round 5 call 55: the final build of the round -- whole GPU suite, the default bench line, kernel stats of the KL step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c55; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
python bench.py > $O/bench_plain.json 2>$O/bench_plain.err; head -c 400 $O/bench_plain.json; echo
bash tools/prof_kl.sh > $O/kl_stats.txt 2>&1; head -14 $O/kl_stats.txt | cut -c1-300
cp $(find gpurun_out/prof_kl/stats -name "*kernel_stats.csv" | head -1) $O/kl_step_kernel_stats.csv
