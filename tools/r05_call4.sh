#!/bin/bash
# GPU box, round 5 call 4 (first call after the container was re-created): the whole -m gpu suite with tracebacks of failures,
# the default bench line, a kernel trace of the KL step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short 2>&1 | grep -vE "^\s*$|Warning" | grep -E "passed|failed|^FAILED|^ERROR|^E |assert|KL gradient|device kernels|^tests/.*Error" | cut -c1-300 | tail -60 > $O/suite.txt
cat $O/suite.txt
timeout 600 python bench.py > $O/bench_plain.json 2>$O/bench_plain.err; tail -3 $O/bench_plain.err; head -c 3000 $O/bench_plain.json; echo
bash tools/prof_kl.sh > $O/kl_stats.txt 2>&1; head -40 $O/kl_stats.txt
cp $(find gpurun_out/prof_kl/stats -name "*kernel_stats.csv" | head -1) $O/kl_step_kernel_stats.csv
