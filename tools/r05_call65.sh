#!/bin/bash
# GPU box, round 5 call 65: the profile round again with the round's final build (tools/profile_round.sh r05), then the whole GPU suite and the smoke entry
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
tail -30 gpurun_out/profile_round_r05.log | cut -c1-250
O=gpurun_out/r05c65; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
