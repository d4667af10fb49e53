#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of the default bench workload -> gpurun_out/<tag>/traffic.{json,txt}
TAG=${1:-traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --kl-steps 0 --no-extras --steps 2 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
python tools/traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/traffic.json > $OUT/traffic.txt
cat $OUT/traffic.txt
