#!/bin/bash
# PMC passes of bench.py summarised on the GPU box.  usage: tools/pmc_round.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh $TAG/pmc $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@"
python tools/pmc_summary.py $OUT/pmc $OUT/pmc.json > $OUT/pmc_top.txt
rm -rf $OUT/pmc
