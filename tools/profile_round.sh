#!/bin/bash
# One call on the GPU box: bench line, steady-state kernel stats (rocprofv3 --kernel-trace) and the PMC passes of the same
# command, summarised there (the raw traces are too big to travel back).  usage: tools/profile_round.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof/r_results.db $OUT/kernel_stats.csv
rm -rf $OUT/prof
bash tools/pmc_pass.sh $TAG/pmc $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@"
python tools/pmc_summary.py $OUT/pmc $OUT/pmc.json > $OUT/pmc_top.txt
rm -rf $OUT/pmc
ls -la $OUT
