#!/bin/bash
# dev: steady-state kernel stats of the default bench (summary only): dev_kstats.sh <tag>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof/r_results.db $OUT/kernel_stats.csv
rm -rf $OUT/prof
