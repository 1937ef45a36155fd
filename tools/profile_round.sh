#!/bin/bash
# One call on the GPU box: steady-state kernel stats (rocprofv3 --kernel-trace) and the PMC passes of the bench command,
# summarised there (the raw traces are too big to travel back), THEN the bench line — so that its roofline block reads the
# PMC summary of these very sources (pmc_stale: false).  usage: tools/profile_round.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
WL=config2
for a in "$@"; do case $a in config[1-5]) WL=$a;; esac; done
for a in "$@"; do case $a in --bf16-maps) WL=${WL}_bf16maps;; esac; done    # (bench._pmc_for keys the pass on the kind of feature maps)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof/r_results.db $OUT/kernel_stats.csv
rm -rf $OUT/prof
bash tools/pmc_pass.sh $TAG/pmc $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@"
python tools/pmc_summary.py $OUT/pmc $OUT/pmc.json > $OUT/pmc_top.txt
rm -rf $OUT/pmc
cp $OUT/pmc.json profiles/${TAG}_bench_${WL}_pmc.json        # (bench.py reads the newest profiles/*_bench_<workload>[_bf16maps]_pmc.json)
timeout -k 10 300 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
