#!/bin/bash
# kernel trace of the default bench (forked streams, graph replay) -> timeline of the last step
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/bench.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $OUT/prof/r_results.db gpurun_out/r06_timeline.txt
