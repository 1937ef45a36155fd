#!/bin/bash
# dev: kernel durations of tools/dbg/bench_thin.py under rocprofv3 (event timing through ctypes is host-bound below ~15 us)
OUT=$GRAFT_REPO_ROOT/gpurun_out/thinprof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dbg in 0; do
export RTG_THIN_DBG=$dbg
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -f csv -- python3 $GRAFT_REPO_ROOT/tools/dbg/bench_thin.py > $OUT/run.log 2>&1
echo "dbg=$dbg"; grep -E "cout1_long|cin1_flat" $OUT/prof/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
cp $OUT/prof/*kernel_stats.csv $OUT/stats_$dbg.csv
rm -rf $OUT/prof
done
