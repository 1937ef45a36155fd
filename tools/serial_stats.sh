#!/bin/bash
# Serialized (RTG_STREAMS=0) steady-state kernel stats of a bench workload: a launch has the chip to itself, so the per-kernel
# averages are the kernels' own durations.  usage: tools/serial_stats.sh <tag> [bench args]   -> gpurun_out/<tag>_serial_kernel_stats.csv
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RTG_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $OUT/${TAG}_serial.json 2> $OUT/${TAG}_serial.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof_$TAG/r_results.db $OUT/${TAG}_serial_kernel_stats.csv --skip-steps 2
rm -rf $OUT/prof_$TAG
