#!/bin/bash
# Bench lines (with the live roofline block) of the other single-GPU workloads + the serialized kernel stats of the
# headline workload, for profiles/.  usage: tools/profile_configs.sh <tag>
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for w in config3 config4 config5; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.loads(open('$OUT/bench_$w.json').read()); print('$w', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
RTG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench_serial.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/prof/r_results.db $OUT/serial_kernel_stats.csv
rm -rf $OUT/prof
ls $OUT
