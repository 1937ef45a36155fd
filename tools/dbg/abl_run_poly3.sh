#!/bin/bash
# dev: dense-kernel variants over the 1-D layer shapes (stride-1 / stride-3 forward, polyphase backward-data), best dense
# block shape per problem.  usage: abl_run_poly3.sh <outdir> name ...
out=gpurun_out/$1; shift
mkdir -p $out
for v in base "$@"; do
  if [ $v = base ]; then L=""; else L="$PWD/transtacos-retunegan_amd/librtg_dev_$v.so"; fi
  echo "== $v" >> $out/abl.log
  RTG_DEV_LIB=$L BD_PICK=0,3,4,7,8,11,12,13,14,15,16,17,18,19,20,21,22 timeout -k 10 300 python tools/dbg/bench_dconv.py fwd dgrad poly 2>&1 | grep "^poly\|^fwd\|^dgrad" | cut -c1-170 >> $out/abl.log
done
cat $out/abl.log
