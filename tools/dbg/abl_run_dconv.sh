#!/bin/bash
# dev: bench_dconv (bf16, a 2-D and a 1-D layer) over ablation libraries.  usage: abl_run_dconv.sh <outdir> name ...
out=gpurun_out/$1; shift
mkdir -p $out
for v in base "$@"; do
  if [ $v = base ]; then L=""; else L="$PWD/transtacos-retunegan_amd/librtg_dev_$v.so"; fi
  echo "== $v" >> $out/abl.log
  RTG_DEV_LIB=$L BD_BF=${BD_BF-1} BD_PICK=1,2 timeout -k 10 120 python tools/dbg/bench_dconv.py 2d 2>&1 | grep fwd2d | cut -c1-170 >> $out/abl.log
  RTG_DEV_LIB=$L BD_BF=${BD_BF-1} BD_PICK=0,11 timeout -k 10 120 python tools/dbg/bench_dconv.py fwd 2>&1 | grep "^fwd" | cut -c1-170 >> $out/abl.log
done
cat $out/abl.log
