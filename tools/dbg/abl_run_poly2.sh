#!/bin/bash
# dev: fp32 polyphase backward-data layers (and their forward twins) over ablation libraries, every dense block shape listed.
# usage: abl_run_poly2.sh <outdir> name ...
out=gpurun_out/$1; shift
mkdir -p $out
for v in base "$@"; do
  if [ $v = base ]; then L=""; else L="$PWD/transtacos-retunegan_amd/librtg_dev_$v.so"; fi
  echo "== $v" >> $out/abl.log
  RTG_DEV_LIB=$L BD_ALL=1 BD_PICK=15,16,17,18 timeout -k 10 200 python tools/dbg/bench_dconv.py poly 2>&1 | grep "^poly\|^      8" | cut -c1-170 >> $out/abl.log
  RTG_DEV_LIB=$L BD_PICK=12 timeout -k 10 120 python tools/dbg/bench_dconv.py fwd 2>&1 | grep "^fwd" | cut -c1-170 >> $out/abl.log
done
cat $out/abl.log
