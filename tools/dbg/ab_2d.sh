#!/bin/bash
# dev: the 2-D layer bench (fp32 and bf16) with the regular library and a dev library on one box
dev=$1
for L in "" "$dev"; do
  echo "== ${L:-regular}"
  RTG_DEV_LIB=$L BD_PICK=1,2,4 timeout -k 10 200 python tools/dbg/bench_dconv.py 2d 2>&1 | grep fwd2d | cut -c1-170
  RTG_DEV_LIB=$L BD_BF=1 BD_PICK=1,2,4 timeout -k 10 200 python tools/dbg/bench_dconv.py 2d 2>&1 | grep fwd2d | cut -c1-170
done
