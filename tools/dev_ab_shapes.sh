#!/bin/bash
# dev: same-box A/B of library builds over the single-shape micro-benchmarks: dev_ab_shapes.sh <out dir> <lib> ...
out=$1; shift
mkdir -p $out
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    RTG_DEV_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_conv.py 2>/dev/null | grep "^fwd" | sed "s/^/$name r$round /" >> $out/shapes.txt || exit 1
  done
done
