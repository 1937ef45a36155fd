#!/bin/bash
# dev: same-box A/B of library builds over the default bench: dev_ab.sh <out dir> <lib> [<lib> ...]   (two rounds each)
out=$1; shift
mkdir -p $out
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    RTG_DEV_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$name round $round', d['ms_per_step'], d['roofline']['all_conv_kernels'])" >> $out/ab.txt || exit 1
  done
done
