#!/bin/bash
# dev: same-box A/B of environment settings over the default bench: dev_env_ab.sh <out file> "VAR=val ..." "VAR=val ..." ...
out=$1; shift; mkdir -p $(dirname $out)
for round in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('[$e] round $round', d['ms_per_step'])" >> $out || exit 1
  done
done
