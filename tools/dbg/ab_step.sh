#!/bin/bash
# dev: A/B of the train step on ONE box: bench.py with the regular library and with a dev library, alternating
# usage: tools/dbg/ab_step.sh <dev .so> [workload] [reps]
dev=$1; wl=${2:-config2}; reps=${3:-2}
for r in $(seq $reps); do
  for L in "" "$dev"; do
    RTG_DEV_LIB=$L timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${L:-regular}'.split('/')[-1], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
