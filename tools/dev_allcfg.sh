#!/bin/bash
# dev: one bench line per workload / operand type (for the numbers quoted in README.md / DESIGN.md)
out=$1; mkdir -p $(dirname $out)
run() { timeout -k 10 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'], d['value'], d['dtype'], d['roofline'].get('kernel'), d['roofline'].get('achieved'), d['roofline'].get('frac'))" >> $out; }
run --workload config4 --steps 10 --warmup 3
run --workload config5 --steps 10 --warmup 3
run --workload config2 --dtype bf16 --steps 20 --warmup 5
run --workload config4 --dtype bf16 --steps 10 --warmup 3
run --workload config3 --steps 8 --warmup 3
