# dev: same-box A/B of environment knobs over the default bench (two rounds each).  usage: run_knobs.sh "NAME=VAL ..." ...
mkdir -p gpurun_out/r04z
for round in 1 2; do
  for spec in "$@"; do
    env $spec timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('[$spec] round $round', d['ms_per_step'])"
  done
done
