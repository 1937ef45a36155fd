# same-box bench lines: tools/dbg/ab_cfg.sh <tag> [workload args...]  -> gpurun_out/ab_<tag>.json
tag=$1; shift
timeout -k 10 280 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2> gpurun_out/ab_$tag.err | tail -1 > gpurun_out/ab_$tag.json
python -c "
import json,sys
d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', d['ms_per_step'], d['value'], d['dtype'])"
