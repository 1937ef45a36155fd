mkdir -p gpurun_out/r04z
for round in 1 2; do
for v in base nset1 nset2; do
  for wl in config4 config2; do
    RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_dev_$v.so timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$v', '$wl', 'round $round', d['ms_per_step'])"
  done
done
done
