cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for w in config4 config5; do
    bash tools/dbg/ab_cfg.sh ${w}_pair_$round --workload $w
    timeout -k 10 280 python tools/dbg/ab_pair2d.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload $w 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${w}_unpaired_$round', d['ms_per_step'])"
  done
done
