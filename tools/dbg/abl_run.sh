set -e
cd $GRAFT_REPO_ROOT
for F in 1000000000 0 6000 20000; do
  echo "== RTG_DEV_XCDF=$F"
  export RTG_DEV_XCDF=$F
  for bf in 1 ""; do
    echo "-- BD_BF=$bf"
    BD_WT=1 BD_BF=$bf BD_PICK=0,1,3,4,5 timeout -k 10 200 python tools/dbg/bench_dconv.py dgrad2d 2>&1 | grep "2d" | cut -c1-150 || true
  done
  bash tools/dbg/ab_cfg.sh c3_$F --workload config3
  bash tools/dbg/ab_cfg.sh c4_$F --workload config4
done
