set -e
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in base noring; do
  if [ $v = base ]; then unset RTG_DEV_LIB; else export RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_abl_$v.so; fi
  bash tools/dbg/ab_cfg.sh c2_${v}_$round --workload config2
  bash tools/dbg/ab_cfg.sh c4_${v}_$round --workload config4
done
done
