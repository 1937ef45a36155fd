cd $GRAFT_REPO_ROOT
for round in 1 2; do
  bash tools/dbg/ab_cfg.sh c3_fp_$round --workload config3
  for m in 256 512; do RTG_DEV_MINC=$m bash tools/dbg/ab_cfg.sh c3bf_${m}_$round --workload config3 --bf16-maps; done
done
