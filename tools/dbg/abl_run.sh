cd $GRAFT_REPO_ROOT
for v in base slotskip; do
  echo "== $v"
  if [ $v = base ]; then unset RTG_DEV_LIB; else export RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_abl_$v.so; fi
  for io in 1 7; do
    BD_IO=$io BD_BF=1 BD_WT=1 BD_PICK=0,1,3 timeout -k 10 200 python tools/dbg/bench_dconv.py dgrad2d 2>&1 | grep "2d" | cut -c1-200
  done
  bash tools/dbg/ab_cfg.sh c3bf_$v --workload config3 --bf16-maps
done
