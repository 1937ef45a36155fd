set -e
cd $GRAFT_REPO_ROOT
for v in base vcache; do
  echo "== $v"
  if [ $v = base ]; then unset RTG_DEV_LIB; else export RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_abl_$v.so; fi
  for bf in "" 1; do
  echo "-- BD_BF=$bf"
  BD_BF=$bf BD_WT=1 BD_PICK=0,1,2,3 timeout -k 10 200 python tools/dbg/bench_dconv.py dgrad2d 2>&1 | grep "2d" | cut -c1-170 || true
  BD_BF=$bf BD_WT=1 BD_PICK=0,1,2,3 timeout -k 10 200 python tools/dbg/bench_dconv.py 2d 2>&1 | grep "^fwd2d" | cut -c1-250 || true
  done
done
