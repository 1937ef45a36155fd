set -e
cd $GRAFT_REPO_ROOT
for v in base mmaonly nomem nobar; do
  echo "== $v"
  if [ $v = base ]; then unset RTG_DEV_LIB; else export RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_abl_$v.so; fi
  BD_WT=1 BD_PICK=0,1,2 timeout -k 10 200 python tools/dbg/bench_dconv.py dgrad2d 2>&1 | grep "2d" | cut -c1-170 || true
  BD_WT=1 BD_PICK=0,1,2 timeout -k 10 200 python tools/dbg/bench_dconv.py 2d 2>&1 | grep "^fwd2d" | cut -c1-250 || true
  BD_PICK=0,3,4,7,11,15,16 timeout -k 10 200 python tools/dbg/bench_dconv.py fwd dgrad poly 2>&1 | grep "^fwd\|^dgrad\|^poly" | cut -c1-170 || true
done
