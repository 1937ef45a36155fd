set -e
cd $GRAFT_REPO_ROOT
for v in base late; do
  echo "== $v"
  if [ $v = base ]; then unset RTG_DEV_LIB; else export RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_dev_$v.so; fi
  for bf in "" 1; do
    echo "-- BD_BF=$bf"
    BD_BF=$bf BD_PICK=23,24,27,29,31 timeout -k 10 200 python tools/dbg/bench_dconv.py wgrad 2>&1 | grep "^wgrad" | cut -c1-330 || true
    BD_WT=1 BD_BF=$bf BD_PICK=0,1,2,3 timeout -k 10 200 python tools/dbg/bench_dconv.py 2d 2>&1 | grep "^wgrad2d" | cut -c1-330 || true
  done
done
