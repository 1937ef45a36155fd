mkdir -p gpurun_out/r03y
for v in base nostage nostw noa nob nomma nobar noab; do
  if [ $v = base ]; then L=""; else L="$PWD/transtacos-retunegan_amd/librtg_dev_$v.so"; fi
  echo "== $v" >> gpurun_out/r03y/abl.log
  RTG_DEV_LIB=$L BD_BF=1 BD_PICK=1,2 timeout -k 10 120 python tools/dbg/bench_dconv.py 2d 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03y/abl.log
  RTG_DEV_LIB=$L BD_BF=1 BD_PICK=0,11 timeout -k 10 120 python tools/dbg/bench_dconv.py fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03y/abl.log
done
