mkdir -p gpurun_out/r03G
for v in base noload nostore nofma; do
  if [ $v = base ]; then L=""; else L="$PWD/transtacos-retunegan_amd/librtg_dev_$v.so"; fi
  echo "== $v" >> gpurun_out/r03G/t2.log
  RTG_DEV_LIB=$L timeout -k 10 120 python tools/dbg/bench_thin2d.py 2>&1 | grep -v amdgpu >> gpurun_out/r03G/t2.log
done
cat gpurun_out/r03G/t2.log
