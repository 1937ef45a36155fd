#!/bin/bash
# Diagnostic build: librtg_dev.so = the regular objects with rtg_conv1d*.hip recompiled with extra flags (default -DRTG_STAMPS).
# usage: tools/dev_build.sh [extra hipcc flags...]   then run tools with RTG_DEV_LIB=$PWD/transtacos-retunegan_amd/librtg_dev.so
set -e
cd "$(dirname "$0")/../transtacos-retunegan_amd"
EXTRA="${@:--DRTG_STAMPS}"
mkdir -p /tmp/rtg_dev
pids=()
for f in csrc/rtg_conv1d.hip csrc/rtg_conv1d_t*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unused-result $EXTRA -c $f -o /tmp/rtg_dev/$(basename ${f%.hip}).o &
  pids+=($!)
done
# rtg_build_info (rtg_elem.hip) must say ABLATION for every diagnostic library: rtg/lib.py refuses it as the product library
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unused-result -DRTG_ABLATION=1 -c csrc/rtg_elem.hip -o /tmp/rtg_dev/rtg_elem.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
others=$(ls csrc/*.o | grep -v rtg_conv1d | grep -v "csrc/rtg_elem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librtg_dev.so /tmp/rtg_dev/*.o $others
echo built librtg_dev.so
