#!/bin/bash
# dev: ablation libraries of rtg_dwgrad.hip (results are wrong by design): librtg_dev_<name>.so = the regular objects with
# rtg_dwgrad.hip recompiled with -D<flag>; run a tool with RTG_DEV_LIB pointing at one
set -e
cd "$(dirname "$0")/../../transtacos-retunegan_amd"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unused-result $flags -c csrc/rtg_dwgrad.hip -o /tmp/rtg_dwgrad_$name.o
  others=$(ls csrc/*.o | grep -v rtg_dwgrad.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librtg_dev_$name.so /tmp/rtg_dwgrad_$name.o $others
  echo "built librtg_dev_$name.so ($flags)"
done
