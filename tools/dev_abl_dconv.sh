#!/bin/bash
# Ablation builds of the dense conv kernel (fp32 tensors): tools/dev_abl_dconv.sh name=-DFLAG [name=-DFLAG ...]
# -> transtacos-retunegan_amd/librtg_abl_<name>.so (rtg_dconv.hip recompiled with the flag, the other objects as built;
# rtg_build_info says ABLATION, rtg/lib.py refuses such a library as the product).  Run tools with RTG_DEV_LIB=<that file>.
set -e
cd "$(dirname "$0")/../transtacos-retunegan_amd"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unused-result"
mkdir -p /tmp/rtg_abl
$CC -DRTG_ABLATION=1 -c csrc/rtg_elem.hip -o /tmp/rtg_abl/rtg_elem.o &
pids=($!)
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  $CC ${flags//,/ } -c csrc/rtg_dconv.hip -o /tmp/rtg_abl/dconv_$name.o &
  pids+=($!)
  if [ -n "$ABL_IO" ]; then       # ABL_IO=1: the bf16-tensor instances (rtg_dconv_io{1,2,3}.hip) as well
    for i in 1 2 3; do
      $CC ${flags//,/ } -c csrc/rtg_dconv_io$i.hip -o /tmp/rtg_abl/dconv_io${i}_$name.o &
      pids+=($!)
    done
  fi
done
for p in "${pids[@]}"; do wait $p; done
others=$(ls csrc/*.o | grep -v "csrc/rtg_dconv.o" | grep -v "csrc/rtg_elem.o")
[ -n "$ABL_IO" ] && others=$(echo "$others" | grep -v "csrc/rtg_dconv_io")
for spec in "$@"; do
  name=${spec%%=*}
  ios=""; [ -n "$ABL_IO" ] && ios="/tmp/rtg_abl/dconv_io1_$name.o /tmp/rtg_abl/dconv_io2_$name.o /tmp/rtg_abl/dconv_io3_$name.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librtg_abl_$name.so /tmp/rtg_abl/dconv_$name.o $ios /tmp/rtg_abl/rtg_elem.o $others
  echo built librtg_abl_$name.so
done
