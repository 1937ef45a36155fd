#!/bin/bash
# dev: ablation libraries of ONE kernel source (results are wrong by design): librtg_dev_<name>.so = the regular objects with
# csrc/<stem>.hip recompiled with extra flags.  usage: tools/dbg/abl.sh <stem> name:"-DFLAG ..." ...; run a tool with
# RTG_DEV_LIB pointing at one (the loader refuses ablation builds otherwise)
set -e
cd "$(dirname "$0")/../../transtacos-retunegan_amd"
stem=$1; shift
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Wno-comment -Wno-unused-result -DRTG_ABLATION=1 $flags -c csrc/$stem.hip -o /tmp/${stem}_$name.o
    objs=/tmp/${stem}_$name.o
    if [ "$stem" != rtg_elem ]; then   # rtg_build_info lives in rtg_elem.hip: the library must say ABLATION whatever the stem
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Wno-comment -Wno-unused-result -DRTG_ABLATION=1 -c csrc/rtg_elem.hip -o /tmp/rtg_elem_abl_$name.o
      objs="$objs /tmp/rtg_elem_abl_$name.o"
    fi
    others=$(ls csrc/*.o | grep -v "csrc/$stem.o" | grep -v "csrc/rtg_elem.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librtg_dev_$name.so $objs $others
    echo "built librtg_dev_$name.so ($flags)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
