#!/bin/bash
# build several diagnostic variants: tools/dev_variants.sh name1:"flags" name2:"flags" ... -> transtacos-retunegan_amd/librtg_dev_<name>.so
set -e
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  tools/dev_build.sh $flags > /dev/null
  mv transtacos-retunegan_amd/librtg_dev.so transtacos-retunegan_amd/librtg_dev_$name.so
  echo "built $name ($flags)"
done
