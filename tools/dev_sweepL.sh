#!/bin/bash
# dev: sweep of row length at constant total columns over the block shapes (512->512 k5)
for L in 10 15 21 34 64 128; do
  B=$((7040 / L))
  for t in 121 122 124 221 222 224 141 142 212; do
    BENCH_TILE=$t python tools/bench_conv.py fwd $B 512 512 $L 5 1 1 2 1 30 2>&1 | grep "TF/s\|Error" | sed "s/^/t$t /"
  done
done
true
