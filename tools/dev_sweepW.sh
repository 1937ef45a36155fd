#!/bin/bash
# dev: weight-gradient sweep of row length at constant total positions (512->512 k5 and 256->512 k5 s3)
for L in 10 15 21 34 51 64 128; do
  B=$((7040 / L))
  python tools/bench_conv.py wgrad $B 512 512 $L 5 1 1 2 1 30 2>&1 | grep "TF/s\|Error"
done
for L in 28 44 61 102 152; do
  B=$((21120 / L))
  python tools/bench_conv.py wgrad $B 256 512 $L 5 3 1 2 1 30 2>&1 | grep "TF/s\|Error"
  python tools/bench_conv.py fwd $B 256 512 $L 5 3 1 2 1 30 2>&1 | grep "TF/s\|Error"
done
true
