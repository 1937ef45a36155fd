#!/bin/bash
# dev: micro-benchmarks of the half-filled 512-channel tails
for sh in "32 512 512 32 5 1 1 2 1" "64 512 512 32 5 1 1 2 1" "32 512 512 64 5 1 1 2 1" "96 512 512 34 5 1 1 2 1" "320 512 512 21 5 1 1 2 1" "32 128 128 256 7 1 1 3 1"; do
  for t in 0 111 112 114 121; do
    BENCH_TILE=$t python tools/bench_conv.py fwd $sh 30 2>&1 | grep "TF/s" | sed "s/^/t$t /"
  done
done
true
