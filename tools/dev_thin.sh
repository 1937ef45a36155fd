#!/bin/bash
# dev: micro-benchmarks of the one-channel bandwidth shapes
python tools/bench_conv.py fwd 64 512 1 128 3 1 1 1 1 50 2>&1 | grep TF
python tools/bench_conv.py fwd 32 512 1 32 3 1 1 1 1 50 2>&1 | grep TF
python tools/bench_conv.py fwd 704 512 1 10 3 1 1 1 1 50 2>&1 | grep TF
python tools/bench_conv.py fwd 32 32 1 8192 7 1 1 3 1 50 2>&1 | grep TF
python tools/bench_conv.py fwd 64 1 32 8192 15 1 1 7 1 50 2>&1 | grep TF
python tools/bench_conv.py fwd 64 1 512 128 3 1 1 1 1 50 2>&1 | grep TF
python tools/bench_conv.py fwd 192 1 32 2731 5 3 1 2 1 50 2>&1 | grep TF
true
