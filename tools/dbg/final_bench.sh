#!/bin/bash
# the round's bench lines on the final host code (the PMC summaries under profiles/ are read by the roofline block)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06zz; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err
timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err
timeout -k 10 300 python bench.py --workload config3 --bf16-maps --no-cpu-baseline > $O/bench_config3_bf16maps.json 2> $O/bench_config3_bf16maps.err
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline > $O/bench_config4.json 2> $O/bench_config4.err
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err
for w in config2 config3 config3_bf16maps config4 config5; do python tools/show_bench.py $O/bench_$w.json | head -1; done
