#!/bin/bash
# dev: instruction-cache counters of single conv shapes (rocprofv3 --pmc, kernel trace only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for shape in "fwd 32 128 128 256 7 1 1 3 1 20" "fwd 64 512 512 128 5 1 1 2 1 20" "fwd 32 32 32 8192 7 1 9 27 1 20"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/s$i -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $shape > $OUT/s$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/s$i/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k,v in agg.items():
    if 'conv1d' in k: print('$shape',k, n[k], {c: round(x/max(n[k],1)) for c,x in v.items()})
PY
  rm -rf $OUT/s$i
done
