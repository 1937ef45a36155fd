#!/bin/bash
# dev: one SQ-counter pass over a python tool, per-kernel averages printed.  usage: pmc_quick.sh <outdir> <script> [args...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o p -- python3 "$@" > $OUT/sq.log 2>&1 || { echo "sq failed"; tail -5 $OUT/sq.log; }
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq2 -o p -- python3 "$@" > $OUT/sq2.log 2>&1 || { echo "sq2 failed"; tail -5 $OUT/sq2.log; }
python3 - $OUT <<'PY'
import csv, sys, re, glob
from collections import defaultdict
for sub in ('sq', 'sq2'):
    fs = glob.glob(f'{sys.argv[1]}/{sub}/**/p_counter_collection.csv', recursive=True)
    if not fs: print('no csv for', sub); continue
    d = {}
    for r in csv.DictReader(open(fs[0])):
        e = d.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'dur': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    agg = defaultdict(lambda: defaultdict(float))
    for e in d.values():
        m = re.search(r'([A-Za-z_0-9]+_kernel(<[^>]*>)?)', e['name'])
        k = m.group(1) if m else e['name'][:50]
        a = agg[k]; a['n'] += 1
        for c, v in e.items():
            if c != 'name': a[c] += v
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['dur'])[:12]:
        n = a['n']
        print(f'{k[:70]:70s} n={int(n):4d} us={a["dur"]/n/1e3:8.1f} ' + ' '.join(f'{c[3:]}={v/n:.3g}' for c, v in a.items() if c.startswith('SQ_')))
PY
