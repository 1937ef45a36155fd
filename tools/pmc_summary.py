#!/usr/bin/env python3
"""Summarise the PMC passes of tools/pmc_pass.sh over bench.py into per-kernel HBM traffic and matrix-pipe figures.

    python tools/pmc_summary.py gpurun_out/pmc_bench profiles/<name>_pmc.json

Steady-state launches only (from the 2nd adamw-delimited step to the end; the first step holds the block-shape tuner's
trial launches).  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts the 128-B requests of coalesced reads at 64 B, so it is doubled (checked on a conv launch of known traffic:
32->32 channels, 32 x 8192 samples, k7 d9: 2 x FETCH_SIZE = 36.8 MB against 33.5 MB of input + 10 % halo re-reads, and
WRITE_SIZE = 33.5 MB exactly)."""
import csv
import json
import re
import sys
from collections import OrderedDict, defaultdict


def load(path):
    d = OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = int(r['Dispatch_Id'])
            e = d.setdefault(k, {'name': r['Kernel_Name'], 'dur': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
            e[r['Counter_Name']] = float(r['Counter_Value'])
    return list(d.values())


def steady(rows):
    ad = [i for i, r in enumerate(rows) if 'adamw' in r['name']]
    gaps = [(ad[i + 1] - ad[i], ad[i + 1]) for i in range(len(ad) - 1)]
    big = max(g for g, _ in gaps[len(gaps) // 2:])
    ends = [e for g, e in gaps if g >= big - 4]
    ends = [max(a for a in ad if a - e < 8 and a >= e) for e in ends]
    return rows[ends[0] + 1:], len(ends) - 1 + 1      # launches after the first complete step; (whole steps + tail)


def short(n):
    m = re.search(r'(conv1d_mfma_group_kernel<[^>]*>|conv1d_mfma_kernel<[^>]*>|dconv_kernel<[^>]*>|dwgrad_kernel<[^>]*>|wgrad_kernel<[^>]*>|[A-Za-z_0-9]+_kernel)', n)
    return m.group(1) if m else n[:60]


def main():
    src, out = sys.argv[1], sys.argv[2]
    res = defaultdict(lambda: defaultdict(float))
    for sub in ('fetch', 'write', 'sq'):
        rows, _ = steady(load(f'{src}/{sub}/p_counter_collection.csv'))
        for r in rows:
            k = short(r['name'])
            a = res[k]
            a[f'n_{sub}'] += 1
            a[f'dur_{sub}'] += r['dur']
            for c, v in r.items():
                if c not in ('name', 'dur'):
                    a[c] += v
    table = {}
    for k, a in res.items():
        n = a['n_fetch'] or 1
        e = {'launches': int(n), 'avg_us': round(a['dur_fetch'] / n / 1e3, 2),
             'hbm_read_MB_per_launch': round(2 * a['FETCH_SIZE'] * 1024 / n / 1e6, 3),
             'hbm_write_MB_per_launch': round(a['WRITE_SIZE'] * 1024 / max(a['n_write'], 1) / 1e6, 3)}
        e['hbm_GBps'] = round((2 * a['FETCH_SIZE'] * 1024 / n + a['WRITE_SIZE'] * 1024 / max(a['n_write'], 1)) /
                              max(a['dur_fetch'] / n, 1.0), 1)
        if a['GRBM_GUI_ACTIVE']:
            e['clock_GHz'] = round(a['GRBM_GUI_ACTIVE'] / 8 / a['dur_fetch'], 2)
        tot = a['TCC_HIT_sum'] + a['TCC_MISS_sum']
        if tot:
            e['l2_hit'] = round(a['TCC_HIT_sum'] / tot, 3)
        if a['SQ_VALU_MFMA_BUSY_CYCLES'] and a['dur_sq']:
            # cycles in which a SIMD's matrix pipe is busy, summed over the 1024 SIMDs, against SIMDs x duration x clock
            clk = e.get('clock_GHz', 2.4)
            e['mfma_busy_frac'] = round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * a['dur_sq'] * clk), 3)
            wc = a['SQ_WAVE_CYCLES'] or 1
            e['wave_wait_any'] = round(a['SQ_WAIT_ANY'] / wc, 3)
            e['wave_wait_inst'] = round(a['SQ_WAIT_INST_ANY'] / wc, 3)
            e['lds_bank_conflict_frac'] = round(a['SQ_LDS_BANK_CONFLICT'] / wc, 4)
        table[k] = e
    table = dict(sorted(table.items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches']))
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({'source': 'rocprofv3 --pmc (3 separate passes: SQ, FETCH_SIZE+GRBM, WRITE_SIZE+TCC) over bench.py',
               'src_digest': bench._src_digest(),   # kernel sources the counters were taken on (bench.py flags a mismatch)
               'note': 'per-launch averages over steady-state launches; kernels run one at a time under counter '
                       'collection, so durations are serial (no stream overlap)', 'kernels': table}, open(out, 'w'), indent=1)
    for k, e in list(table.items())[:14]:
        print(k, e)


if __name__ == '__main__':
    main()
