#!/usr/bin/env python3
"""Per-layer timing of one train step (dev tool): every conv launch bracketed by events, grouped by (pass, layer)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
from rtg import ops  # noqa: E402
from train import Trainer  # noqa: E402
import hparam as hp  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'config2'
desc, use_mpd, use_mtd, d_times, batch, T = bench.WORKLOADS[wl]
hp.compute_dtype = os.environ.get('RTG_DTYPE', 'fp32')
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev='cuda')
data = bench.synthetic_batch(batch, T, 1, 'cuda')
for _ in range(3):
    tr.train_step(*data)
torch.cuda.synchronize()
ops.PROFILE = []
tr.train_step(*data)
torch.cuda.synchronize()
rec, ops.PROFILE = ops.PROFILE, None
bw = [r for r in rec if r[0].startswith('bw:')]
rec = [r for r in rec if not r[0].startswith('bw:')]
agg = {}
for kernel, variant, flop, e0, e1, label, _nb in rec:
    a = agg.setdefault((label, variant), [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flop
tot = sum(a[1] for a in agg.values())
print(f'total conv ms {tot:.2f}')
only = os.environ.get('LP_ONLY')  # 'G': only the generator's layers
items = [kv for kv in agg.items() if not only or (only == 'G') == ('discriminators' not in kv[0][0])]
for (label, variant), (n, ms, flop) in sorted(items, key=lambda kv: -kv[1][1])[:int(os.environ.get("LP_TOP", "70"))]:
    print(f'{ms:8.3f} ms  n={n:2d}  {flop / ms / 1e9:7.2f} TF/s  v{variant:<5d} {label}')

# ---- totals by (pass, model part)
import collections
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for kernel, variant, flop, e0, e1, label, _nb in rec:
    parts = label.split()
    name = parts[1]
    if name.startswith('discriminators'):
        B = int(parts[2][1:])
        model = 'MPD' if ('B192' in label or 'B320' in label or 'B448' in label or 'B704' in label or
                          'B96 ' in label or 'B160' in label or 'B224' in label or 'B352' in label) else 'MSD'
    else:
        model = 'G'
    k = (model, parts[0])
    tot[k][0] += e0.elapsed_time(e1); tot[k][1] += flop; tot[k][2] += 1
print('---- by model / pass')
for k in sorted(tot):
    ms, fl, n = tot[k]
    print(f'{k[0]:4s} {k[1]:6s} n={n:4d} {ms:7.2f} ms  {fl / ms / 1e9:6.1f} TF/s')
for m in ('G', 'MSD', 'MPD'):
    ms = sum(v[0] for k, v in tot.items() if k[0] == m); fl = sum(v[1] for k, v in tot.items() if k[0] == m)
    print(f'{m}: {ms:.2f} ms {fl / 1e9:.0f} GFLOP {fl / ms / 1e9:.1f} TF/s')
print('---- bandwidth kernels')
agg_bw = {}
for kernel, variant, flop, e0, e1, label, nb in bw:
    a = agg_bw.setdefault(kernel[3:], [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += nb
for k, (n, ms, nb) in sorted(agg_bw.items()):
    print(f'{k:10s} n={n:3d} {ms:7.3f} ms  {nb / ms / 1e6:8.1f} GB/s')
