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
agg = {}
for kernel, variant, flop, e0, e1, label in rec:
    a = agg.setdefault((label, variant), [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flop
tot = sum(a[1] for a in agg.values())
print(f'total conv ms {tot:.2f}')
for (label, variant), (n, ms, flop) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f'{ms:8.3f} ms  n={n:2d}  {flop / ms / 1e9:7.2f} TF/s  v{variant:<5d} {label}')
