#!/usr/bin/env python3
"""Which host-side op issues every ATen (non-rtg) GPU kernel of one train step (dev tool): torch.profiler with stacks."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
from train import Trainer  # noqa: E402
import hparam as hp  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'config2'
desc, use_mpd, use_mtd, d_times, batch, T = bench.WORKLOADS[wl]
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev='cuda')
data = bench.synthetic_batch(batch, T, 1, 'cuda')
for _ in range(4):
    tr.train_step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(*data)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.kernels and not any(c.kernels for c in e.cpu_children) and
       not any('rtg' in k.name or 'anonymous' in k.name for k in e.kernels)]
agg = {}
for e in evs:
    st = [s for s in (e.stack or []) if 'transtacos' in s or 'train.py' in s][:3]
    par, p = [], e.cpu_parent
    while p is not None and len(par) < 3:
        par.append(p.name)
        p = p.cpu_parent
    k = (e.name, tuple(st) if st else tuple(par))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += sum(k.duration for k in e.kernels)
print('n_events', len(evs))
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{n:4d} {us:8.1f} us  {k[0]:24s} {" <- ".join(k[1])[:260]}')
