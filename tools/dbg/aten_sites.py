#!/usr/bin/env python3
"""dev: which Python lines of the package launch ATen kernels in one eager config-2 train step"""
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import bench  # noqa: E402
from train import Trainer  # noqa: E402
import hparam as hp  # noqa: E402

desc, use_mpd, use_mtd, d_times, batch, T = bench.WORKLOADS['config2']
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev='cuda')
data = bench.synthetic_batch(batch, T, 1, 'cuda')
for _ in range(3):
    tr.train_step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(*data)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    # the innermost frame inside the repository; autograd-engine launches have no Python stack
    frame = 'autograd engine (no Python frame)'
    for f in ev.stack:
        if 'transtacos-retunegan_amd' in f or '/bench.py' in f:
            frame = f.split('transtacos-retunegan_amd/')[-1]
            break
    sites[(ev.name, frame if ev.stack else str(ev.input_shapes)[:90], tuple(k.name[:40] for k in ev.kernels)[:1])] += 1
for (name, frame, k), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f'{n:3d}  {name:22s} {frame[:90]:92s} {k}')
print('total launches', sum(sites.values()))
