#!/usr/bin/env python3
"""labels of the bandwidth-kernel launches of one step (wn_bwd: layers and split-partial MB per launch) (dev tool)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
from rtg import ops  # noqa: E402
from train import Trainer  # noqa: E402
import hparam as hp  # noqa: E402

desc, use_mpd, use_mtd, d_times, batch, T = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'config2']
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=use_mpd, use_mtd=use_mtd, d_train_times=d_times, dev='cuda')
data = bench.synthetic_batch(batch, T, 1, 'cuda')
for _ in range(3):
    tr.train_step(*data)
ops.PROFILE = []
tr.train_step(*data)
torch.cuda.synchronize()
rec, ops.PROFILE = ops.PROFILE, None
for kernel, variant, flop, e0, e1, label, nb in rec:
    if kernel.startswith('bw:wn_bwd') or kernel.startswith('bw:adamw'):
        print(f'{kernel[3:]:8s} {e0.elapsed_time(e1) * 1e3:7.1f} us  algorithmic {nb / 1e6:7.1f} MB  {label}')
